"""Micro-benchmark of the merge kernel through the C ABI (tuning aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from frankensearch_amd import _lib
from frankensearch_amd.errors import check

dev = torch.device("cuda", 0)
L = _lib.lib()

def ordkey(bits):
    bits = bits.astype(np.uint32)
    neg = (bits & 0x80000000) != 0
    return np.where(neg, ~bits, bits | 0x80000000).astype(np.uint64)

def run(nlists, k, nq, label, heavy=False):
    rng = np.random.default_rng(1)
    n_rows = nlists * 20000
    packed = np.empty((nq, nlists, k), np.uint64)
    for q in range(nq):
        for l in range(nlists):
            sc = np.sort(rng.standard_normal(20000 if not heavy else 200).astype(np.float32))[::-1][:k]
            rows = rng.integers(0, n_rows, k).astype(np.uint64)
            packed[q, l] = (sc.view(np.uint32).astype(np.uint64) << np.uint64(32)) | rows
    t = torch.from_numpy(packed.view(np.int64)).to(dev)
    rows = torch.empty((nq, k), dtype=torch.int32, device=dev)
    scores = torch.empty((nq, k), dtype=torch.float32, device=dev)
    counts = torch.empty((nq,), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def call():
        check(L.fsgpu_merge_topk_device(0, t.data_ptr(), nq, nlists, k, nlists * k, k, k, rows.data_ptr(), scores.data_ptr(), counts.data_ptr(), st))
    for _ in range(5): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): call()
    e1.record(); torch.cuda.synchronize()
    # expected survivors
    keys = (ordkey((packed[0] >> np.uint64(32)).astype(np.uint32)))
    thr = keys[:, k - 1].max()
    surv = int((keys >= thr).sum())
    # check
    flat = packed[0].reshape(-1)
    sc = (flat >> np.uint64(32)).astype(np.uint32).view(np.float32)
    order = np.lexsort((flat & np.uint64(0xFFFFFFFF), -sc.astype(np.float64)))[:k]
    ok = np.array_equal(rows[0].cpu().numpy().view(np.uint32), (flat[order] & np.uint64(0xFFFFFFFF)).astype(np.uint32))
    print(f"{label}: nlists={nlists} k={k} nq={nq}: {e0.elapsed_time(e1)/50*1e3:.1f} us/call, survivors~{surv}, ok={ok}")

run(512, 10, 1, "typical")
run(512, 10, 2, "typical nq2")
run(512, 64, 1, "k64")
run(512, 256, 1, "k256")
run(8, 10, 2, "cross-shard")
run(512, 10, 1, "heavy-overlap", heavy=True)
run(512, 30, 1, "k30")
run(1024, 30, 1, "k30 x1024 lists")
run(2048, 30, 1, "k30 x2048 lists")
run(512, 90, 1, "k90")
