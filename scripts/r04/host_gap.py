"""What a 1,024-query step pays on the host between its kernels: the same steps (a) through GpuShardBackend.search_batched (what the
bench loop calls: torch allocations, stream lookup, ctypes), (b) through one bare ctypes call per step with everything preallocated.
python scripts/r04/host_gap.py [rows]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import frankensearch_amd as fa
from frankensearch_amd import _lib
from frankensearch_amd.sharded import GpuShardBackend
import bench

dev = torch.device("cuda", 0)
for rows in [int(a) for a in sys.argv[1:]] or [1_250_000, 10_000_000]:
    slab = bench.gen_corpus(0, rows, 384, dev)
    q = bench.gen_queries(2048, 384, dev)
    idx = fa.VectorIndex.from_device_slab(slab.data_ptr(), rows, 384, device=0, keepalive=slab)
    be = GpuShardBackend(idx, dev, batched=True)
    B, k, steps = 1024, 10, 200
    for i in range(10):
        be.search_batched(q[(i % 2) * B:(i % 2) * B + B], k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        be.search_batched(q[(i % 2) * B:(i % 2) * B + B], k)
    torch.cuda.synchronize(); a = (time.perf_counter() - t0) / steps * 1e3
    L = _lib.lib()
    rows_t = torch.empty((B, k), dtype=torch.int32, device=dev); sc_t = torch.empty((B, k), dtype=torch.float32, device=dev)
    cn_t = torch.empty((B,), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    fb = C.c_uint32()
    qp = [q[:B].data_ptr(), q[B:2 * B].data_ptr()]
    fn = L.fsgpu_search_topk_batched_device
    args = (idx._h, None, B, 384, k, None, rows_t.data_ptr(), sc_t.data_ptr(), cn_t.data_ptr(), stream, C.byref(fb))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        fn(idx._h, qp[i & 1], B, 384, k, None, args[6], args[7], args[8], stream, args[10])
    torch.cuda.synchronize(); b = (time.perf_counter() - t0) / steps * 1e3
    print(f"{rows} rows: through the backend wrapper {a:.4f} ms per step, bare ctypes call with preallocated buffers {b:.4f} ms per step", flush=True)
    idx.close(); del slab
