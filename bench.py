#!/usr/bin/env python3
"""bench.py — queries/sec of the f16 cosine scan + top-k on a 10M x 384 f16 corpus (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one batch of B queries (default 1024) searched against the WHOLE corpus.  Default path: the batched
matrix-core scan (128 queries per HBM pass, provable candidate filter + exact-order re-score, results bit-identical
to the reference order); `--exact` times the exact VALU kernels (4-8 queries per pass) instead.  Each rank scans its
contiguous row shard; N > 1 adds one all-gather of the packed per-shard top-k (RCCL) and a merge.
The corpus is fixed (strong scaling: 10M rows total, sharded N ways) and already resident in HBM when the
timed region starts; queries are device-resident f32.  Rank 0 prints ONE JSON line.

Extra objects on that line (task contract):
  roofline     — the scan kernel against the HBM roof: algorithmic bytes (rows x dim x 2 per launch)
                 / mean launch duration measured live with HIP events on the launch stream.
  cpu_baseline — the oracle's AVX2+F16C restatement of the reference CPU path (kind "port"), timed on the
                 host cores of this box on a bounded sample of the same corpus, in the same run, and used as
                 the bit-exact parity checker for the GPU result on that sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16/bf16 matrix-core peak (same guide)
MFMA_I8_PEAK_TOPS = 5000.0     # v_mfma_i32_16x16x64_i8 issues at twice the f16 rate (the guide's microbenchmark: >= 3944 TOPS)
PROFILE_ROUND = "r06"   # profiles/<round>/pmc_summary.json: the PMC pass that belongs to this build's kernels
CLUSTERS = 64
NOISE = 0.30


def baseline_metric() -> str:
    """BASELINE.json's metric string (value = queries/sec of the exact top-k search; the p50 phase-1 latency it also names
    is reported next to it as p50_phase1_latency_ms)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except (OSError, ValueError, KeyError):
        return "queries/sec + p50 phase-1 latency, 10Mx384 f16 corpus, 1/2/4/8 MI355X"


def _fixture(first: int, n: int, dim: int, device, seed_base: int, as_f16: bool) -> torch.Tensor:
    """The reference's own bench generator (frankensearch/benches/fsvi_4bit_vs_incumbent.rs:56-101,344-365: xorshift64
    raw_vector, 64 normalised centroids, row i = normalize(centroid[i % 64] + 0.30 * raw_vector(i + 1)), f32 -> f16 RNE;
    queries = make_vector(centroids, q % 64, 0xdead0000 + q)) run by a small HIP kernel straight into HBM
    (fsgpu_bench_fixture_device): any row range of the corpus is the same bytes whatever the number of shards."""
    from frankensearch_amd import _lib
    from frankensearch_amd.errors import check

    out = torch.empty((n, dim), dtype=torch.float16 if as_f16 else torch.float32, device=device)
    check(_lib.lib().fsgpu_bench_fixture_device(device.index or 0, first, n, dim, CLUSTERS, NOISE, seed_base, 1 if as_f16 else 0,
                                                out.data_ptr(), None))
    return out


def gen_corpus(lo: int, hi: int, dim: int, device) -> torch.Tensor:
    return _fixture(lo, hi - lo, dim, device, 1, True)


def gen_queries(n: int, dim: int, device) -> torch.Tensor:
    return _fixture(0, n, dim, device, 0xDEAD0000, False)


def measured_copy_gbps(device) -> float:
    """What a plain device-to-device copy reaches on THIS box (read + written bytes per second), BASELINE.md section 3: the
    vendor peak is 8 TB/s, a float4 copy measures about 6.3 (MI355X_MICROARCH.md)."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=device)
    b = torch.empty_like(a)
    b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def cpu_baseline_and_parity(slab_dev: torch.Tensor, queries: torch.Tensor, k: int, rows_total: int, index_cls,
                            sample_rows: int = 1 << 62):
    """Oracle (AVX2+F16C restatement, the host cores this container may use) on the SAME slab, one query
    at a time; also the parity checker: the exact kernels' and the batched path's answers against it, rows and f32 bits."""
    from oracle import oracle

    oracle.build()
    sample = int(min(slab_dev.shape[0], sample_rows))   # default: the whole slab (7.68 GB of host memory at 10M x 384)
    host = slab_dev[:sample].contiguous().view(torch.int16).cpu().numpy().view(np.uint16)
    cores = os.cpu_count() or 1
    nq = 32
    q_host = queries[:nq].cpu().numpy()
    sub = index_cls.from_device_slab(slab_dev.data_ptr(), sample, slab_dev.shape[1], device=slab_dev.device.index or 0,
                                     keepalive=slab_dev)
    g_rows, g_scores, g_counts = sub.search_batch(q_host, k)
    # the batched matrix-core path must give the very same bits (checked on a 160-query batch: 128 + ragged 32)
    q_many = queries[:160].cpu().numpy()
    b_rows, b_scores, _, _ = sub.search_batched(q_many, k)
    e_rows, e_scores, _ = sub.search_batch(q_many, k)
    batched_ok = bool(np.array_equal(b_rows, e_rows) and np.array_equal(b_scores.view(np.uint32), e_scores.view(np.uint32)))
    # threads: the container's CPU quota when there is one (cgroup cpu.max; more runnable threads than quota only get
    # throttled), else the best of a few counts on one probe query each
    quota = None
    try:
        mx, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if mx != "max":
            quota = max(1, int(int(mx) / int(period)))
    except (OSError, ValueError):
        pass
    if quota is not None:
        nthreads = min(cores, quota)
        oracle.search_top_k(host[:200_000], q_host[0], k, nthreads=nthreads)  # start the worker pool
    else:
        best = None
        for cand in sorted({min(cores, c) for c in (16, 32, 64, 128)}):
            oracle.search_top_k(host[:200_000], q_host[0], k, nthreads=cand)
            t1 = time.perf_counter()
            oracle.search_top_k(host, q_host[0], k, nthreads=cand)
            dt1 = time.perf_counter() - t1
            if best is None or dt1 < best[0]:
                best = (dt1, cand)
        nthreads = best[1]
    # the corpus IS the reference bench recipe: a 4,096-row prefix (and the queries) against the oracle's restatement
    prefix_ok = bool(np.array_equal(host[:4096], oracle.clustered_corpus_f16(0, min(4096, sample), slab_dev.shape[1])) and
                     all(np.array_equal(q_host[i], oracle.clustered_query(i, slab_dev.shape[1])) for i in range(4)))
    for qi in range(3):                                  # warm-up passes (BASELINE.md section 2 protocol)
        oracle.search_top_k(host, q_host[qi], k, nthreads=nthreads)
    ok = True
    per_query = []
    t0 = time.perf_counter()
    for qi in range(nq):
        t1 = time.perf_counter()
        er, es = oracle.search_top_k(host, q_host[qi], k, nthreads=nthreads)
        per_query.append(time.perf_counter() - t1)
        ok &= bool(np.array_equal(g_rows[qi, :len(er)], er) and
                   np.array_equal(g_scores[qi, :len(es)].view(np.uint32), es.view(np.uint32)))
    dt = time.perf_counter() - t0
    # the batched matrix-core path against the oracle directly (not through the exact kernels): 8 more queries of its batch
    batched_vs_oracle = True
    for qi in (32, 33, 63, 64, 100, 127, 128, 159):
        er, es = oracle.search_top_k(host, q_many[qi], k, nthreads=nthreads)
        batched_vs_oracle &= bool(np.array_equal(b_rows[qi, :len(er)], er) and
                                  np.array_equal(b_scores[qi, :len(es)].view(np.uint32), es.view(np.uint32)))
    # one thread (anchor: the reference's published 18.9 GB/s per thread, docs/PERF_LEDGER.md:2149) on a smaller sample
    one_rows = min(sample, 1_500_000)   # 1.15 GB of rows: past the host's last-level cache
    oracle.search_top_k(host[:one_rows], q_host[0], k, nthreads=1)
    t1 = time.perf_counter()
    for qi in range(6):
        oracle.search_top_k(host[:one_rows], q_host[qi], k, nthreads=1)
    one_gbps = one_rows * slab_dev.shape[1] * 2 * 6 / (time.perf_counter() - t1) / 1e9
    sub.close()
    qps_sample = nq / dt
    row_bytes = slab_dev.shape[1] * 2
    gbps = sample * row_bytes * nq / dt / 1e9
    lat = np.sort(np.array(per_query)) * (rows_total / sample)   # scaled to a pass over the full corpus
    try:
        model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except (OSError, StopIteration):
        model = "unknown"
    return {
        "value": qps_sample * sample / rows_total,
        "unit": "queries/sec",
        "cores": nthreads,
        "kind": "port",
        "sample": f"{nq} queries x {'all' if sample == rows_total else 'the first'} {sample} rows of the same corpus, one at a time, "
                  f"after 3 warm-up passes" + ("" if sample == rows_total else f"; value scaled by {sample}/{rows_total} to the full corpus") +
                  f"; {gbps:.1f} GB/s of f16 on {nthreads} threads "
                  f"({cores} host cpus visible, cgroup quota {quota if quota is not None else 'none'}; {model})",
        "GBps": gbps,
        "one_thread_GBps": one_gbps,
        "p50_ms_per_query": float(lat[len(lat) // 2] * 1e3),
        "p95_ms_per_query": float(lat[min(len(lat) - 1, int(len(lat) * 0.95))] * 1e3),
        "cpu_model": model,
        "host_cpus_visible": cores,
        "corpus_prefix_equals_reference_recipe": prefix_ok,
        "parity_bit_exact": ok,
        "batched_path_equals_exact_path": batched_ok,
        "batched_path_equals_oracle_8_queries": batched_vs_oracle,
        "rows_scanned_per_query": sample,
    }


CHUNK = 1_000_000   # the adversarial corpora are generated in fixed 1M-row chunks, each from its own seed

def slab_checksum(slab: torch.Tensor, row_lo: int) -> int:
    """Position-sensitive checksum of a resident row shard (f16 bits as integers): sum over rows of (row's element sum) x
    ((global row id mod 1,000,003) + 1), mod 2^63.  Every rank reports it for ITS slab; rank 0 recomputes it from rows it generates
    itself — a rank holding the wrong rows (or a stale buffer) cannot produce the right number."""
    total = 0
    for a in range(0, slab.shape[0], CHUNK):
        part = slab[a:a + CHUNK].view(torch.int16)
        rs = torch.sum(part, dim=1, dtype=torch.int64)
        w = (torch.arange(row_lo + a, row_lo + a + part.shape[0], device=slab.device, dtype=torch.int64) % 1_000_003) + 1
        total = (total + int(torch.sum(rs * w).item())) & ((1 << 63) - 1)
    return total


def host_corpus(rows: int, dim: int, device) -> np.ndarray:
    """The WHOLE bench corpus in host memory (rows x dim f16 bits; 7.68 GB at 10M x 384), generated on `device` in 1M-row pieces by
    the same fixture kernel every rank built its shard with and copied out piece by piece: the oracle's input for the check of an
    N-rank answer.  Nothing here touches another rank's memory."""
    host = np.empty((rows, dim), dtype=np.uint16)
    for a in range(0, rows, CHUNK):
        b = min(rows, a + CHUNK)
        piece = gen_corpus(a, b, dim, device)
        host[a:b] = piece.view(torch.int16).cpu().numpy().view(np.uint16)
        del piece
    return host


def device_identity(index: int) -> dict:
    """What tells one GPU of the node from another in the record of an N-rank run: the HIP device's UUID / PCI address and, where
    the kernel driver exposes it, the board's unique_id."""
    out = {"device": index}
    try:
        p = torch.cuda.get_device_properties(index)
        out["name"] = p.name
        for attr in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id"):
            if hasattr(p, attr):
                out[attr] = str(getattr(p, attr))
        if all(k in out for k in ("pci_bus_id", "pci_device_id", "pci_domain_id")):
            bdf = f"{int(out['pci_domain_id']):04x}:{int(out['pci_bus_id']):02x}:{int(out['pci_device_id']):02x}.0"
            out["pci"] = bdf
            try:
                out["unique_id"] = open(f"/sys/bus/pci/devices/{bdf}/unique_id").read().strip()
            except OSError:
                pass
    except Exception as e:   # noqa: BLE001 — identification must never cost the line
        out["error"] = f"{type(e).__name__}: {e}"
    return out


def oracle_threads_for_bench() -> int:
    cores = os.cpu_count() or 1
    try:
        mx, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if mx != "max":
            return max(1, min(cores, int(int(mx) / int(period))))
    except (OSError, ValueError):
        pass
    return min(cores, 32)


def merged_answer_vs_oracle(host: np.ndarray, queries_host: np.ndarray, picks, rows, scores, counts, k: int) -> dict:
    """The N-rank answer itself — what came out of the all-gather + merge of a timed step — against the oracle's search_top_k over
    ALL rows of the corpus (search.rs:1013-1036 partitions, :1704-1720 merges: the result must be that of the unpartitioned scan):
    row ids and f32 score bits of `picks` queries of that step."""
    from oracle import oracle

    oracle.build()
    nthreads = oracle_threads_for_bench()
    ok, bad = True, []
    r, sc, c = rows.cpu().numpy(), scores.cpu().numpy(), counts.cpu().numpy()
    for qi in picks:
        er, es = oracle.search_top_k(host, queries_host[qi], k, nthreads=nthreads)
        same = bool(int(c[qi]) == len(er) and np.array_equal(r[qi, :len(er)].astype(np.uint32), er.astype(np.uint32)) and
                    np.array_equal(sc[qi, :len(es)].view(np.uint32), es.view(np.uint32)))
        ok &= same
        if not same:
            bad.append(int(qi))
    return {"equal": bool(ok), "queries_checked": [int(x) for x in picks], "mismatching_queries": bad, "oracle_threads": nthreads,
            "rows_scanned_by_the_oracle_per_query": int(host.shape[0])}



def _chunks(lo: int, hi: int):
    c = lo // CHUNK
    while c * CHUNK < hi:
        a, b = max(lo, c * CHUNK), min(hi, (c + 1) * CHUNK)
        yield c, a - c * CHUNK, b - c * CHUNK, a - lo, b - lo
        c += 1


def gen_uniform_corpus(lo: int, hi: int, dim: int, device) -> torch.Tensor:
    """SURVEY 8d's adversarial low-separation case: uniform-random unit vectors (seeded per 1M-row chunk), f32 -> f16 RNE.
    No cluster structure: the k-th best of 10M scores sits ~5 sigma out (0.26 at dim 384) with thousands of rows within a few
    hundredths below it — the candidate filter's margin has the least to work with."""
    out = torch.empty((hi - lo, dim), dtype=torch.float16, device=device)
    for c, a, b, oa, ob in _chunks(lo, hi):
        g = torch.Generator(device=device)
        g.manual_seed(0x5EED0000 + c)
        x = torch.randn((min(CHUNK, max(b, 1)), dim), generator=g, device=device, dtype=torch.float32)[a:b]
        out[oa:ob] = (x / x.norm(dim=1, keepdim=True)).half()
    return out


OUTLIER_DIMS = (3, 57, 101, 160, 222, 287, 313, 380)


def gen_outlier_corpus(lo: int, hi: int, dim: int, device) -> torch.Tensor:
    """Anisotropic corpus with outlier dimensions, as trained embedding models have: 256 clusters of Zipf sizes (cluster c
    holds ~1/(c+1) of the rows), noise 0.30, then 8 fixed dimensions scaled x10 before the normalisation — they carry most of
    every row's norm and stretch the corpus-wide int8 scale, so the other 376 dimensions quantise to a handful of levels."""
    g0 = torch.Generator(device=device)
    g0.manual_seed(0x0D1A)
    cent = torch.randn((256, dim), generator=g0, device=device, dtype=torch.float32)
    cent /= cent.norm(dim=1, keepdim=True)
    w = 1.0 / torch.arange(1, 257, device=device, dtype=torch.float64)
    cdf = torch.cumsum(w / w.sum(), 0).float()
    scale = torch.ones(dim, device=device)
    scale[[d for d in OUTLIER_DIMS if d < dim]] = 10.0
    out = torch.empty((hi - lo, dim), dtype=torch.float16, device=device)
    for c, a, b, oa, ob in _chunks(lo, hi):
        g = torch.Generator(device=device)
        g.manual_seed(0x0D1A0000 + c)
        n = min(CHUNK, max(b, 1))
        u = torch.rand((n,), generator=g, device=device)
        cl = torch.searchsorted(cdf, u).clamp_(max=255)
        x = cent[cl] + 0.30 * torch.randn((n, dim), generator=g, device=device, dtype=torch.float32)
        x = (x * scale)[a:b]
        out[oa:ob] = (x / x.norm(dim=1, keepdim=True)).half()
    return out


def adversarial_queries(kind: str, slab: torch.Tensor, n: int, device) -> torch.Tensor:
    g = torch.Generator(device=device)
    g.manual_seed(0xAD7 + len(kind))
    dim = slab.shape[1]
    if kind == "uniform":
        q = torch.randn((n, dim), generator=g, device=device, dtype=torch.float32)
    else:   # a stored row plus noise: every query has true neighbours inside its cluster
        pick = torch.randint(0, slab.shape[0], (n,), generator=g, device=device)
        q = slab[pick].float() + 0.2 / (dim ** 0.5) * torch.randn((n, dim), generator=g, device=device, dtype=torch.float32)
    return (q / q.norm(dim=1, keepdim=True)).contiguous()


def _oracle_threads() -> int:
    cores = os.cpu_count() or 1
    try:
        mx, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if mx != "max":
            return max(1, min(cores, int(int(mx) / int(period))))
    except (OSError, ValueError):
        pass
    return min(cores, 16)


def adversarial_section(kind: str, rows: int, dim: int, k: int, device, local_rank: int, steps: int = 6):
    """The batched exact search on a corpus the candidate filter likes least: throughput, how often the int8
    filter hands queries back, exact fallbacks — and EIGHT of the batched answers at full size against the oracle directly
    (rows and f32 score bits), plus the rest of a batch against the exact kernels."""
    import frankensearch_amd as fa
    from oracle import oracle

    slab = gen_uniform_corpus(0, rows, dim, device) if kind == "uniform" else gen_outlier_corpus(0, rows, dim, device)
    B = 1024
    queries = adversarial_queries(kind, slab, 2 * B, device)
    index = fa.VectorIndex.from_device_slab(slab.data_ptr(), rows, dim, device=local_rank, keepalive=slab)
    backend_cls = __import__("frankensearch_amd.sharded", fromlist=["GpuShardBackend"]).GpuShardBackend
    be = backend_cls(index, device, batched=True)
    fb = 0
    for i in range(2):
        be.search_batched(queries[(i % 2) * B:(i % 2) * B + B], k)
    f0 = index.batched_filter_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = be.search_batched(queries[(i % 2) * B:(i % 2) * B + B], k)
        fb += be.last_fallbacks
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    f1 = index.batched_filter_stats()
    last = ((steps - 1) % 2) * B
    g_rows, g_scores = out[0].cpu().numpy().astype(np.uint32), out[1].cpu().numpy()
    # the oracle on the same bytes, 8 queries of the last step at full size
    host = slab.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)
    qh = queries[last:last + B].cpu().numpy()
    nthreads = _oracle_threads()
    ok = True
    picks = [0, 1, 2, 3, 511, 512, 1000, 1023]
    for qi in picks:
        er, es = oracle.search_top_k(host, qh[qi], k, nthreads=nthreads)
        ok &= bool(np.array_equal(g_rows[qi, :len(er)], er) and np.array_equal(g_scores[qi, :len(es)].view(np.uint32), es.view(np.uint32)))
    del host
    # ... and 64 more against the exact kernels (a different code path over the same slab)
    e_rows, e_scores, _ = index.search_batch(qh[64:128], k, exact=True)
    same = bool(np.array_equal(g_rows[64:128], e_rows) and np.array_equal(g_scores[64:128].view(np.uint32), e_scores.view(np.uint32)))
    kth = float(np.median(g_scores[:, k - 1]))
    res = {"corpus": kind, "rows": rows, "queries_per_step": B, "queries_per_sec": steps * B / dt, "ms_per_step": dt / steps * 1e3,
           "int8_filter_active_after": bool(f1["int8_active"]), "int8_filter_copy_rotated": bool(index.filter_rotated()),
           "int8_filter_queries": f1["int8_queries"] - f0["int8_queries"],
           "refiltered_on_f16_queries": f1["refiltered_f16"] - f0["refiltered_f16"], "exact_fallback_queries": fb,
           "median_kth_score": kth, "batched_equals_oracle_rows_and_bits": ok, "oracle_checked_queries": len(picks),
           "batched_equals_exact_kernels_64_queries": same}
    index.close()
    del slab
    torch.cuda.empty_cache()
    return res


def flatten_for_the_driver(line: dict) -> None:
    """The driver's record of a bench line keeps the SCALAR fields of `roofline` (nested objects are dropped) and the tail of
    stdout: the north-star kernel's figures (the exact f16 scan against the HBM roof), the encoders' matrix-core fractions and the
    main-pass kernel's register spill count are therefore repeated as scalars inside `roofline` and at the top level, in place."""
    roof = line.get("roofline") or {}
    flat = {}
    ex = roof.get("exact_f16_scan")
    if isinstance(ex, dict):
        flat["exact_f16_frac"] = ex.get("frac")
        flat["exact_f16_ms"] = ex.get("avg_launch_ms")
        flat["exact_f16_GBps"] = ex.get("achieved")
        flat["exact_f16_traffic"] = ex.get("traffic")
        flat["exact_f16_algorithmic_bytes"] = ex.get("algorithmic_bytes_per_launch")
    enc = (line.get("encoders") or {}).get("minilm_l6") or {}
    if isinstance(enc.get("roofline"), dict):
        flat["encoder_queries_mfma_frac"] = enc["roofline"].get("frac")
        flat["encoder_queries_ms"] = enc.get("gpu_ms_per_batch")
    docs = enc.get("documents_32x512") or {}
    if isinstance(docs.get("roofline"), dict):
        flat["encoder_documents_mfma_frac"] = docs["roofline"].get("frac")
        flat["encoder_documents_ms"] = docs.get("gpu_ms_per_batch")
    if isinstance(roof.get("joint"), dict):
        flat["joint_frac"] = roof["joint"].get("frac")
    if isinstance(roof.get("hbm"), dict):
        flat["main_pass_hbm_frac"] = roof["hbm"].get("frac")
    res = main_pass_kernel_resources()
    if res:
        flat["main_pass_vgpr_count"] = res.get("vgpr_count")
        flat["main_pass_vgpr_spill_count"] = res.get("vgpr_spill_count")
        flat["main_pass_scratch_bytes"] = res.get("private_segment_fixed_size")
    roof.update({k: v for k, v in flat.items() if v is not None})
    top = {"roofline_" + k: v for k, v in flat.items() if v is not None}
    # (early in the line: right behind the contract's first keys)
    head = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step")
    rest = {k: v for k, v in line.items() if k not in head}
    first = {k: line[k] for k in head if k in line}
    for key in ("p50_latency_ms_single_query", "p50_phase1_latency_ms", "p50_phase0_latency_ms", "end_to_end_queries_per_sec",
                "end_to_end_many_queries_per_sec", "merged_answer_equals_oracle_8_queries", "sharded_handle_equals_oracle_8_queries",
                "ranks_in_collective", "comm_backend", "rccl_version", "strong_scaling_queries_per_sec", "weak_scaling_queries_per_sec",
                "single_shard_parity_bit_exact"):
        if key in rest and not isinstance(rest[key], dict):
            first[key] = rest.pop(key)
    line.clear()
    line.update(first)
    line.update(top)
    line.update(rest)


def main_pass_kernel_resources() -> dict | None:
    """.vgpr_count / .vgpr_spill_count / .private_segment_fixed_size of the headline main-pass instantiation
    (scan_wide_kernel<384, 1, 4, 3, 30, 0>), read from the metadata note of the device code inside the shipped libfsgpu.so."""
    import re
    import subprocess
    lib = os.path.join(ROOT, "frankensearch_amd", "libfsgpu.so")
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(lib) and os.path.exists(readelf)):
        return None
    name = "_ZN5fsgpu16scan_wide_kernelILi384ELi1ELi4ELi3ELi30ELi0EEEvNS_12MfmaScanArgsE"
    item = None
    try:
        import struct
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            # the library's .hip_fatbin section is a run of clang offload bundles (one per translation unit): magic, entry count,
            # then (offset, size, triple length, triple) per entry
            fat = os.path.join(td, "fat.bin")
            subprocess.check_call(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=60)
            blob = open(fat, "rb").read()
            magic = b"__CLANG_OFFLOAD_BUNDLE__"
            at = blob.find(magic)
            while at >= 0 and item is None:
                n, = struct.unpack_from("<Q", blob, at + 24)
                cur = at + 32
                for _ in range(n):
                    off, size, tl = struct.unpack_from("<QQQ", blob, cur)
                    triple = blob[cur + 24:cur + 24 + tl]
                    cur += 24 + tl
                    if b"gfx950" not in triple or size == 0:
                        continue
                    elf = blob[at + off:at + off + size]
                    if name.encode() not in elf:
                        continue
                    co = os.path.join(td, "dev.co")
                    with open(co, "wb") as f:
                        f.write(elf)
                    notes = subprocess.check_output([readelf, "--notes", co], stderr=subprocess.DEVNULL, timeout=60).decode(errors="replace")
                    pos = notes.find(".name:           " + name)
                    if pos < 0:
                        pos = notes.find(name)
                    start = notes.rfind("\n  - ", 0, pos)
                    end = notes.find("\n  - ", pos)
                    item = notes[start:end if end > 0 else len(notes)]
                    break
                at = blob.find(magic, at + 24)
    except Exception:
        return None
    if not item:
        return None
    out = {}
    for key in ("vgpr_count", "vgpr_spill_count", "private_segment_fixed_size", "sgpr_spill_count"):
        mm = re.search(r"\." + key + r":\s+(\d+)", item)
        if mm:
            out[key] = int(mm.group(1))
    return out or None


def exact_scan_roofline(index, queries, k: int, rows: int, dim: int, device):
    """The north-star's HBM target, timed in THIS run: the exact f16 kernel (scan_topk_kernel, one query per pass over the f16
    slab, reference operation order) with HIP events on its launch stream; algorithmic bytes = rows x dim x 2 per launch."""
    from frankensearch_amd.sharded import GpuShardBackend
    be = GpuShardBackend(index, device, batched=False)
    for i in range(3):
        be.search_unsharded(queries[i:i + 1], k)
    torch.cuda.synchronize()
    index.scan_stats(reset=True)
    index.set_profiling(True)
    n = 30
    for i in range(n):
        be.search_unsharded(queries[i:i + 1], k)
    torch.cuda.synchronize()
    index.set_profiling(False)
    ms, launches, srows = index.scan_stats(reset=True)
    per = ms / max(launches, 1)
    alg = srows // max(launches, 1) * dim * 2
    gbps = alg / (per * 1e-3) / 1e9 if per > 0 else 0.0
    return {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
            "kernel": "scan_topk_kernel<384, 1, 64> (exact f16 scan + wave top-k, one query per pass)", "avg_launch_ms": per,
            "launches": launches, "algorithmic_bytes_per_launch": alg, "traffic": None,
            "note": "north_star: >= 0.70 of the HBM roofline on the 10M x 384 f16 cosine scan"}


def encoder_section(device, local_rank: int):
    """Encoders next to their CPU baselines: MiniLM-L6 (random-init weights of that shape) on 256 query-like
    texts through the C ABI, against the oracle's multi-threaded f32 C restatement of the same forward (kind "port") on a 64-text
    sample; Model2Vec pooling (potion shape) against the oracle's per-text pool.  Outputs cross-checked in the same run."""
    import frankensearch_amd as fa
    from frankensearch_amd.synthetic import random_bert_weights
    from oracle import bert_oracle, oracle

    w = random_bert_weights(1, 30522, 384, 6, 1536)
    bert = fa.NativeEmbedder(w, device=local_rank)
    rng = np.random.default_rng(5)
    texts = [[101] + rng.integers(1000, 30000, int(rng.integers(6, 31))).tolist() + [102] for _ in range(256)]
    offs = np.zeros(257, dtype=np.uint32)
    offs[1:] = np.cumsum([len(t) for t in texts])
    ids = np.concatenate([np.asarray(t, dtype=np.int32) for t in texts])
    emb = np.empty((256, 384), dtype=np.float32)
    for _ in range(3):
        bert.embed_flat(ids, offs, emb)
    lat = []
    for _ in range(20):
        t0 = time.perf_counter()
        bert.embed_flat(ids, offs, emb)
        lat.append((time.perf_counter() - t0) * 1e3)
    gpu_ms = float(np.median(lat))
    one = []
    for i in range(24):
        t0 = time.perf_counter()
        bert.embed_flat(ids[offs[i]:offs[i + 1]], np.array([0, offs[i + 1] - offs[i]], dtype=np.uint32), emb[:1])
        one.append((time.perf_counter() - t0) * 1e3)
    bert.embed_flat(ids, offs, emb)
    # the document shape of an index build (index_builder.rs:191,416): 32 texts x 512 tokens, a plain M = 16,384 GEMM chain
    doc_ids = np.concatenate([np.concatenate([[101], rng.integers(1000, 30000, 510), [102]]).astype(np.int32) for _ in range(32)])
    doc_offs = (np.arange(33) * 512).astype(np.uint32)
    doc_emb = np.empty((32, 384), dtype=np.float32)
    # (an index build is a long run of such calls: the median of 100 after 30 — the clocks take tens of milliseconds to settle, and a
    # median of 8 cold calls read 10 % high)
    for _ in range(30):
        bert.embed_flat(doc_ids, doc_offs, doc_emb)
    dlat = []
    for _ in range(100):
        t0 = time.perf_counter()
        bert.embed_flat(doc_ids, doc_offs, doc_emb)
        dlat.append((time.perf_counter() - t0) * 1e3)
    doc_ms = float(np.median(dlat))

    def bert_flops(lengths):
        """21.23 MFLOP per token in the linears + 6 layers x 4 S x 384 per token of attention (S = the text's length)."""
        lengths = np.asarray(lengths, dtype=np.float64)
        return float(np.sum(lengths * 21.23e6 + lengths * 6 * 4 * lengths * 384))

    def enc_roofline(flops, ms):
        tf = flops / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": tf, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F16_PEAK_TFLOPS,
                "algorithmic_flops": flops, "ms": ms,
                "note": "whole forward at the C ABI (host pointers in and out) against the dense f16 matrix-core peak"}
    nthreads = _oracle_threads()
    cpu = bert_oracle.CForward(w, 6)
    sample = 64
    cpu.run(texts[:8], nthreads)
    t0 = time.perf_counter()
    want = cpu.run(texts[:sample], nthreads)
    cpu_dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    cpu.run(texts[:4], 1)
    cpu_one = (time.perf_counter() - t0) / 4
    err = float(np.max(np.abs(emb[:sample] - want)))
    cos = float(np.min(np.sum(emb[:sample] * want, axis=1)))
    bert.close()
    # Model2Vec (potion-multilingual-128M shape: 500,353 x 256 table)
    table = np.random.default_rng(0).standard_normal((500_353, 256)).astype(np.float32)
    m2v = fa.Model2VecEmbedder(table, device=local_rank)
    docs = [rng.integers(0, 500_353, int(rng.integers(4, 33))).astype(np.uint32) for _ in range(256)]
    moffs = np.zeros(257, dtype=np.uint32)
    moffs[1:] = np.cumsum([len(d) for d in docs])
    mids = np.concatenate(docs)
    mout = np.empty((256, 256), dtype=np.float32)
    for _ in range(3):
        m2v.embed_flat(mids, moffs, mout)
    t0 = time.perf_counter()
    for _ in range(20):
        m2v.embed_flat(mids, moffs, mout)
    m_gpu = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    mwant = np.stack([oracle.m2v_embed(table, d) for d in docs])
    m_cpu = time.perf_counter() - t0
    m_ok = bool(np.array_equal(mout.view(np.uint32), mwant.view(np.uint32)))
    m2v.close()
    return {
        "minilm_l6": {"texts_per_batch": 256, "tokens_per_batch": int(ids.size), "gpu_ms_per_batch": gpu_ms,
                      "gpu_texts_per_sec": 256 / (gpu_ms * 1e-3), "gpu_single_text_p50_ms": float(np.median(one)),
                      "max_abs_err_vs_cpu_f32": err, "min_cosine_vs_cpu_f32": cos,
                      "roofline": enc_roofline(bert_flops(np.diff(offs)), gpu_ms),
                      "documents_32x512": {"gpu_ms_per_batch": doc_ms, "tokens": 16384, "roofline": enc_roofline(bert_flops([512] * 32), doc_ms)},
                      "cpu_baseline": {"value": sample / cpu_dt, "unit": "texts/sec", "cores": nthreads, "kind": "port",
                                       "sample": f"{sample} of the same 256 texts, one call, oracle/bert_oracle_c.c (f32, AVX2 FMA, "
                                                 f"{nthreads} threads over ~128-token blocks); the reference's native backend and its "
                                                 "ONNX backend cannot be built here",
                                       "single_text_ms_one_thread": cpu_one * 1e3}},
        "model2vec": {"texts_per_batch": 256, "gpu_ms_per_batch": m_gpu * 1e3, "gpu_texts_per_sec": 256 / m_gpu, "bit_exact_vs_cpu": m_ok,
                      "cpu_baseline": {"value": 256 / m_cpu, "unit": "texts/sec", "cores": 1, "kind": "port",
                                       "sample": "the same 256 texts, one thread, oracle fso_m2v_embed per text"}},
    }


def two_tier_section(quality_index, rows: int, k: int, device, local_rank: int):
    """BASELINE config 3 shape on one GPU: potion fast tier (10M x 256) + MiniLM quality tier (10M x 384), RRF
    with a deterministic stub lexical list (BM25 stays on the CPU in the reference and is not built here).
    Latency of the Initial (phase 0) and Refined (phase 1) deliveries for one caller at a time, and end-to-end
    queries/sec with many concurrent callers."""
    import frankensearch_amd as fa
    from frankensearch_amd.synthetic import random_bert_weights  # seeded synthetic weights (none exist offline)

    fast_dim = 256
    fast_slab = gen_corpus(0, rows, fast_dim, device)
    fast_index = fa.VectorIndex.from_device_slab(fast_slab.data_ptr(), rows, fast_dim, device=local_rank,
                                                 keepalive=fast_slab)
    rng = np.random.default_rng(0)
    table = rng.standard_normal((500_353, fast_dim)).astype(np.float32)   # potion-multilingual-128M shape
    m2v = fa.Model2VecEmbedder(table, device=local_rank)
    bert = fa.NativeEmbedder(random_bert_weights(1, 30522, 384, 6, 1536), device=local_rank)
    # the host side is native: libfshost.so = the reference's SyncTwoTierSearcher flow in C++ over the C ABI, driven by
    # native threads the way a multi-threaded Rust host would drive it (include/fshost.h)
    from frankensearch_amd.host import NativeTwoTierSearcher
    # fast tier through search_top_k_int8_two_pass(query, fetch, 3): the reference's default (two_tier.rs:1318-1337)
    # (quality_int8_latency: a lone caller's quality-tier search goes through the int8 filter + exact re-score — same hits)
    searcher = NativeTwoTierSearcher(fast_index, quality_index, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=3,
                                     quality_int8_latency=True)
    seq_plain = searcher.run_load(threads=1, queries=200, warmup_queries=16, k=k, fast_vocab=500_353, corpus_rows=rows)
    # the reference's phase 2 for an UNATTESTED quality tier — every FSVI v1 pair (sync_searcher.rs:810-818): the fast pool is
    # re-scored on the quality tier (quality_scores_for_hits: a gather of 3k rows) instead of a second scan
    rescored = NativeTwoTierSearcher(fast_index, quality_index, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=3, quality_pool=1)
    seq_resc = rescored.run_load(threads=1, queries=200, warmup_queries=16, k=k, fast_vocab=500_353, corpus_rows=rows)
    rescored_pre = NativeTwoTierSearcher(fast_index, quality_index, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=3, quality_pool=1,
                                         prefetch_quality_embed=True)
    seq_resc_pre = rescored_pre.run_load(threads=1, queries=200, warmup_queries=16, k=k, fast_vocab=500_353, corpus_rows=rows)
    rescored_pre.close()
    # a lone caller: the MiniLM embedding of the query starts with the search and overlaps the fast tier's scan
    prefetching = NativeTwoTierSearcher(fast_index, quality_index, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=3,
                                        prefetch_quality_embed=True, quality_int8_latency=True)
    seq = prefetching.run_load(threads=1, queries=200, warmup_queries=16, k=k, fast_vocab=500_353, corpus_rows=rows)
    prefetching.close()
    # ... and so does the quality tier's search (nothing in it depends on phase 0): phase 0 later, phase 1 earlier
    speculative = NativeTwoTierSearcher(fast_index, quality_index, m2v, bert, doc_id_mode=1, fast_tier_int8_multiplier=3,
                                        prefetch_quality_embed=2, quality_int8_latency=True)
    seq_spec = speculative.run_load(threads=1, queries=200, warmup_queries=16, k=k, fast_vocab=500_353, corpus_rows=rows)
    speculative.close()
    # concurrent callers, coalesced inside the library into batched launches (fsgpu_*_set_coalescing)
    max_batch, wait_us = 256, 1000   # 1,024 callers: 49 k queries/s at 128, 55 k at 256, 54 k at 512 (scripts/exp_two_tier_load.py)
    fast_index.set_coalescing(max_batch, wait_us)
    quality_index.set_coalescing(max_batch, wait_us)
    m2v.set_coalescing(2 * max_batch, wait_us // 2)
    bert.set_coalescing(2 * max_batch, wait_us)

    def concurrent(threads: int, queries: int):
        fb0, fr0 = fast_index.coalescing_stats()
        qb0, qr0 = quality_index.coalescing_stats()
        con = searcher.run_load(threads=threads, queries=queries, warmup_queries=2 * threads, k=k, fast_vocab=500_353,
                                corpus_rows=rows)
        fb, fr = fast_index.coalescing_stats()
        qb, qr = quality_index.coalescing_stats()
        filt = quality_index.batched_filter_stats()
        return {
            "quality_tier_filter": {"int8_active": filt["int8_active"], "int8_queries_so_far": filt["int8_queries"],
                                    "refiltered_on_f16_so_far": filt["refiltered_f16"]},
            "threads": threads, "coalescing": {"max_batch": max_batch, "max_wait_us": wait_us},
            "queries_per_sec": con.queries_per_sec, "completed": con.completed, "failed": con.failed,
            "phase0_p50_ms": con.phase0_p50_ms, "phase0_p95_ms": con.phase0_p95_ms,
            "phase1_p50_ms": con.phase1_p50_ms, "phase1_p95_ms": con.phase1_p95_ms, "phase1_p99_ms": con.phase1_p99_ms,
            "mean_queries_per_scan_batch": {"fast": (fr - fr0) / max(fb - fb0, 1), "quality": (qr - qr0) / max(qb - qb0, 1)},
        }

    # the per-stage coalescers of libfsgpu (round 2-5's route for concurrent callers: four wake-ups per query): kept as a reference point
    stage_coalescers = concurrent(1024, 40_000)
    for h in (fast_index, quality_index, m2v, bert):
        h.set_coalescing(0, 0)
    # Round 6: the many-queries engine of libfshost.  (a) fshost_two_tier_search_many — one call, 64k queries, chunks of 1,024 through
    # the four-stage pipeline; (b) the same engine behind the per-query call (fshost_two_tier_set_batching): concurrent callers are
    # collected into chunks of up to 256 (the whole population below that, half of it above: DESIGN 3.7) and woken once per query.
    searcher.run_load_many(queries=8192, warmup_queries=2048, k=k, fast_vocab=500_353, corpus_rows=rows)
    many = searcher.run_load_many(queries=65_536, warmup_queries=4096, k=k, fast_vocab=500_353, corpus_rows=rows)
    many_rescored = rescored.run_load_many(queries=16_384, warmup_queries=2048, k=k, fast_vocab=500_353, corpus_rows=rows)
    rescored.close()
    batch_chunk, batch_wait_us = 256, 3000
    searcher.set_batching(batch_chunk, batch_wait_us)

    def batched_callers(threads: int, queries: int):
        c0, r0 = searcher.batching_stats()
        con = searcher.run_load(threads=threads, queries=queries, warmup_queries=2 * threads, k=k, fast_vocab=500_353, corpus_rows=rows)
        c1, r1 = searcher.batching_stats()
        return {"threads": threads, "batching": {"max_chunk": batch_chunk, "max_wait_us": batch_wait_us},
                "queries_per_sec": con.queries_per_sec, "completed": con.completed, "failed": con.failed, "first_error": con.first_error,
                "phase0_p50_ms": con.phase0_p50_ms, "phase0_p95_ms": con.phase0_p95_ms,
                "phase1_p50_ms": con.phase1_p50_ms, "phase1_p95_ms": con.phase1_p95_ms, "phase1_p99_ms": con.phase1_p99_ms,
                "mean_queries_per_chunk": (r1 - r0) / max(c1 - c0, 1)}

    con_one = batched_callers(1, 300)
    con_lo = batched_callers(64, 30_000)
    con = batched_callers(256, 60_000)
    con_hi = batched_callers(1024, 100_000)
    searcher.set_batching(0, 0)

    def many_fields(m):
        return {"queries_per_sec": m["queries_per_sec"], "queries": m["queries"], "chunk_queries": m["chunk_queries"], "chunks": m["chunks"],
                "per_chunk_ms": {"fast_embed": m["mean_fast_embed_ms"], "fast_search": m["mean_fast_search_ms"],
                                 "quality_embed": m["mean_quality_embed_ms"], "quality_search": m["mean_quality_search_ms"],
                                 "fusion_busy_summed_over_threads": m["fusion_busy_ms_per_chunk"]},
                "fusion_threads": m["fusion_threads"], "first_chunk_initial_ms": m["first_chunk_initial_ms"],
                "first_chunk_refined_ms": m["first_chunk_refined_ms"], "refinement_failed": m["refinement_failed"],
                "exact_fallbacks": {"fast": m["fast_fallbacks"], "quality": m["quality_fallbacks"]},
                "embeddings_stay_in_device_memory": {"fast": bool(m["device_resident_handoff"] & 1), "quality": bool(m["device_resident_handoff"] & 2)},
                "queries_with_k_initial_and_refined_hits": m["queries_with_k_initial_and_refined_hits"]}
    res = {
        "workload": f"{rows}x256 fast tier (int8 two-pass, multiplier 3) + {rows}x384 quality tier (exact), top-{k}, fetch "
                    f"{3 * k} per tier, stub lexical list of {3 * k}, RRF + blend on the host; per-query C ABI calls from "
                    "native threads (libfshost.so)",
        "phase0_p50_ms": seq.phase0_p50_ms,
        "phase1_p50_ms": seq.phase1_p50_ms,
        "sequential_queries_per_sec": seq.queries_per_sec,
        "sequential_with_quality_search_prefetch": {"phase0_p50_ms": seq_spec.phase0_p50_ms, "phase1_p50_ms": seq_spec.phase1_p50_ms,
                                                    "queries_per_sec": seq_spec.queries_per_sec},
        "sequential_without_quality_embed_prefetch": {"phase0_p50_ms": seq_plain.phase0_p50_ms,
                                                      "phase1_p50_ms": seq_plain.phase1_p50_ms},
        "rescored_fast_pool": {"phase0_p50_ms": seq_resc.phase0_p50_ms, "phase1_p50_ms": seq_resc.phase1_p50_ms,
                               "queries_per_sec": seq_resc.queries_per_sec, "mean_quality_rescore_ms": seq_resc.mean_quality_search_ms,
                               "phase1_p50_ms_with_quality_embed_prefetch": seq_resc_pre.phase1_p50_ms,
                               "note": "SyncQualityPool::RescoredFastPool (unattested / FSVI v1 pairs): quality_scores_for_hits "
                                       "gathers the fast pool's rows on the quality tier, blend_two_tier_aligned"},
        "sequential_breakdown_ms": {"fast_embed": seq.mean_fast_embed_ms, "fast_search": seq.mean_fast_search_ms,
                                    "quality_embed": seq.mean_quality_embed_ms, "quality_search": seq.mean_quality_search_ms,
                                    "fusion": seq.mean_fusion_ms},
        "many_queries_one_call": dict(many_fields(many), note="fshost_two_tier_search_many: Retrieved pool (an independent quality-tier search per "
                                      "query, the headline flow); results = the per-query flow's on the same tier answers (tests/test_gpu_two_tier_many.py)"),
        "many_queries_one_call_rescored_fast_pool": many_fields(many_rescored),
        "concurrent_1_thread_batching_on": con_one,
        "concurrent_64_threads": con_lo,
        "concurrent": con,
        "concurrent_1024_threads": con_hi,
        "concurrent_1024_threads_per_stage_coalescers": stage_coalescers,
    }
    searcher.close()
    fast_index.close()
    del fast_slab
    return res


def host_pointer_section(index, k: int, queries, batch: int):
    """The headline search at the host-pointer C ABI (fsgpu_search_topk_batched: the queries come from host memory and the hits go back
    to it inside every call — the PCIe-inclusive rate; `value` is the device-resident one).  Same batch size, same index, 20 calls."""
    qb = queries[:batch].cpu().numpy()
    index.search_batched(qb, k)
    t0 = time.perf_counter()
    reps, fb = 20, 0
    for _ in range(reps):
        fb += index.search_batched(qb, k)[3]
    dt = time.perf_counter() - t0
    return {"queries_per_sec": reps * batch / dt, "ms_per_step": dt / reps * 1e3, "queries_per_step": batch, "exact_fallback_queries": int(fb),
            "note": "fsgpu_search_topk_batched, one blocking call per step: H2D of the queries, the search, ONE D2H of rows | scores | counts "
                    "through the index's pinned block, host memcpys to the caller's arrays"}


def quantized_section(index, rows: int, dim: int, k: int, queries, bits: int, mult: int):
    """Two-pass searches of the reference on the same corpus: quantised pass-1 scan + exact f16 rescore.
    bits 8 = search_top_k_int8_two_pass (the production fast-tier path, multiplier 3; N*dim bytes per pass),
    bits 4 = search_top_k_4bit_two_pass (multiplier 5 as the reference's bench; N*dim/2 bytes).
    Sequential single queries, host-pointer ABI."""
    fn = index.search_top_k_int8_two_pass if bits == 8 else index.search_top_k_4bit_two_pass
    q = queries[:24].cpu().numpy()
    fn(q[0], k, mult)   # builds the quantised slab (lazy, once)
    index.set_profiling(True)
    index.scan_time(reset=True)
    lat = []
    for i in range(24):
        t0 = time.perf_counter()
        fn(q[i], k, mult)
        lat.append((time.perf_counter() - t0) * 1e3)
    index.set_profiling(False)
    scan_ms, launches, scan_rows = index.scan_stats(reset=True)
    lat = sorted(lat[4:])
    per = scan_ms / max(launches, 1)
    alg = rows * dim * bits // 8
    gbps = alg / (per * 1e-3) / 1e9 if per > 0 else 0.0
    recall = 0
    for i in range(8):
        exact = {h.index for h in index.search_top_k(q[i], k)}
        recall += len(exact & {h.index for h in fn(q[i], k, mult)})
    out = {"candidate_multiplier": mult, "p50_latency_ms": lat[len(lat) // 2], "pass1_kernel_ms": per, "pass1_GBps": gbps,
           "pass1_frac_of_hbm_peak": gbps / HBM_PEAK_GBPS, "algorithmic_bytes": alg,
           "recall_at_k_vs_exact": recall / (8 * k)}
    if bits == 8:
        # the same search for a whole batch: int8 pass 1 on the matrix cores, 128 queries per pass over the int8 slab
        qb = queries[:1024].cpu().numpy()
        index.search_int8_two_pass_batched(qb, k, mult)
        index.scan_stats(reset=True)
        index.set_profiling(True)
        t0 = time.perf_counter()
        reps, fb = 8, 0
        for _ in range(reps):
            fb += index.search_int8_two_pass_batched(qb, k, mult)[3]
        dt = time.perf_counter() - t0
        index.set_profiling(False)
        ms, launches, srows = index.scan_stats(reset=True)
        per_b = ms / max(launches, 1)
        alg_b = srows // max(launches, 1) * dim
        one = index.search_int8_two_pass_batched(qb[:16], k, mult)
        same = all([h.index for h in fn(qb[i], k, mult)] == one[0][i, :one[2][i]].tolist() for i in range(16))
        out["batched"] = {"queries_per_step": len(qb), "queries_per_sec": reps * len(qb) / dt,
                          "pass1_kernel_ms": per_b, "algorithmic_bytes": alg_b,
                          "pass1_GBps": alg_b / (per_b * 1e-3) / 1e9 if per_b > 0 else 0.0,
                          "pass1_frac_of_hbm_peak": alg_b / (per_b * 1e-3) / 1e9 / HBM_PEAK_GBPS if per_b > 0 else 0.0,
                          "per_query_fallbacks": fb, "equals_per_query_search": bool(same),
                          "note": "host-pointer ABI: includes H2D of the queries and D2H of the hits"}
    return out


def mrl_section(index, rows: int, dim: int, k: int, queries):
    """MRL truncated scan (mrl.rs): the kernel reads the first search_dims of every row — N*search_dims*2 bytes."""
    q = queries[:24].cpu().numpy()
    out = {}
    for sd in (128,):
        exact = [set(index.search_batch(q[i], k)[0][0].tolist()) for i in range(8)]
        index.scan_stats(reset=True)
        index.set_profiling(True)
        lat, hit = [], 0
        for i in range(24):
            t0 = time.perf_counter()
            hits = index.mrl_search(q[i], k, search_dims=sd)
            lat.append((time.perf_counter() - t0) * 1e3)
            if i < 8:
                hit += len(exact[i] & {h.index for h in hits})
        index.set_profiling(False)
        ms, launches, _ = index.scan_stats(reset=True)
        per = ms / max(launches, 1)
        alg = rows * sd * 2
        lat = sorted(lat[4:])
        # the same search for a whole batch: truncated scan on the matrix cores (256-384 queries per pass over the prefixes)
        qb = queries[:1024].cpu().numpy()
        index.mrl_search_batched(qb, k, sd)
        index.scan_stats(reset=True)
        index.set_profiling(True)
        t0 = time.perf_counter()
        reps = 6
        for _ in range(reps):
            brows, bscores, bcounts, bfb = index.mrl_search_batched(qb, k, sd)
        dtb = time.perf_counter() - t0
        index.set_profiling(False)
        bms, blaunches, _ = index.scan_stats(reset=True)
        same = all([h.index for h in index.mrl_search(qb[i], k, search_dims=sd)] == brows[i, :bcounts[i]].tolist() for i in range(0, 1024, 97))
        batched = {"queries_per_step": 1024, "queries_per_sec": reps * 1024 / dtb, "pass1_kernel_ms": bms / max(blaunches, 1),
                   "per_query_fallbacks": int(bfb), "equals_per_query_search": bool(same)}
        out[f"search_dims_{sd}"] = {
            "p50_latency_ms": lat[len(lat) // 2], "pass1_kernel_ms": per, "algorithmic_bytes": alg,
            "pass1_GBps": alg / (per * 1e-3) / 1e9 if per > 0 else 0.0,
            "pass1_frac_of_hbm_peak": alg / (per * 1e-3) / 1e9 / HBM_PEAK_GBPS if per > 0 else 0.0,
            "recall_at_k_vs_exact": hit / (8 * k),
            "note": "synthetic corpus is not Matryoshka-trained: recall here only shows the plumbing",
            "batched": batched,
        }
    return out


def config5_section(index, rows: int, k: int, local_rank: int):
    """BASELINE config 5 end to end on one GPU (random-init weights of the MiniLM-L6 shape): see config5_stream_section."""
    import frankensearch_amd as fa
    from frankensearch_amd.synthetic import random_bert_weights
    bert = fa.NativeEmbedder(random_bert_weights(1, 30522, 384, 6, 1536), device=local_rank)
    out = config5_stream_section(fa, index, bert, rows, k)
    bert.close()
    return out


def _shard_devices(n: int, virtual: bool):
    """One shard per GPU — or, for the single-GPU rehearsal of the N-way path, n virtual shards on device 0 (peer copies)."""
    return [0] * n if virtual else list(range(n))


def _sharded_from_generator(fa, devices, rows: int, dim: int, exchange: int, query_groups: int = 1):
    """A sharded handle over the bench corpus: every device's rows are generated in place (contiguous ceil split over the row
    shards; with query groups, device r holds row shard r % (devices / groups) — virtual shards of one device share the tensor)."""
    n = len(devices)
    shards = n // query_groups
    per = (rows + shards - 1) // shards
    slabs, counts, made = [], [], {}
    for r, dev in enumerate(devices):
        s = r % shards
        lo, hi = min(rows, s * per), min(rows, (s + 1) * per)
        if (dev, s) not in made:
            made[(dev, s)] = gen_corpus(lo, hi, dim, torch.device("cuda", dev))
        slabs.append(made[(dev, s)])
        counts.append(hi - lo)
    return fa.NativeShardedIndex.from_device_slabs(devices, dim, counts, [t.data_ptr() for t in slabs], exchange=exchange, keepalive=slabs,
                                                   query_groups=query_groups)


def default_query_groups(n_gpus: int) -> int:
    """Query groups x row shards for an N-GPU run when --query-groups is not given (0 = auto).  A shard's step has a fixed part
    (sample pass, selections, launches) that does not shrink with its rows, and its main pass runs further below the matrix-core
    roof the shorter it is.  One GPU rehearsing ONE rank's share of every layout (profiles/r05/hybrid_layout_sweep.txt; 1,024 queries
    over 10M x 384 per step, ms per step): N = 2: 1 x 2 1.507, 2 x 1 1.447 | N = 4: 1 x 4 0.824, 2 x 2 0.781, 4 x 1 0.925 |
    N = 8: 1 x 8 0.490, 2 x 4 0.444, 4 x 2 0.511, 8 x 1 0.776.  Two query groups win at every N (8 GPUs: 2.31 M queries/s projected
    against 2.09 M row-sharded 8 ways)."""
    return 2 if n_gpus >= 2 and n_gpus % 2 == 0 else 1


def two_tier_sharded_section(fa, devices, quality_index, rows: int, k: int, exchange: int, query_groups: int = 1):
    """BASELINE config 3's flow with BOTH tiers row-sharded over the node: potion fast tier rows x 256 (sharded int8
    two-pass: the corpus-wide candidate set) + MiniLM quality tier rows x 384 (sharded exact search, or the fast pool re-scored by
    a gather routed to the owning shards), encoders on device 0, RRF + blend on the host — libfshost's SyncTwoTierSearcher over two
    fsgpu_sharded handles (fshost_two_tier_create_sharded), per-query C ABI calls from native threads."""
    from frankensearch_amd.host import NativeTwoTierSearcher
    from frankensearch_amd.synthetic import random_bert_weights

    fast_index = _sharded_from_generator(fa, devices, rows, 256, exchange, query_groups)
    table = np.random.default_rng(0).standard_normal((500_353, 256)).astype(np.float32)   # potion-multilingual-128M shape
    m2v = fa.Model2VecEmbedder(table, device=devices[0])
    bert = fa.NativeEmbedder(random_bert_weights(1, 30522, 384, 6, 1536), device=devices[0])
    common = dict(doc_id_mode=1, fast_tier_int8_multiplier=3)
    load = dict(queries=200, warmup_queries=16, k=k, fast_vocab=500_353, corpus_rows=rows)
    # (quality_int8_latency: a lone caller's quality-tier search takes every shard's certified int8 pass — same hits, half the bytes)
    plain = NativeTwoTierSearcher(fast_index, quality_index, m2v, bert, quality_int8_latency=True, **common)
    seq_plain = plain.run_load(threads=1, **load)
    rescored = NativeTwoTierSearcher(fast_index, quality_index, m2v, bert, quality_pool=1, **common)
    seq_resc = rescored.run_load(threads=1, **load)
    rescored.close()
    spec = NativeTwoTierSearcher(fast_index, quality_index, m2v, bert, prefetch_quality_embed=2, quality_int8_latency=True, **common)
    seq_spec = spec.run_load(threads=1, **load)
    spec.close()
    quality_index.set_int8_latency(True)   # (spec's destructor switched it off; `plain` below still wants it)
    # round 6: the many-queries engine over the sharded handles — one call over 32k queries, then 64 / 1,024 per-query callers batched
    # by the engine (fshost_two_tier_set_batching), as in the unsharded section
    plain.run_load_many(queries=4096, warmup_queries=1024, k=k, fast_vocab=500_353, corpus_rows=rows)
    many = plain.run_load_many(queries=32_768, warmup_queries=2048, k=k, fast_vocab=500_353, corpus_rows=rows)
    max_batch, wait_us = 256, 3000
    plain.set_batching(max_batch, wait_us)
    con_lo = plain.run_load(threads=64, queries=20_000, warmup_queries=128, k=k, fast_vocab=500_353, corpus_rows=rows)
    plain.run_load(threads=256, queries=30_000, warmup_queries=512, k=k, fast_vocab=500_353, corpus_rows=rows)
    c0, r0 = plain.batching_stats()
    con = plain.run_load(threads=1024, queries=80_000, warmup_queries=2048, k=k, fast_vocab=500_353, corpus_rows=rows)
    c1, r1 = plain.batching_stats()
    plain.set_batching(0, 0)
    plain.close()
    res = {
        "workload": f"{rows}x256 fast tier (sharded int8 two-pass, multiplier 3) + {rows}x384 quality tier (sharded exact), both row-sharded "
                    f"{len(devices)} way(s), top-{k}, fetch {3 * k} per tier, stub lexical list of {3 * k}, RRF + blend on the host; per-query "
                    "C ABI calls from native threads (libfshost.so over fsgpu_sharded handles)",
        "shards": len(devices),
        "phase0_p50_ms": seq_plain.phase0_p50_ms, "phase1_p50_ms": seq_plain.phase1_p50_ms,
        "sequential_queries_per_sec": seq_plain.queries_per_sec,
        "sequential_breakdown_ms": {"fast_embed": seq_plain.mean_fast_embed_ms, "fast_search": seq_plain.mean_fast_search_ms,
                                    "quality_embed": seq_plain.mean_quality_embed_ms, "quality_search": seq_plain.mean_quality_search_ms,
                                    "fusion": seq_plain.mean_fusion_ms},
        "sequential_with_quality_search_prefetch": {"phase0_p50_ms": seq_spec.phase0_p50_ms, "phase1_p50_ms": seq_spec.phase1_p50_ms},
        "rescored_fast_pool": {"phase0_p50_ms": seq_resc.phase0_p50_ms, "phase1_p50_ms": seq_resc.phase1_p50_ms,
                               "mean_quality_rescore_ms": seq_resc.mean_quality_search_ms},
        "concurrent_1024_threads": {"queries_per_sec": con.queries_per_sec, "completed": con.completed, "failed": con.failed,
                                    "first_error": con.first_error, "phase0_p50_ms": con.phase0_p50_ms, "phase1_p50_ms": con.phase1_p50_ms,
                                    "phase1_p99_ms": con.phase1_p99_ms, "batching": {"max_chunk": max_batch, "max_wait_us": wait_us},
                                    "mean_queries_per_chunk": (r1 - r0) / max(c1 - c0, 1)},
        "concurrent_64_threads": {"queries_per_sec": con_lo.queries_per_sec, "failed": con_lo.failed, "phase0_p50_ms": con_lo.phase0_p50_ms,
                                  "phase1_p50_ms": con_lo.phase1_p50_ms, "phase1_p99_ms": con_lo.phase1_p99_ms},
        "many_queries_one_call": {"queries_per_sec": many["queries_per_sec"], "queries": many["queries"], "chunk_queries": many["chunk_queries"],
                                  "per_chunk_ms": {"fast_embed": many["mean_fast_embed_ms"], "fast_search": many["mean_fast_search_ms"],
                                                   "quality_embed": many["mean_quality_embed_ms"], "quality_search": many["mean_quality_search_ms"]},
                                  "refinement_failed": many["refinement_failed"],
                                  "queries_with_k_initial_and_refined_hits": many["queries_with_k_initial_and_refined_hits"]},
    }
    fast_index.close()
    m2v.close()
    bert.close()
    return res


def config5_batches(n_batches: int, seed: int = 5):
    """SURVEY 8d config 5 queries: token lengths uniform 8..32 incl. [CLS]=101 / [SEP]=102, ids uniform in [1000, 30000); flat ids +
    cumulative offsets over n_batches x 256 texts (the C ABI's argument shape)."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(8, 33, n_batches * 256)
    offs = np.zeros(lens.size + 1, dtype=np.uint32)
    offs[1:] = np.cumsum(lens)
    ids = rng.integers(1000, 30000, int(offs[-1])).astype(np.int32)
    ids[offs[:-1]] = 101
    ids[offs[1:] - 1] = 102
    return ids, offs


def config5_stream_section(fa, index, bert, rows: int, k: int, n_batches: int = 24, encoders=None):
    """BASELINE config 5 end to end through the native pipeline (fshost_embed_search_stream): batches of 256 token-id queries ->
    MiniLM-L6 on the GPU -> batched exact top-k, the encode of group g + 1 running under the search of group g; two encoder batches
    per pass of the slab.  `index` is an fsgpu_index or a row-sharded handle.  encoders (a sharded handle only): one encoder per
    device — the data-parallel form, every encoder embeds its slice of each group on its own device and the shards fetch
    their query group's slice peer to peer."""
    from frankensearch_amd.host import embed_search_stream
    ids, offs = config5_batches(n_batches)
    embed_search_stream(bert, index, ids[:offs[512]], offs[:513], 256, k, group=2, overlap=True, want_hits=False)   # warm-up
    out = {}
    hits = {}
    forms = [("serial_one_scan_per_batch", 1, False, False, bert), ("serial_two_batches_per_scan", 2, False, False, bert),
             ("overlapped_two_batches_per_scan_host_vectors", 2, True, True, bert), ("overlapped_two_batches_per_scan", 2, True, False, bert)]
    if encoders:
        embed_search_stream(list(encoders), index, ids[:offs[512]], offs[:513], 256, k, group=2, overlap=True, want_hits=False)
        forms.append(("overlapped_two_batches_per_scan_data_parallel_encoders", 2, True, False, list(encoders)))
    for name, group, overlap, host, enc in forms:
        r, s, c, st = embed_search_stream(enc, index, ids, offs, 256, k, group=group, overlap=overlap, host_handoff=host)
        hits[name] = (r, s)
        out[name] = {"queries_per_sec": st["queries_per_sec"], "encode_ms_per_group": st["mean_encode_ms"],
                     "search_ms_per_group": st["mean_search_ms"], "groups": int(st["groups"]), "exact_fallbacks": int(st["exact_fallbacks"]),
                     "embeddings_stay_in_device_memory": bool(st["device_resident_handoff"]), "all_counts_full": bool(np.all(c == k)),
                     "encoders": int(st["encoders"])}
    base = hits["serial_one_scan_per_batch"]
    dp_name = "overlapped_two_batches_per_scan_data_parallel_encoders"
    same = all(np.array_equal(h[0], base[0]) and np.array_equal(h[1].view(np.uint32), base[1].view(np.uint32))
               for n, h in hits.items() if n != dp_name)
    if dp_name in hits:
        # an encoder's slice is a batch of its own size: the encoder picks its kernels by batch shape, so the embeddings agree to the
        # encoder's tolerance (cos >= 0.999, 2e-3), not bit for bit — the hits are compared the same way
        d = hits[dp_name]
        out[dp_name]["top1_agreement_with_single_encoder"] = float(np.mean(d[0][:, 0] == base[0][:, 0]))
        out[dp_name]["max_abs_score_difference"] = float(np.max(np.abs(d[1] - base[1])))
    out["workload"] = (f"{n_batches} batches of 256 token-id queries ({int(offs[-1])} tokens) -> MiniLM-L6 on the GPU -> batched exact "
                       f"scan of {rows}x384 f16, top-{k}; native pipeline (libfshost), host-pointer C ABI")
    best = "overlapped_two_batches_per_scan_data_parallel_encoders" if encoders else "overlapped_two_batches_per_scan"
    out["queries_per_sec"] = max(out[best]["queries_per_sec"], out["overlapped_two_batches_per_scan"]["queries_per_sec"])
    out["hits_identical_across_forms"] = bool(same)
    return out


def sharded_handle_main(args) -> None:
    """`bench.py --sharded-handle --gpus N`: ONE process drives N devices through fsgpu_sharded_* (row shards adopted from
    device memory, host-pointer queries, RCCL all-gather + merge inside the library) — the scan, then the metric's second half over
    sharded handles: the two-tier flow (phase 0 + phase 1) and config 5 (on-GPU MiniLM encoding + 50M x 384 scan).  Prints one JSON
    object.  --virtual-shards: N shards on device 0 exchanged by peer copies (the single-GPU rehearsal)."""
    from __graft_entry__ import build
    build()
    import frankensearch_amd as fa

    n = args.gpus
    if not args.virtual_shards and torch.cuda.device_count() < n:
        sys.exit(f"bench.py --sharded-handle --gpus {n}: only {torch.cuda.device_count()} GPU(s) visible")
    devices = _shard_devices(n, args.virtual_shards)
    exchange = fa.NativeShardedIndex.EXCHANGE_PEER_COPY if args.virtual_shards else fa.NativeShardedIndex.EXCHANGE_AUTO
    groups = args.query_groups if args.query_groups > 0 else default_query_groups(n)
    if n % groups:
        sys.exit(f"bench.py --sharded-handle: --query-groups {groups} does not divide --gpus {n}")
    idx = _sharded_from_generator(fa, devices, args.rows, args.dim, exchange, groups)
    B, k = args.batch, args.k
    q = gen_queries(2 * B, args.dim, torch.device("cuda", 0)).cpu().numpy()
    for i in range(max(args.warmup, 2)):
        out = idx.search_batch(q[(i % 2) * B:(i % 2) * B + B], k, batched=args.batched)
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = idx.search_batch(q[(i % 2) * B:(i % 2) * B + B], k, batched=args.batched)
    dt = time.perf_counter() - t0
    # the same steps through begin / end, two searches in flight: the exchange + merge of step i run underneath the scan of step i + 1
    mode = idx.BATCHED if args.batched else idx.EXACT
    pend = idx.search_begin(q[:B], k, mode)
    idx.search_end(pend)
    t1 = time.perf_counter()
    pend = idx.search_begin(q[:B], k, mode)
    for i in range(1, args.steps):
        nxt = idx.search_begin(q[(i % 2) * B:(i % 2) * B + B], k, mode)
        piped = idx.search_end(pend)
        pend = nxt
    piped = idx.search_end(pend)
    dt_piped = time.perf_counter() - t1
    last = q[((args.steps - 1) % 2) * B:((args.steps - 1) % 2) * B + B]
    ref = idx.search_batch(last, k, batched=args.batched)
    piped_same = bool(np.array_equal(piped[0], ref[0]) and np.array_equal(piped[1].view(np.uint32), ref[1].view(np.uint32)))
    # the answer must not depend on the sharding: a few queries against ONE index over all rows when they fit a device comfortably
    same = None
    if args.rows * args.dim * 2 <= 64 << 30 and n > 1:
        whole_slab = gen_corpus(0, args.rows, args.dim, torch.device("cuda", 0))
        whole = fa.VectorIndex.from_device_slab(whole_slab.data_ptr(), args.rows, args.dim, device=0, keepalive=whole_slab)
        wr, ws, _ = whole.search_batch(q[:8], k)
        sr, ss, _ = idx.search_batch(q[:8], k)[:3]
        same = bool(np.array_equal(wr, sr) and np.array_equal(ws.view(np.uint32), ss.view(np.uint32)))
        whole.close()
        del whole_slab
        torch.cuda.empty_cache()
    # ... and against the ORACLE over the whole corpus (the parent run computed its answers on the host and handed them over): exact
    # kernels, the matrix-core batched path and a lone query through the handle
    vs_oracle = None
    if args.expect and os.path.exists(args.expect):
        exp = np.load(args.expect)
        q8 = exp["q"]
        vs_oracle = bool(np.array_equal(q8, q[:8]))
        for mode_ in (idx.EXACT, idx.BATCHED):
            r8, s8, c8, _ = idx.search(q8, k, mode_)
            vs_oracle &= bool(np.array_equal(r8.astype(np.uint32), exp["rows"]) and np.array_equal(s8.view(np.uint32), exp["bits"]))
        r1, s1, c1, _ = idx.search(q8[3], k, idx.EXACT)
        vs_oracle &= bool(np.array_equal(r1[0].astype(np.uint32), exp["rows"][3]) and np.array_equal(s1[0].view(np.uint32), exp["bits"][3]))
    res = {"queries_per_sec": args.steps * B / dt, "ms_per_step": dt / args.steps * 1e3,
           "equals_oracle_8_queries": vs_oracle,
           "pipelined_begin_end": {"queries_per_sec": args.steps * B / dt_piped, "ms_per_step": dt_piped / args.steps * 1e3,
                                   "in_flight": 2, "hits_equal_blocking_call": piped_same}, "n_gpus": n,
           "virtual_shards_on_one_device": bool(args.virtual_shards),
           "layout": f"{groups} query group(s) x {n // groups} row shard(s)",
           "queries_per_step": B, "rows": args.rows, "steps": args.steps,
           "exchange": "rccl ncclAllGather" if idx.exchange_mode() == 1 else "peer copies",
           "path": "matrix-core batched" if args.batched else "exact VALU scan",
           "equals_unsharded_bits": same, "all_counts_full": bool(np.all(out[2] == min(k, args.rows))),
           "note": "host-pointer C ABI: each step includes staging + H2D of the queries and D2H of the hits"}
    if not args.no_two_tier and args.dim == 384:
        try:
            res["two_tier"] = two_tier_sharded_section(fa, devices, idx, args.rows, k, exchange, groups)
        except Exception as e:   # noqa: BLE001 — a failing section must not cost the scan figures above
            res["two_tier"] = {"error": f"{type(e).__name__}: {e}"}
    idx.close()
    del idx
    torch.cuda.empty_cache()
    if not args.no_config5 and args.dim == 384:
        try:
            from frankensearch_amd.synthetic import random_bert_weights
            big = _sharded_from_generator(fa, devices, args.config5_rows, 384, exchange, groups)
            weights = random_bert_weights(1, 30522, 384, 6, 1536)
            bert = fa.NativeEmbedder(weights, device=devices[0])
            # one encoder per device: every device embeds 1/N of each batch and scans its shard — with the encoder on the root only,
            # device 0 does both for the whole batch and is every step's straggler (the rehearsal on one GPU puts them all on device 0)
            encoders = [bert] + [fa.NativeEmbedder(weights, device=d) for d in devices[1:]] if n > 1 else None
            res["config5"] = config5_stream_section(fa, big, bert, args.config5_rows, k, encoders=encoders)
            res["config5"]["shards"] = n
            res["config5"]["layout"] = f"{groups} query group(s) x {n // groups} row shard(s)"
            big.close()
            for e in (encoders or [bert]):
                e.close()
        except Exception as e:   # noqa: BLE001
            res["config5"] = {"error": f"{type(e).__name__}: {e}"}
    print(json.dumps(res), flush=True)
    # (every handle is closed; what is left is interpreter teardown, where the two RCCL copies a process can end up with — torch's
    # bundled one and /opt/rocm's, whichever the loader resolved first for libfsgpu — have been seen to abort in their destructors
    # AFTER the line above: leave without running them)
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(0)


def sharded_handle_leg(args, world: int, virtual: bool):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--sharded-handle", "--gpus", str(world), "--rows", str(args.rows), "--dim",
           str(args.dim), "--k", str(args.k), "--batch", str(args.batch), "--steps", str(min(args.steps, 50)), "--warmup", "3",
           "--config5-rows", str(args.config5_rows), "--query-groups", str(args.query_groups)]
    if getattr(args, "expect", None):
        cmd += ["--expect", args.expect]
    for flag, on in (("--exact", args.exact), ("--virtual-shards", virtual), ("--no-two-tier", args.no_two_tier), ("--no-config5", args.no_config5)):
        if on:
            cmd.append(flag)
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE",
                        "ROLE_RANK", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID", "GROUP_WORLD_SIZE", "ROLE_NAME")}
    limit = 420
    try:
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=limit)
        last = [l for l in res.stdout.splitlines() if l.startswith("{")]
        if last:   # (the line is complete once printed: an abort during teardown does not cost it)
            out = json.loads(last[-1])
            if res.returncode != 0:
                out["child_rc"] = res.returncode
            return out
        return {"error": f"rc={res.returncode}", "stderr_tail": res.stderr[-400:]}
    except subprocess.TimeoutExpired:
        return {"error": f"timed out after {limit} s"}
    except (OSError, ValueError) as e:
        return {"error": str(e)}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None,
                    help="queries per step (default: 1024 on the batched path = 8 passes of 128; 4 on the exact path)")
    ap.add_argument("--blocking-steps", action="store_true",
                    help="N = 1: one blocking fsgpu_search_topk_batched_device call per step instead of begin / end with one step enqueued ahead")
    ap.add_argument("--exact", action="store_true",
                    help="time the exact VALU kernels (4-8 queries per HBM pass) instead of the batched matrix-core path")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--batched", action="store_true", help="(default) the matrix-core batched path; kept for old command lines")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-two-tier", action="store_true")
    ap.add_argument("--no-adversarial", action="store_true", help="skip the uniform-random / outlier-dimension corpus sections")
    ap.add_argument("--no-encoders", action="store_true", help="skip the encoder section (GPU vs CPU baseline)")
    ap.add_argument("--config5", action="store_true",
                    help="also time BASELINE config 5 (batch-256 on-GPU MiniLM encoding + scan) on this rank's rows; "
                         "quoted for --rows 50000000")
    ap.add_argument("--sharded-handle", action="store_true",
                    help="time the in-library sharded handle (fsgpu_sharded_*: ONE process, --gpus devices, RCCL all-gather "
                         "inside libfsgpu.so) instead of the one-process-per-GPU launcher; prints its own JSON line")
    ap.add_argument("--no-sharded-handle", action="store_true", help="skip the sharded-handle leg of an N > 1 run")
    ap.add_argument("--virtual-shards", action="store_true",
                    help="--sharded-handle: --gpus shards on device 0 exchanged by peer copies (single-GPU rehearsal of the N-way handle)")
    ap.add_argument("--no-config5", action="store_true", help="skip config 5 (encode + 50M x 384 scan) in the sharded-handle leg")
    ap.add_argument("--query-groups", type=int, default=0,
                    help="N > 1: query groups G of the hybrid layout (G groups x N/G row shards; every rank scans row shard rank %% (N/G) "
                         "for 1/G of each batch).  0 = auto (default_query_groups), 1 = row shards only")
    ap.add_argument("--config5-rows", type=int, default=50_000_000, help="rows of the config 5 corpus in the sharded-handle leg")
    ap.add_argument("--strong", action="store_true",
                    help="N > 1: the headline step is 1,024 queries for the WHOLE job at every N (\"scaling\": \"strong\") — the same as "
                         "--batch 1024; without it the headline is 1,024 queries per GPU (weak) and the strong form is timed next to it "
                         "and reported as `strong_scaling`")
    ap.add_argument("--no-strong-leg", action="store_true", help="N > 1: skip the second timed loop (the other scaling form)")
    ap.add_argument("--no-merged-check", action="store_true",
                    help="N > 1: skip the check of the merged N-rank answer against the oracle over the whole corpus (7.68 GB of host memory at 10M x 384)")
    ap.add_argument("--expect", type=str, default=None,
                    help="--sharded-handle: an .npz with the oracle's answers (rows, score bits) for the first queries of the pool, written by the "
                         "parent run: the handle's answers are compared with it (equals_oracle_8_queries)")
    args = ap.parse_args()
    if args.strong and args.batch is None:
        args.batch = 1024 if not args.exact else 4
    args.batched = not args.exact
    # Queries per step.  Default: 1,024 PER GPU on the batched path (the launcher branch below multiplies by the world size): every
    # GPU-step is then the same work at every N — 1,024 queries against 10M rows' worth of (rows x queries) — and the line says
    # "scaling": "weak".  An explicit --batch is the WHOLE job's step at any N ("strong").
    args.batch_given = args.batch is not None
    if args.batch is None:
        args.batch = 1024 if args.batched else 4

    if args.sharded_handle:
        return sharded_handle_main(args)
    backend = os.environ.get("FSGPU_BENCH_BACKEND", "nccl")  # "gloo": single-GPU rehearsal of the N>1 path only
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by hand: start the N ranks ourselves (the driver's torch.distributed.run line sets
        # WORLD_SIZE and lands in the branch below)
        ngpu = torch.cuda.device_count()
        if backend == "nccl" and ngpu < args.gpus:
            sys.exit(f"bench.py --gpus {args.gpus}: only {ngpu} GPU(s) visible to this process")
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; the launcher's world size is what runs and what "
              "n_gpus reports", file=sys.stderr)
    ngpu = torch.cuda.device_count()
    if backend == "nccl" and world > ngpu:
        sys.exit(f"bench.py: {world} ranks but only {ngpu} GPU(s) visible (one rank per GPU)")
    if backend != "nccl":
        local_rank = local_rank % max(ngpu, 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from __graft_entry__ import build
    if rank == 0:
        build()
    if world > 1:
        dist.barrier()
    import frankensearch_amd as fa
    from frankensearch_amd.sharded import GpuShardBackend, ShardedVectorIndex, shard_range

    if not args.batch_given and world > 1:
        args.batch *= world
    groups = args.query_groups if args.query_groups > 0 else default_query_groups(world)
    if world % groups:
        sys.exit(f"bench.py: --query-groups {groups} does not divide the {world} ranks")
    row_shards = world // groups
    lo, hi = shard_range(args.rows, rank % row_shards, row_shards)   # rank r holds row shard r % S and serves query group r // S
    slab = gen_corpus(lo, hi, args.dim, device)
    pool = 64
    # (N > 1 times both scaling forms: the pool holds two steps of the larger one)
    other_batch = 0
    if world > 1 and args.batched and not args.no_strong_leg:
        other_batch = 1024 if args.batch != 1024 else 1024 * world
    big = max(args.batch, other_batch)
    queries = gen_queries(max(pool, big) + big, args.dim, device)
    index = fa.VectorIndex.from_device_slab(slab.data_ptr(), hi - lo, args.dim, device=local_rank, row_base=lo,
                                            keepalive=slab)
    index.set_variant(args.variant)
    # N > 1: the all-gather + merge of step i are enqueued on a side stream and run underneath the scan of step i + 1
    sharded = ShardedVectorIndex(GpuShardBackend(index, device, batched=args.batched), overlap=world > 1, query_groups=groups)
    B, k = args.batch, args.k
    B_rank = (B + groups - 1) // groups   # queries this rank scans per step

    shard_backend = sharded.backend
    fallbacks = [0]

    def batch_of(i: int):
        s = (i * B) % (queries.shape[0] - B + 1)
        return queries[s:s + B]

    def run_steps(first: int, n: int, batch_of=batch_of):
        """n whole searches; returns the last step's (rows, scores, counts).  Every step's result is complete when this
        returns (the caller synchronises the device)."""
        out = None
        if world == 1 and args.batched and not args.blocking_steps:
            # one step enqueued ahead (fsgpu_search_topk_batched_device_begin / _end): the GPU starts step i + 1 while the host is
            # still reading step i's verdicts — the same K steps, each complete when this returns (the caller synchronises)
            prev = None
            for i in range(first, first + n):
                cur = shard_backend.scan_begin(batch_of(i), k, packed=False)
                if prev is not None:
                    fallbacks[0] += shard_backend.scan_end(prev[1])
                    out = prev[0]
                prev = cur
            if prev is not None:
                fallbacks[0] += shard_backend.scan_end(prev[1])
                out = prev[0]
            return out
        if world == 1:
            for i in range(first, first + n):
                out = sharded.search(batch_of(i), k)
                if args.batched:
                    fallbacks[0] += shard_backend.last_fallbacks
            return out
        def count_fallbacks():
            if args.batched:
                fallbacks[0] += shard_backend.last_fallbacks
        # this rank's scan of step i; inside it (after its kernels are enqueued, before it blocks) the all-gather + merge of step
        # i - 1 are enqueued on the side stream (ShardedVectorIndex.search_steps)
        return sharded.search_steps(batch_of, first, n, k, after_scan=count_fallbacks)

    run_steps(0, args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    # (every 4th step's main launch is bracketed by HIP events on its stream: a pair idles the stream ~6 us on either side)
    profile_period = 4 if args.steps >= 16 and args.batched and B_rank <= 1024 else 1   # (larger batches take several rounds per call)
    timed_steps = (args.steps + profile_period - 1) // profile_period   # steps whose main launch carries the event pair
    index.set_profiling(profile_period if profile_period > 1 else True)
    fallbacks[0] = 0
    filt0 = index.batched_filter_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run_steps(args.warmup, args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    index.set_profiling(False)
    scan_ms, launches, scan_rows = index.scan_stats(reset=True)
    filt1 = index.batched_filter_stats()
    # which approximate scores filtered the slab in the timed steps: the int8 copy (integer matrix cores, 1 byte per element)
    # or the f16 slab itself; the emitted rows and score bits are the exact search's either way
    i8_queries = filt1["int8_queries"] - filt0["int8_queries"]
    i8_refiltered = filt1["refiltered_f16"] - filt0["refiltered_f16"]
    int8_filter = args.batched and i8_queries * 2 > args.steps * args.batch
    eb = 1 if int8_filter else 2
    mfma_peak = MFMA_I8_PEAK_TOPS if int8_filter else MFMA_F16_PEAK_TFLOPS
    mfma_unit = "TOP/s" if int8_filter else "TFLOP/s"
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- N > 1: what the collective really saw, the other scaling form, and the merged answer against the oracle ---------------------
    multi = None
    if world > 1:
        multi = {}
        # (1) one more all-gather of one word per rank, counted on arrival: the number of DISTINCT ranks whose word came back — not WORLD_SIZE
        tag = torch.tensor([rank + 1], dtype=torch.int64, device=device)
        got = sharded._gather(tag.view(1, 1)).reshape(-1).cpu().tolist()
        multi["ranks_in_collective"] = len({int(x) for x in got if int(x) > 0})
        multi["comm_backend"] = str(dist.get_backend())
        try:
            multi["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else None
        except Exception:   # noqa: BLE001
            multi["rccl_version"] = None
        ident = device_identity(local_rank)
        ident.update({"rank": rank, "local_rank": local_rank, "row_shard": rank % row_shards, "query_group": rank // row_shards,
                      "rows_lo": lo, "rows_hi": hi, "slab_checksum": slab_checksum(slab, lo), "pid": os.getpid()})
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        multi["ranks"] = idents
        # (2) the other scaling form in the same run: 1,024 queries for the whole job (strong) when the headline is weak, and the reverse
        if args.batched and not args.no_strong_leg:
            B2 = other_batch
            fb_keep = fallbacks[0]
            if B2 and B2 != B and queries.shape[0] >= 2 * B2:
                def batch_of2(i: int):
                    s2 = (i * B2) % (queries.shape[0] - B2 + 1)
                    return queries[s2:s2 + B2]
                run_steps(0, max(3, min(args.warmup, 10)), batch_of2)
                torch.cuda.synchronize()
                dist.barrier()
                t2 = time.perf_counter()
                run_steps(3, args.steps, batch_of2)
                torch.cuda.synchronize()
                dist.barrier()
                el2 = torch.tensor([time.perf_counter() - t2], dtype=torch.float64, device=device)
                dist.all_reduce(el2, op=dist.ReduceOp.MAX)
                el2 = float(el2.item())
                fb2, fallbacks[0] = fallbacks[0] - fb_keep, fb_keep
                multi["other_scaling"] = {"scaling": "strong" if B2 == 1024 else "weak", "exact_fallback_queries": fb2, "queries_per_step": B2, "queries_per_step_per_gpu": B2 / world,
                                          "value": args.steps * B2 / el2, "unit": "queries/sec", "ms_per_step": el2 / args.steps * 1e3,
                                          "steps": args.steps, "timed": "barrier + synchronize on both sides, max over ranks, as the headline"}

    # single-query latency through the host-pointer boundary (H2D query, scan, merge, D2H hits, sync)
    lat = []
    if world == 1:
        q1 = queries[:32].cpu().numpy()
        # fsgpu_search_topk as a host calls it: after the batched steps above the index holds the int8 copy of its slab, so a lone
        # query takes ONE certified pass over it + the exact re-score (rows and score bits of the exact kernels); then the exact
        # kernels themselves (fsgpu_search_topk_exact), and both against each other
        for i in range(48):
            t1 = time.perf_counter()
            index.search_batch(q1[i % 32], k)
            lat.append((time.perf_counter() - t1) * 1e3)
        lat = sorted(lat[8:])
        lat_exact, lone_same = [], True
        for i in range(40):
            t1 = time.perf_counter()
            e = index.search_batch(q1[i % 32], k, exact=True)
            lat_exact.append((time.perf_counter() - t1) * 1e3)
            if i < 8:
                r = index.search_batch(q1[i], k)
                lone_same &= bool(np.array_equal(r[0], e[0]) and np.array_equal(r[1].view(np.uint32), e[1].view(np.uint32)))
        lat_exact = sorted(lat_exact[8:])
        # the same boundary with fsgpu_index_set_int8_latency: ONE pass over the int8 copy, the rows within the proven margin re-scored
        # from the f16 slab, the answer certified on the host (rows and score bits of the exact search; the staged path behind it)
        lat_i8, lat_i8_same = [], True
        if args.batched and int8_filter:   # (the batched steps above built the int8 copy and its statistics)
            exact_hits = [index.search_batch(q1[i], k, exact=True) for i in range(8)]
            index.set_int8_latency(True)
            for i in range(72):
                t1 = time.perf_counter()
                r = index.search_batch(q1[i % 32], k)
                lat_i8.append((time.perf_counter() - t1) * 1e3)
                if i < 8:
                    lat_i8_same &= bool(np.array_equal(r[0], exact_hits[i][0]) and
                                        np.array_equal(r[1].view(np.uint32), exact_hits[i][1].view(np.uint32)))
            index.set_int8_latency(False)
            lat_i8 = sorted(lat_i8[8:])

    if rank == 0:
        rows, scores, counts = out
        assert int(counts.min().item()) == min(k, args.rows)
        sc = scores.cpu().numpy()
        assert np.all(np.diff(sc, axis=1) <= 0), "results must be best-first"
        per_launch_ms = scan_ms / max(launches, 1)
        alg_bytes = scan_rows // max(launches, 1) * args.dim * eb   # rows the timed kernel streams per launch x bytes per row
        achieved = alg_bytes / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
        line = {
            "metric": baseline_metric(),
            "value": args.steps * B / elapsed,
            "unit": "queries/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.batch_given else "weak",
            "vs_baseline": None,
            "dtype": "f16 x f32 -> f32" + (" (every emitted score, in the reference's operation order); candidate filter i8 x i8 -> i32" if int8_filter else ""),
            "data": "synthetic",
            "config": {
                "workload": f"{args.rows}x{args.dim} f16 corpus (clustered unit vectors), exact brute-force cosine "
                            f"top-{k}, {B} queries per step, rows sharded {row_shards} way(s)" + (f" x {groups} query groups" if groups > 1 else ""),
                "rows": args.rows, "dim": args.dim, "k": k, "queries_per_step": B, "queries_per_step_per_gpu": B / world,
                "query_groups": groups, "row_shards": row_shards,
                "parallelism": (f"row-shard x{world}" if groups == 1 else f"{groups} query groups x {row_shards} row shards") + ((" + all-gather(top-k) over RCCL" if backend == "nccl" else
                                                          f" + all-gather(top-k) over {backend} (single-GPU rehearsal)") if world > 1 else ""),
                "kernel_variant": args.variant,
                "path": ("matrix-core batched (" + ("int8 slab filter, proven margin" if int8_filter else "f16 filter, proven margin") +
                         ") + exact re-score") if args.batched else "exact VALU scan",
                "filter": ("int8" if int8_filter else "f16") if args.batched else None,
                "filter_refiltered_on_f16_queries": i8_refiltered if int8_filter else None,
                "exact_fallback_queries": fallbacks[0] if args.batched else None,
                "host_loop": ("fsgpu_search_topk_batched_device_begin / _end, one step enqueued ahead" if args.batched and not args.blocking_steps
                              else "one blocking call per step") + "; device-resident queries and hits (the host-pointer ABI's rate: host_pointer_abi)",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": None,
                "kernel": "scan_wide_kernel / scan_mfma_kernel (main pass)" if args.batched else "scan_topk_kernel / scan_mq_topk_kernel",
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_ms": per_launch_ms,
                "launches": launches,
                "timed": f"HIP events on the kernel's stream around the main launch of every {profile_period}. step of the timed region" if profile_period > 1
                         else "HIP events on the kernel's stream around every launch of the timed region",
            },
        }
        if args.batched and launches:
            # the main-pass kernel against BOTH of its roofs: it streams rows * dim * 2 bytes and contracts them with the
            # queries of its launch on the matrix cores; the roof that asks for more time is the one that bounds it
            # (a launch of the register-resident-query kernel takes ALL of a step's 512-query groups — gridDim.y passes over the slab —,
            # so its rows streamed are passes x shard rows while every query still meets every row once)
            q_per_launch = timed_steps * B_rank / launches
            passes_per_launch = (scan_rows / launches) / max(hi - lo, 1)
            flops = 2.0 * (hi - lo) * args.dim * q_per_launch
            tflops = flops / (per_launch_ms * 1e-3) / 1e12 if per_launch_ms > 0 else 0.0
            hbm = dict(line["roofline"])
            if tflops / mfma_peak > achieved / HBM_PEAK_GBPS:
                line["roofline"] = {
                    "bound": "mfma", "achieved": tflops, "peak": mfma_peak, "unit": mfma_unit,
                    "frac": tflops / mfma_peak, "traffic": None,
                    "kernel": "scan_wide_kernel / scan_mfma_kernel (main pass, average over the step's launches)",
                    "algorithmic_flops_per_launch": flops, "queries_per_launch": q_per_launch, "passes_over_the_slab_per_launch": passes_per_launch,
                    "avg_launch_ms": per_launch_ms, "launches": launches, "timed": hbm.get("timed"),
                    "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                            "algorithmic_bytes_per_launch": alg_bytes},
                    "note": ("v_mfma_i32_16x16x64_i8 sustains ~4.2 POP/s on random operands with nothing else in the kernel"
                             if int8_filter else "v_mfma_f32_16x16x32_f16 sustains ~1.85 PFLOP/s on random operands with nothing else in the kernel") +
                            " (profiles/r02/mfma_rate.txt, DESIGN 3.1e: clocks drop under the matrix load), so frac is measured "
                            "against a roof the chip does not reach on real data",
                }
            else:
                line["roofline"]["mfma"] = {"achieved": tflops, "peak": mfma_peak, "unit": mfma_unit, "frac": tflops / mfma_peak}
        # The step against its own two roofs: every query group streams the slab once (HBM) and contracts it with its
        # queries on the matrix cores (2 * rows * dim flops per query); the step cannot beat max(bytes / 8 TB/s, flops / peak)
        if args.batched:
            launches_per_step = launches / max(timed_steps, 1)
            passes = scan_rows / max(timed_steps, 1) / max(hi - lo, 1)   # passes over the slab per step
            t_hbm = launches_per_step * alg_bytes / (HBM_PEAK_GBPS * 1e9)
            t_mfma = 2.0 * (hi - lo) * args.dim * B_rank / (mfma_peak * 1e12)
            bound_s = max(t_hbm, t_mfma)
            line["roofline"]["joint"] = {
                "hbm_ms": t_hbm * 1e3, "mfma_ms": t_mfma * 1e3, "bound_ms": bound_s * 1e3, "bound": "hbm" if t_hbm >= t_mfma else "mfma",
                "frac": bound_s / (elapsed / args.steps), "passes_per_step": passes, "queries_per_pass": B_rank / max(passes, 1e-9),
                "note": "max(filter-slab bytes streamed per step / 8 TB/s, 2*rows*dim*queries ops / " +
                        ("5 POP/s dense int8" if int8_filter else "2.5 PFLOP/s dense f16") + ") / measured step time",
            }
            line["config"]["exact_fallback_rate"] = fallbacks[0] / max(args.steps * B, 1)
        # HBM traffic per launch of the dominant kernel: PMC counters need their own rocprofv3 run, so the figure comes from
        # THIS round's committed summary of that run (scripts/pmc_summary.py) and only when it names the kernel this build
        # launches; a stale or missing summary leaves traffic null
        pmc_path = os.path.join(ROOT, "profiles", PROFILE_ROUND, "pmc_summary.json")
        if world == 1 and args.rows == 10_000_000 and args.dim == 384 and os.path.exists(pmc_path):
            want = fa._lib.lib().fsgpu_last_main_pass_kernel().decode()
            main_only = False
            if want.startswith("scan_wide_kernel"):
                # row bytes and element width identify the pass; a step may mix query-tile counts (same rows, same bytes), and
                # the sample stage is its own instantiation (last template argument 3), not the main pass (0)
                want = ",".join(want.split(",")[:2]) + ","
                main_only = True
            want = want if args.batched else ("scan_mq_topk_kernel<384" if B >= 4 else "scan_topk_kernel<384, 1")
            for e in json.load(open(pmc_path)):
                if main_only and ", 0>(" not in e.get("kernel", ""):
                    continue
                if e.get("counter") == "FETCH_SIZE" and want and want in e.get("kernel", "") and "hbm_read_bytes_corrected" in e:
                    line["roofline"]["traffic"] = e["hbm_read_bytes_corrected"]
                    line["roofline"]["traffic_source"] = (f"profiles/{PROFILE_ROUND}/pmc_summary.json: rocprofv3 --pmc FETCH_SIZE of "
                                                          f"this command for {e['kernel'][:60]}, KiB x 1024 x 2 (gfx950 correction)")
                    break
        if world == 1:
            line["roofline"]["measured_copy_GBps"] = measured_copy_gbps(device)
        if lat:
            line["p50_latency_ms_single_query"] = lat[len(lat) // 2]
            line["p50_latency_ms_single_query_exact_kernels"] = {
                "p50_ms": lat_exact[len(lat_exact) // 2], "default_path_hits_equal_exact_kernels_8_queries": lone_same,
                "note": "fsgpu_search_topk_exact: the exact f16 kernels (one pass over the f16 slab); p50_latency_ms_single_query is "
                        "fsgpu_search_topk, which answers a lone query of an index that holds the int8 copy with the certified pass over it"}
            if lat_i8:
                line["p50_latency_ms_single_query_int8_certified"] = {
                    "p50_ms": lat_i8[len(lat_i8) // 2], "hits_equal_exact_kernels_8_queries": lat_i8_same,
                    "note": "fsgpu_index_set_int8_latency: one pass over the int8 copy + exact re-score of the rows within the proven margin, "
                            "certified (every block's dropped rows lie below the threshold); uncertified queries take the staged path"}
        # the CPU baseline runs before the thousand-thread load test below: after it the container's CPU quota throttles the
        # oracle's workers for a while (measured: 120-150 GB/s instead of ~290 GB/s on the same 16 threads)
        if args.batched and not args.exact:
            # the north-star's HBM target in the same run: the exact f16 kernel, one query per pass over the f16 slab
            # (N > 1: over rank 0's shard — every rank streams its own shard at this rate)
            line["roofline"]["exact_f16_scan"] = exact_scan_roofline(index, queries, k, hi - lo, args.dim, device)
            if world > 1:
                line["roofline"]["exact_f16_scan"]["note"] += f"; measured on rank 0's shard ({hi - lo} of {args.rows} rows)"
            if world == 1 and args.rows == 10_000_000 and args.dim == 384 and os.path.exists(pmc_path):
                # HBM bytes of that kernel from this round's counter pass of `bench.py --exact --batch 1` (the same kernel, slab and launch shape)
                for e in json.load(open(pmc_path)):
                    if (e.get("counter") == "FETCH_SIZE" and e.get("run") == "pmc_fetch_b1" and "scan_topk_kernel<384, 1, 64" in e.get("kernel", "")
                            and "hbm_read_bytes_corrected" in e):
                        line["roofline"]["exact_f16_scan"]["traffic"] = e["hbm_read_bytes_corrected"]
                        line["roofline"]["exact_f16_scan"]["traffic_source"] = (
                            f"profiles/{PROFILE_ROUND}/pmc_summary.json (run pmc_fetch_b1): rocprofv3 --pmc FETCH_SIZE, KiB x 1024 x 2 (gfx950 correction)")
                        break
        if not args.no_cpu_baseline:
            # N > 1: rank 0 times the port on ITS shard (the first rows of the same corpus) and scales to the full corpus
            line["cpu_baseline"] = cpu_baseline_and_parity(slab, queries, k, args.rows, fa.VectorIndex)
            if world > 1:
                # (what that check covers at N > 1: a fresh single-shard index over rank 0's rows — the N-rank answer is checked below)
                cb = line["cpu_baseline"]
                cb["single_shard_parity_bit_exact"] = cb.pop("parity_bit_exact")
                cb["parity_scope"] = f"rank 0's row shard ({hi - lo} rows) searched as a single-shard index; the merged N-rank answer: merged_answer_vs_oracle"
                line["single_shard_parity_bit_exact"] = cb["single_shard_parity_bit_exact"]
        if multi is not None:
            line["ranks_in_collective"] = multi["ranks_in_collective"]
            line["comm_backend"] = multi["comm_backend"]
            line["rccl_version"] = multi["rccl_version"]
            line["ranks"] = multi["ranks"]
            line["device_unique_ids"] = [r.get("unique_id") or r.get("uuid") or r.get("pci") for r in multi["ranks"]]
            if "other_scaling" in multi:
                o = multi["other_scaling"]
                line[o["scaling"] + "_scaling"] = o
                line[o["scaling"] + "_scaling_queries_per_sec"] = o["value"]
            # every row shard's hits must show up in a step's merged answer (1,024+ queries x k hits over evenly spread clusters)
            per_shard = (args.rows + row_shards - 1) // row_shards
            hit_rows = rows.cpu().numpy().astype(np.uint32).astype(np.int64)
            owners = np.unique(hit_rows[hit_rows < args.rows] // per_shard)   # (padding entries are 0xffffffff)
            line["row_shards_with_hits_in_the_merged_answer"] = int(owners.size)
            if not args.no_merged_check:
              try:
                # THE N-rank check: rank 0 rebuilds the whole corpus on the host from its own generator, verifies every rank's slab
                # checksum against it, and compares 8 queries of the LAST TIMED STEP's merged output — the tensors the all-gather +
                # merge left — with the oracle's search over all rows: row ids and f32 score bits
                t_chk = time.perf_counter()
                host = host_corpus(args.rows, args.dim, device)
                slabs_ok = True
                for r in multi["ranks"]:
                    piece = torch.from_numpy(host[r["rows_lo"]:r["rows_hi"]].view(np.int16))
                    slabs_ok &= slab_checksum(piece, r["rows_lo"]) == r["slab_checksum"]
                last_batch = batch_of(args.warmup + args.steps - 1).cpu().numpy()
                per_q = (B + groups - 1) // groups
                picks = sorted({0, 1, per_q - 1, min(per_q, B - 1), min(per_q + 1, B - 1), B // 3, B - 2, B - 1})
                chk = merged_answer_vs_oracle(host, last_batch, picks, rows, scores, counts, k)
                chk["every_rank_slab_checksum_matches_rank0_regeneration"] = bool(slabs_ok)
                chk["what"] = (f"rows and f32 score bits of {len(picks)} queries of the last timed step's merged output ({world} ranks, {groups} query "
                               f"group(s) x {row_shards} row shard(s), all-gather over {multi['comm_backend']}) == oracle.search_top_k over all {args.rows} rows")
                # the oracle's answers for the first pool queries travel to the sharded-handle leg (its own process): --expect
                from oracle import oracle as _oracle
                q8 = queries[:8].cpu().numpy()
                exp_rows = np.full((8, k), 0xFFFFFFFF, dtype=np.uint32)
                exp_bits = np.zeros((8, k), dtype=np.uint32)
                for qi in range(8):
                    er, es = _oracle.search_top_k(host, q8[qi], k, nthreads=chk["oracle_threads"])
                    exp_rows[qi, :len(er)] = er
                    exp_bits[qi, :len(es)] = es.view(np.uint32)
                import tempfile
                fd, args.expect = tempfile.mkstemp(prefix="fsgpu_expect_", suffix=".npz")
                os.close(fd)
                np.savez(args.expect, rows=exp_rows, bits=exp_bits, q=q8)
                del host
                chk["seconds"] = time.perf_counter() - t_chk
                line["merged_answer_vs_oracle"] = chk
                line["merged_answer_equals_oracle_8_queries"] = bool(chk["equal"] and slabs_ok)
              except Exception as e:   # (the check must not cost the measured line; a failed check reads as "not equal")
                line["merged_answer_vs_oracle"] = {"error": f"{type(e).__name__}: {e}"}
                line["merged_answer_equals_oracle_8_queries"] = False
        if world == 1 and not args.no_adversarial and args.rows >= 1_000_000 and args.batched:
            line["adversarial_corpora"] = {kind: adversarial_section(kind, args.rows, args.dim, k, device, local_rank)
                                           for kind in ("uniform", "outlier")}
        if world == 1 and not args.no_encoders:
            line["encoders"] = encoder_section(device, local_rank)
        if world == 1 and args.config5:
            line["config5"] = config5_section(index, args.rows, k, local_rank)
        if world == 1 and args.batched and not args.exact:
            line["host_pointer_abi"] = host_pointer_section(index, k, queries, B)
            line["host_pointer_abi_queries_per_sec"] = line["host_pointer_abi"]["queries_per_sec"]
        if world == 1 and not args.no_two_tier:
            line["int8_two_pass"] = quantized_section(index, args.rows, args.dim, k, queries, 8, 3)
            line["fourbit_two_pass"] = quantized_section(index, args.rows, args.dim, k, queries, 4, 5)
            line["mrl"] = mrl_section(index, args.rows, args.dim, k, queries)
            tt = two_tier_section(index, args.rows, k, device, local_rank)
            line["two_tier"] = tt
            # a lone caller whose host overlaps the quality tier's embedding AND search with phase 0 (fshost_two_tier_config::
            # prefetch_quality_embed = 2: same results, both scans share the GPU); the same caller with only the embedding
            # overlapped, or nothing, is in two_tier.phase1_p50_ms / sequential_without_quality_embed_prefetch
            # headline: the reference-shaped flow — phase 2 starts when phase 1 has been delivered (searcher.rs:1111,2111); the
            # lone caller whose host overlaps the quality tier's embedding (and search) with phase 0 is a sub-field
            plain = tt["sequential_without_quality_embed_prefetch"]
            spec = tt["sequential_with_quality_search_prefetch"]
            line["p50_phase1_latency_ms"] = plain["phase1_p50_ms"]
            line["p50_phase0_latency_ms"] = plain["phase0_p50_ms"]
            line["p50_phase1_latency_policy"] = "reference order: quality-tier embedding and search start after the phase-0 delivery"
            line["p50_phase1_latency_speculative_ms"] = {"quality_embed_prefetched": tt["phase1_p50_ms"],
                                                         "quality_embed_and_search_prefetched": spec["phase1_p50_ms"]}
            # the reference's flow for FSVI v1 pairs (unattested quality tier): phase 2 re-scores the fast pool (a gather)
            line["p50_phase1_latency_rescored_fast_pool_ms"] = tt["rescored_fast_pool"]["phase1_p50_ms"]
            line["end_to_end_queries_per_sec"] = tt["concurrent_1024_threads"]["queries_per_sec"]
            line["end_to_end_many_queries_per_sec"] = tt["many_queries_one_call"]["queries_per_sec"]
            line["end_to_end_policy"] = ("end_to_end_queries_per_sec: 1,024 host threads, one blocking fshost_two_tier_search each, dynamic batching "
                                         "(fshost_two_tier_set_batching); end_to_end_many_queries_per_sec: one fshost_two_tier_search_many call over 65,536 "
                                         "queries; both phase 0 + phase 1, Retrieved pool, fast tier int8 two-pass x3, fetch 30 per tier")
            line["p50_phase1_latency_ms_64_callers"] = tt["concurrent_64_threads"]["phase1_p50_ms"]
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world > 1 and not args.no_sharded_handle:
            # the same search through ONE C-ABI handle (fsgpu_sharded_*: RCCL inside libfsgpu.so) and, over such handles, the
            # metric's second half — the two-tier flow (phase 0 + phase 1) and config 5 —, in a child process with a hard limit
            # once the ranks have left their GPUs: a problem there cannot cost the line above.  Under the gloo rehearsal the
            # child puts its shards on device 0 (peer copies).
            leg = sharded_handle_leg(args, world, virtual=backend != "nccl")
            tt, c5 = leg.pop("two_tier", None), leg.pop("config5", None)
            line["sharded_handle"] = leg
            line["sharded_handle_equals_oracle_8_queries"] = leg.get("equals_oracle_8_queries")
            if tt is not None:
                line["two_tier"] = tt
                if "error" not in tt:
                    line["p50_phase1_latency_ms"] = tt["phase1_p50_ms"]
                    line["p50_phase0_latency_ms"] = tt["phase0_p50_ms"]
                    line["p50_phase1_latency_policy"] = "reference order: quality-tier embedding and search start after the phase-0 delivery"
                    line["p50_phase1_latency_speculative_ms"] = {"quality_embed_and_search_prefetched": tt["sequential_with_quality_search_prefetch"]["phase1_p50_ms"]}
                    line["p50_phase1_latency_rescored_fast_pool_ms"] = tt["rescored_fast_pool"]["phase1_p50_ms"]
                    line["end_to_end_queries_per_sec"] = tt["concurrent_1024_threads"]["queries_per_sec"]
                    if "many_queries_one_call" in tt:
                        line["end_to_end_many_queries_per_sec"] = tt["many_queries_one_call"]["queries_per_sec"]
            if c5 is not None:
                line["config5"] = c5
        if getattr(args, "expect", None) and world > 1:
            try:
                os.unlink(args.expect)
            except OSError:
                pass
        flatten_for_the_driver(line)
        # RCCL prints its version banner through C stdio, which sits in libc's buffer until exit when stdout is a pipe: push
        # it out first so that the JSON line is the last thing on stdout
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
