"""Builds frankensearch_amd/libfsgpu.so for gfx950 with hipcc (in-tree, no JIT cache).

`python -m frankensearch_amd.build` or `frankensearch_amd.build.build()`.  hipcc cross-compiles
without a GPU, so this also runs in the CPU-only authoring container.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libfsgpu.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

SOURCES = ["scan_kernels.hip", "scan_mq_kernel.hip", "int8_kernels.hip", "f32_kernels.hip", "mfma_scan.hip", "mfma_wide.hip", "sort_general.hip", "m2v_kernels.hip", "bert_kernels.hip", "bert_gemm_w.hip", "bert_docs_w.hip", "bert_query_kernels.hip", "bench_fixture.hip", "vector_index.cpp", "vector_index_batched.cpp", "vector_index_lone.cpp", "two_tier_index.cpp", "sharded_index.cpp",
           "bert_embedder.cpp", "safetensors.cpp", "fusion.cpp", "fsgpu_api.cpp"]
HEADERS = ["device_util.hpp", "scan_common.hpp", "kernels.hpp", "vector_index.hpp", "vector_index_internal.hpp", "bert_embedder.hpp", "coalescer.hpp", "sharded_index.hpp", "two_tier_index.hpp", "lab_env.hpp"]
# libfshost.so: the C++ host-side mirror of the reference's two-tier searcher, over the C ABI only (include/fshost.h)
HOST_LIB = os.path.join(HERE, "libfshost.so")
HOST_SOURCES = ["host/two_tier_searcher.cpp", "host/two_tier_many.cpp", "host/load_driver.cpp", "host/stream_pipeline.cpp", "host/fshost_api.cpp"]
HOST_HEADERS = ["host/two_tier_searcher.hpp"]
# -ffp-contract=off: the scan must issue a separate multiply and add (reference order, simd.rs:398-446).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-result", "-x", "hip"]
# mfma_wide.hip: the tile loop of the 640-query int8 shape must unroll completely (its register-resident query fragments
# are indexed by the chunk counter: left rolled they land in scratch), which is past LLVM's default pragma-unroll budget
# bert_kernels.hip / bert_query_kernels.hip: matrix-instruction results stay in ordinary registers (the compiler's default put the
# attention's accumulators in AccVGPRs and moved them out and back around every softmax step: 40 of the loop's 165 instructions)
EXTRA_FLAGS = {"mfma_wide.hip": ["-mllvm", "-pragma-unroll-threshold=200000"],
               "bert_kernels.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "bert_query_kernels.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _digest(paths: list[str], extra: str = "") -> str:
    """sha256 over the CONTENT of the inputs (and the command line that turns them into the target): a copy of the tree to another
    machine — the GPU box — changes every mtime and no content, and must rebuild exactly what its sources no longer match."""
    import hashlib
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode())
        if os.path.exists(p):
            with open(p, "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def _stale(target: str, deps: list[str], extra: str = "") -> bool:
    """The target is missing, or the digest recorded next to it (<target>.sha256) is not that of its inputs."""
    stamp = target + ".sha256"
    if not os.path.exists(target) or not os.path.exists(stamp):
        return True
    try:
        return open(stamp).read().strip() != _digest(deps, extra)
    except OSError:
        return True


def _record(target: str, deps: list[str], extra: str = "") -> None:
    with open(target + ".sha256", "w") as f:
        f.write(_digest(deps, extra))


def _extra_defs() -> list[str]:
    """FSGPU_BUILD_DEFS="-DFSGPU_EXPERIMENTS ..." builds the lab variants (timing skeletons, A/B switches read from the
    environment); the default build ships none of them."""
    return os.environ.get("FSGPU_BUILD_DEFS", "").split()


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    sources = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    # (every object depends on every header: the kernels' argument structs and the C ABI are shared)
    common = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(INCLUDE, "fsgpu.h"), os.path.join(INCLUDE, "fsgpu_lab.h")]
    hipcc = _hipcc()

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        path = os.path.join(CSRC, src)
        flags = FLAGS + EXTRA_FLAGS.get(src, []) + _extra_defs()   # lab definitions are part of the digest: other defs, other object
        if force or _stale(obj, [path] + common, " ".join(flags)):
            cmd = [hipcc] + flags + ["-I", INCLUDE, "-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
            _record(obj, [path] + common, " ".join(flags))
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(sources))) as pool:
        objs = list(pool.map(compile_one, sources))
    stamps = [o + ".sha256" for o in objs]   # the library is its objects: linked again when any object was rebuilt
    if force or _stale(LIB, stamps):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-pthread"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        _record(LIB, stamps)
    build_host(force, verbose)
    return LIB


def build_host(force: bool = False, verbose: bool = False) -> str:
    """g++ only: libfshost.so has no device code; it links libfsgpu.so (found next to it through $ORIGIN)."""
    # (libfshost links libfsgpu by name and calls it through the C ABI only: it depends on the headers, not on the library's bytes)
    deps = [os.path.join(CSRC, s) for s in HOST_SOURCES + HOST_HEADERS] + [os.path.join(INCLUDE, "fshost.h"), os.path.join(INCLUDE, "fsgpu.h")]
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-pthread", "-I", INCLUDE, "-o", HOST_LIB] + \
          [os.path.join(CSRC, s) for s in HOST_SOURCES] + ["-L", HERE, "-lfsgpu", "-Wl,-rpath,$ORIGIN"]
    if force or _stale(HOST_LIB, deps, " ".join(cmd)):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        _record(HOST_LIB, deps, " ".join(cmd))
    return HOST_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
