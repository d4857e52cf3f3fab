"""CPU-side checks of the C-ABI boundary: the library builds/loads and exports every symbol that
include/fsgpu.h declares; with no GPU, compute entry points fail loudly (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(name="fsgpu.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fsgpu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from frankensearch_amd import _lib
    from frankensearch_amd.build import build

    build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = header_symbols()
    lab = header_symbols("fsgpu_lab.h")   # bench fixtures, kernel timers, A/B switches: exported, but not part of the drop-in surface
    assert len(names) >= 25
    for name in names + lab:
        assert hasattr(L, name), f"{name} declared in include/ but not exported"
    assert not set(names) & set(lab), "an entry point is declared in both headers"
    assert {"fsgpu_index_set_variant", "fsgpu_bench_fixture_device", "fsgpu_last_main_pass_kernel", "fsgpu_index_scan_time"} <= set(lab)
    assert set(names) | set(lab) == set(_lib.SIGNATURES), "python binding and headers disagree"
    assert b"gfx950" in _lib.lib().fsgpu_version()


def test_host_library_exports_every_declared_symbol_and_has_no_device_code():
    # libfshost.so (include/fshost.h): the C++ host-side mirror; it must export its ABI, link libfsgpu.so and carry no
    # GPU code object of its own
    import ctypes
    import subprocess
    from frankensearch_amd import host
    from frankensearch_amd.build import build

    build()
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "fshost.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(fshost_[a-z0-9_]+)\s*\(", text)))
    assert set(names) == set(host.SYMBOLS)
    L = host.lib()
    for name in names:
        assert hasattr(L, name), f"{name} declared in include/fshost.h but not exported"
    needed = subprocess.check_output(["readelf", "-d", host.LIB_PATH]).decode()
    assert "libfsgpu.so" in needed
    assert "libamdhip64" not in needed, "the host mirror must reach the GPU only through the fsgpu C ABI"


def test_scan_kernel_has_no_fused_multiply_add():
    """The scan must issue separate v_mul/v_add (reference order, simd.rs:398-446): check the ISA."""
    import subprocess
    from frankensearch_amd import _lib
    from frankensearch_amd.build import build

    build()
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    import glob
    import shutil
    import tempfile
    obj = os.path.join(os.path.dirname(_lib.LIB_PATH), "_build", "scan_kernels.o")
    tmp = tempfile.mkdtemp(prefix="fsgpu_isa_")
    shutil.copy(obj, os.path.join(tmp, "scan_kernels.o"))
    subprocess.check_call([objdump, "--offloading", "scan_kernels.o"], cwd=tmp, stdout=subprocess.DEVNULL)
    outs = glob.glob(os.path.join(tmp, "*gfx950*"))
    assert outs, "no gfx950 code object in scan_kernels.o"
    out = outs[0]
    asm = subprocess.check_output([objdump, "-d", out]).decode()
    # split per symbol and look only at the fused scan kernels with a compile-time dim
    blocks = re.split(r"\n(?=[0-9a-f]+ <)", asm)
    seen = 0
    for b in blocks:
        head = b.split("\n", 1)[0]
        if "scan_topk_kernelILi384ELi1E" in head or "scan_topk_kernelILi256ELi1E" in head:
            seen += 1
            assert "v_cvt_f32_f16" in b and "v_pk_mul_f32" in b and "v_pk_add_f32" in b
            assert not re.search(r"v_(pk_)?fma_f32|v_fmac_f32|v_mac_f32|v_mad_f32|v_dot2", b), head
    assert seen >= 2


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import frankensearch_amd as fa
    with pytest.raises(fa.NoDevice):
        fa.VectorIndex.from_slab(np.zeros((4, 8), np.uint16))
    with pytest.raises(fa.NoDevice):
        fa.Model2VecEmbedder(np.ones((4, 8), np.float32))
    with pytest.raises(fa.NoDevice):
        fa.encode_f32_to_f16(np.ones(4, np.float32))


def test_product_never_imports_oracle():
    """The shipped package must not reference the oracle (test infrastructure only)."""
    pkg = os.path.join(ROOT, "frankensearch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "fs_oracle" not in text and "fso_" not in text, f
                if f.endswith(".py"):
                    assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f


def _safetensors_blob(tensors, metadata=None):
    """A safetensors file image: 8-byte little-endian header length, JSON header (padded with spaces to 8 bytes), tensor bytes."""
    import json
    import struct
    header, data, off = {}, b"", 0
    if metadata:
        header["__metadata__"] = metadata
    for name, arr in tensors.items():
        raw = np.ascontiguousarray(arr).tobytes()
        dtype = {"float32": "F32", "int64": "I64", "float16": "F16"}[str(arr.dtype)]
        header[name] = {"dtype": dtype, "shape": list(arr.shape), "data_offsets": [off, off + len(raw)]}
        data += raw
        off += len(raw)
    h = json.dumps(header, separators=(",", ":")).encode()
    h += b" " * ((8 - len(h) % 8) % 8)
    return struct.pack("<Q", len(h)) + h + data


def _tiny_bert_tensors(prefix="", hidden=64, layers=2, inter=128, vocab=50, max_pos=40):
    rng = np.random.default_rng(0)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    t = {f"{prefix}embeddings.word_embeddings.weight": f(vocab, hidden), f"{prefix}embeddings.position_embeddings.weight": f(max_pos, hidden),
         f"{prefix}embeddings.token_type_embeddings.weight": f(2, hidden), f"{prefix}embeddings.LayerNorm.weight": f(hidden),
         f"{prefix}embeddings.LayerNorm.bias": f(hidden), f"{prefix}embeddings.position_ids": np.arange(max_pos, dtype=np.int64)[None, :]}
    for l in range(layers):
        p = f"{prefix}encoder.layer.{l}."
        for nm in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
            t[p + nm + ".weight"], t[p + nm + ".bias"] = f(hidden, hidden), f(hidden)
        t[p + "intermediate.dense.weight"], t[p + "intermediate.dense.bias"] = f(inter, hidden), f(inter)
        t[p + "output.dense.weight"], t[p + "output.dense.bias"] = f(hidden, inter), f(hidden)
        for nm in ("attention.output.LayerNorm", "output.LayerNorm"):
            t[p + nm + ".weight"], t[p + nm + ".bias"] = f(hidden), f(hidden)
    return t


def test_safetensors_blob_is_validated_before_a_device_is_needed():
    """fsgpu_bert_create_safetensors = NativeEmbedder::load -> parse_weights (native.rs:1359-1602): a malformed model file is a
    ModelLoadFailed on any host (the blob is parsed first); a well-formed one then asks for the GPU (NoDevice here)."""
    import torch
    import frankensearch_amd as fa

    good = _safetensors_blob(_tiny_bert_tensors(), metadata={"format": "pt"})
    good_prefixed = _safetensors_blob(_tiny_bert_tensors("bert."))
    hlen = int.from_bytes(good[:8], "little")
    header = good[8:8 + hlen].rstrip(b" ")
    header = header[:-1] + b" " * ((1 - len(header)) % 4) + b"}"
    unaligned = len(header).to_bytes(8, "little") + header + good[8 + hlen:]   # header length = 1 (mod 4): tensor bytes at odd addresses
    assert len(header) % 4 == 1
    if not torch.cuda.is_available():
        for blob in (good, good_prefixed, unaligned):
            with pytest.raises(fa.NoDevice):
                fa.NativeEmbedder.from_safetensors_bytes(blob)
    cases = {
        "too small": b"\x01\x02",
        "header length out of range": (1 << 40).to_bytes(8, "little") + b"{}",
        "not an object": (2).to_bytes(8, "little") + b"[]",
        "no F32 tensors": _safetensors_blob({"embeddings.position_ids": np.arange(4, dtype=np.int64)}),
        "missing tensor": _safetensors_blob({k: v for k, v in _tiny_bert_tensors().items() if "layer.1.output.dense.bias" not in k}),
        "bad shape": _safetensors_blob({**_tiny_bert_tensors(), "encoder.layer.0.attention.self.key.weight": np.zeros((64, 32), np.float32)}),
        "hidden not a multiple of 32": _safetensors_blob(_tiny_bert_tensors(hidden=48)),
    }
    for why, blob in cases.items():
        with pytest.raises(fa.ModelLoadFailed) as err:
            fa.NativeEmbedder.from_safetensors_bytes(blob)
        assert str(err.value), why
    # out-of-range data offsets
    t = _tiny_bert_tensors()
    blob = bytearray(_safetensors_blob(t))
    with pytest.raises(fa.ModelLoadFailed):
        fa.NativeEmbedder.from_safetensors_bytes(bytes(blob[:len(blob) - 64]))


def test_integration_document_names_every_entry_point_of_the_product_headers():
    """INTEGRATION.md shows the reference-side binding of the drop-in surface: every function include/fsgpu.h and include/fshost.h
    declare appears in it (a new export without its binding line fails here)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = []
    for header, prefix in (("fsgpu.h", "fsgpu_"), ("fshost.h", "fshost_")):
        text = open(os.path.join(root, "include", header)).read()
        for name in sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text))):
            if name not in doc and name != "fsgpu_status":   # (the return type, in a function-pointer typedef)
                missing.append(name)
    assert missing == [], missing
