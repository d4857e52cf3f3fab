"""Embedders — host-side mirror of the reference's SyncEmbed implementations
(crates/frankensearch-core/src/traits.rs:401-582) over libfsgpu.so.  Tokenisation is the caller's
(third-party `tokenizers` in the reference); the boundary is token ids.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np

from . import _lib
from .errors import check


class Model2VecEmbedder:
    """potion static embedder (crates/frankensearch-embed/src/model2vec_embedder.rs:55-58)."""

    def __init__(self, table: np.ndarray, device: int = 0):
        t = np.ascontiguousarray(table, dtype=np.float32)
        if t.ndim != 2:
            raise TypeError("table must be [vocab, dim] f32")
        self._dim = t.shape[1]
        h = C.c_void_p()
        check(_lib.lib().fsgpu_m2v_create(device, t.ctypes.data, t.shape[0], t.shape[1], C.byref(h)))
        self._h = h

    def dimension(self) -> int:
        return self._dim

    def embed_token_ids(self, ids: Sequence[int]) -> np.ndarray:
        return self.embed_batch_token_ids([ids])[0]

    def embed_batch_token_ids(self, batch: Sequence[Sequence[int]]) -> np.ndarray:
        """embed_batch_sync (model2vec_embedder.rs:409-419) over pre-tokenised texts -> [n, dim] f32."""
        n = len(batch)
        offsets = np.zeros(n + 1, dtype=np.uint32)
        for i, ids in enumerate(batch):
            offsets[i + 1] = offsets[i] + len(ids)
        flat = np.zeros(max(int(offsets[-1]), 1), dtype=np.uint32)
        for i, ids in enumerate(batch):
            flat[offsets[i]:offsets[i + 1]] = np.asarray(ids, dtype=np.uint32)
        return self.embed_flat(flat, offsets)

    def embed_flat(self, ids: np.ndarray, offsets: np.ndarray, out: np.ndarray = None) -> np.ndarray:
        """fsgpu_m2v_embed on arrays already in the C ABI's shape: concatenated uint32 ids, uint32 offsets [n + 1]."""
        if ids.dtype != np.uint32 or offsets.dtype != np.uint32 or not ids.flags.c_contiguous or not offsets.flags.c_contiguous:
            raise TypeError("ids and offsets must be contiguous uint32")
        n = offsets.shape[0] - 1
        if out is None:
            out = np.empty((n, self._dim), dtype=np.float32)
        check(_lib.lib().fsgpu_m2v_embed(self._h, ids.ctypes.data, offsets.ctypes.data, n, out.ctypes.data))
        return out

    def set_coalescing(self, max_batch: int, max_wait_us: int = 100) -> None:
        check(_lib.lib().fsgpu_m2v_set_coalescing(self._h, max_batch, max_wait_us))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().fsgpu_m2v_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _BertConfig(C.Structure):
    _fields_ = [("vocab", C.c_uint32), ("hidden", C.c_uint32), ("layers", C.c_uint32), ("heads", C.c_uint32),
                ("inter", C.c_uint32), ("max_pos", C.c_uint32), ("ln_eps", C.c_float)]


_LAYER_FIELDS = ["q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "ao_w", "ao_b", "ln1_w", "ln1_b", "i_w", "i_b", "o_w", "o_b",
                 "ln2_w", "ln2_b"]


class _BertLayerWeights(C.Structure):
    _fields_ = [(f, C.c_void_p) for f in _LAYER_FIELDS]


class _BertWeights(C.Structure):
    _fields_ = [("word_emb", C.c_void_p), ("pos_emb", C.c_void_p), ("type_emb", C.c_void_p), ("emb_ln_w", C.c_void_p),
                ("emb_ln_b", C.c_void_p), ("layers", C.POINTER(_BertLayerWeights))]


_HF_LAYER_KEYS = {
    "q_w": "attention.self.query.weight", "q_b": "attention.self.query.bias",
    "k_w": "attention.self.key.weight", "k_b": "attention.self.key.bias",
    "v_w": "attention.self.value.weight", "v_b": "attention.self.value.bias",
    "ao_w": "attention.output.dense.weight", "ao_b": "attention.output.dense.bias",
    "ln1_w": "attention.output.LayerNorm.weight", "ln1_b": "attention.output.LayerNorm.bias",
    "i_w": "intermediate.dense.weight", "i_b": "intermediate.dense.bias",
    "o_w": "output.dense.weight", "o_b": "output.dense.bias",
    "ln2_w": "output.LayerNorm.weight", "ln2_b": "output.LayerNorm.bias",
}


class NativeEmbedder:
    """MiniLM-class BERT embedder (crates/frankensearch-rerank/src/native_embedder.rs:40-50).

    `weights` is a dict of f32 arrays in the HuggingFace key layout; bare `embeddings.*` / `encoder.*` keys are
    normalised to the `bert.` prefix exactly like `parse_weights` (native.rs:1466-1476)."""

    def __init__(self, weights: dict, device: int = 0, ln_eps: float = 1e-12):
        w = {}
        for k, v in weights.items():
            if k.startswith("embeddings.") or k.startswith("encoder."):
                k = "bert." + k
            w[k] = np.ascontiguousarray(v, dtype=np.float32)
        word = w["bert.embeddings.word_embeddings.weight"]
        pos = w["bert.embeddings.position_embeddings.weight"]
        layers = 0
        while f"bert.encoder.layer.{layers}.attention.self.query.weight" in w:
            layers += 1
        hidden = word.shape[1]
        inter = w["bert.encoder.layer.0.intermediate.dense.weight"].shape[0]
        cfg = _BertConfig(word.shape[0], hidden, layers, hidden // 32, inter, min(pos.shape[0], 512), ln_eps)
        lw = (_BertLayerWeights * layers)()
        for i in range(layers):
            for f, key in _HF_LAYER_KEYS.items():
                setattr(lw[i], f, w[f"bert.encoder.layer.{i}.{key}"].ctypes.data)
        bw = _BertWeights(word.ctypes.data, pos.ctypes.data,
                          w["bert.embeddings.token_type_embeddings.weight"].ctypes.data,
                          w["bert.embeddings.LayerNorm.weight"].ctypes.data,
                          w["bert.embeddings.LayerNorm.bias"].ctypes.data, lw)
        self._dim = hidden
        h = C.c_void_p()
        check(_lib.lib().fsgpu_bert_create(device, C.byref(cfg), C.byref(bw), C.byref(h)))
        self._h = h

    @classmethod
    def from_safetensors(cls, path: str, device: int = 0, ln_eps: float = 1e-12) -> "NativeEmbedder":
        """NativeEmbedder::load (native_embedder.rs:60-116): the model file goes to the library as it is — the safetensors header and
        the HuggingFace key layout are parsed behind the C ABI (fsgpu_bert_create_safetensors = parse_weights, native.rs:1359-1602)."""
        with open(path, "rb") as f:
            return cls.from_safetensors_bytes(f.read(), device=device, ln_eps=ln_eps)

    @classmethod
    def from_safetensors_bytes(cls, blob: bytes, device: int = 0, ln_eps: float = 1e-12) -> "NativeEmbedder":
        buf = np.frombuffer(blob, dtype=np.uint8)   # (numpy's buffer of a bytes object is 16-byte aligned past its header: checked by the library)
        if buf.ctypes.data % 8:
            buf = np.require(buf.copy(), requirements=["ALIGNED"])
        self = cls.__new__(cls)
        h = C.c_void_p()
        check(_lib.lib().fsgpu_bert_create_safetensors(device, buf.ctypes.data, buf.size, ln_eps, C.byref(h)))
        self._h = h
        self._dim = int(_lib.lib().fsgpu_bert_dimension(h))
        return self

    def dimension(self) -> int:
        return self._dim

    def embed_token_ids(self, ids: Sequence[int]) -> np.ndarray:
        return self.embed_batch_token_ids([ids])[0]

    def embed_batch_token_ids(self, batch: Sequence[Sequence[int]]) -> np.ndarray:
        """embed_batch_sync (native_embedder.rs:218-255) over pre-tokenised texts (special tokens included)."""
        n = len(batch)
        offsets = np.zeros(n + 1, dtype=np.uint32)
        for i, ids in enumerate(batch):
            offsets[i + 1] = offsets[i] + len(ids)
        flat = np.zeros(max(int(offsets[-1]), 1), dtype=np.int32)
        for i, ids in enumerate(batch):
            flat[offsets[i]:offsets[i + 1]] = np.asarray(ids, dtype=np.int32)
        return self.embed_flat(flat, offsets)

    def embed_flat(self, ids: np.ndarray, offsets: np.ndarray, out: np.ndarray = None) -> np.ndarray:
        """The C ABI as a host holds it after tokenisation (fsgpu_bert_embed): concatenated int32 ids, uint32 offsets
        [n + 1]; text i owns ids[offsets[i]:offsets[i + 1]].  No per-text Python work (the list marshalling of
        embed_batch_token_ids costs ~0.3 ms per 256 texts, more than half the GPU forward)."""
        if ids.dtype != np.int32 or offsets.dtype != np.uint32 or not ids.flags.c_contiguous or not offsets.flags.c_contiguous:
            raise TypeError("ids must be contiguous int32 and offsets contiguous uint32")
        n = offsets.shape[0] - 1
        if out is None:
            out = np.empty((n, self._dim), dtype=np.float32)
        check(_lib.lib().fsgpu_bert_embed(self._h, ids.ctypes.data, offsets.ctypes.data, n, out.ctypes.data))
        return out

    def set_coalescing(self, max_batch: int, max_wait_us: int = 200) -> None:
        check(_lib.lib().fsgpu_bert_set_coalescing(self._h, max_batch, max_wait_us))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().fsgpu_bert_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
