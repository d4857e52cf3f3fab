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
        out = np.empty((n, self._dim), dtype=np.float32)
        check(_lib.lib().fsgpu_m2v_embed(self._h, flat.ctypes.data, offsets.ctypes.data, n, out.ctypes.data))
        return out

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().fsgpu_m2v_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
