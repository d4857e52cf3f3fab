"""Deterministic synthetic model weights for benches and tests (there are no real checkpoints offline).

`random_bert_weights` produces a MiniLM-class BERT state dict in the bare sentence-transformers key layout that
`parse_weights` normalises (crates/frankensearch-rerank/src/native.rs:1466-1476)."""
from __future__ import annotations

from typing import Dict

import numpy as np


def random_bert_weights(seed: int, vocab: int, hidden: int, layers: int, inter: int, max_pos: int = 512,
                        scale: float = 0.05) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    F = np.float32

    def t(*shape, s=scale):
        return (rng.standard_normal(shape) * s).astype(F)

    w = {
        "embeddings.word_embeddings.weight": t(vocab, hidden, s=0.5),
        "embeddings.position_embeddings.weight": t(max_pos, hidden, s=0.1),
        "embeddings.token_type_embeddings.weight": t(2, hidden, s=0.1),
        "embeddings.LayerNorm.weight": (1.0 + 0.1 * rng.standard_normal(hidden)).astype(F),
        "embeddings.LayerNorm.bias": t(hidden),
    }
    for layer in range(layers):
        p = f"encoder.layer.{layer}"
        for name in ("query", "key", "value"):
            w[f"{p}.attention.self.{name}.weight"] = t(hidden, hidden, s=0.08)
            w[f"{p}.attention.self.{name}.bias"] = t(hidden)
        w[f"{p}.attention.output.dense.weight"] = t(hidden, hidden)
        w[f"{p}.attention.output.dense.bias"] = t(hidden)
        w[f"{p}.attention.output.LayerNorm.weight"] = (1.0 + 0.1 * rng.standard_normal(hidden)).astype(F)
        w[f"{p}.attention.output.LayerNorm.bias"] = t(hidden)
        w[f"{p}.intermediate.dense.weight"] = t(inter, hidden)
        w[f"{p}.intermediate.dense.bias"] = t(inter)
        w[f"{p}.output.dense.weight"] = t(hidden, inter)
        w[f"{p}.output.dense.bias"] = t(hidden)
        w[f"{p}.output.LayerNorm.weight"] = (1.0 + 0.1 * rng.standard_normal(hidden)).astype(F)
        w[f"{p}.output.LayerNorm.bias"] = t(hidden)
    return w
