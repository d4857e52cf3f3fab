"""CPU oracle of the MiniLM-class BERT embedder (numpy f32).  TEST INFRASTRUCTURE ONLY.

Restates `Model::embed_forward` of the reference's native backend
(crates/frankensearch-rerank/src/native.rs:1142-1236) with its helpers:
  encoder_layer_raw   native.rs:587-626   (fused QKV -> per-document attention -> out-proj -> add+LN ->
                                           FFN 4H GELU -> add+LN)
  fused_attention     native.rs:366-432   (per head, no mask, softmax(scale * QK^T) V, scale 1/sqrt(32))
  softmax_row_fused   native.rs:82-147    (exp((x - max) * scale) / sum)
  gelu_scalar         native.rs:190-200   (exact-form GELU with the Abramowitz-Stegun 7.1.26 erf)
  add_ln_raw          native.rs:560-578   (LayerNorm(a + b), eps 1e-12)
  mean pool + L2      native.rs:1209-1235 (mean over ALL returned tokens incl. [CLS]/[SEP]); the final
                                           normalisation uses the adapter's zero guard
                                           (crates/frankensearch-embed/src/fastembed_embedder.rs:416-426:
                                           zeros when norm^2 <= f32::EPSILON).
Weights use the HuggingFace BERT key layout that `parse_weights` normalises (native.rs:1359-1602):
bare `embeddings.*` / `encoder.*` keys get the `bert.` prefix; Q/K/V are stacked to [3H, H].

Parity pin status: the reference's linears are int8 dynamic-quantised (frankentorch, not vendored) and its
other backend is ONNX Runtime (not vendored); no real weights exist here.  This oracle is the f32 form of
that forward and is pinned only against `transformers.BertModel` (tests/golden/make_bert_golden.py, run in
the authoring container) — "parity unpinned" with respect to the Rust binaries.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np

F = np.float32
ATTN_HEAD_DIM = 32


def normalise_keys(weights: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """native.rs:1466-1476: bare sentence-transformers keys -> `bert.`-prefixed scheme."""
    out = {}
    for k, v in weights.items():
        if k.startswith("embeddings.") or k.startswith("encoder."):
            k = "bert." + k
        out[k] = np.ascontiguousarray(v, dtype=F)
    return out


def gelu(x: np.ndarray) -> np.ndarray:
    x = x.astype(F)
    z = x * F(0.70710678118654752440)
    az = np.abs(z)
    t = F(1.0) / (F(1.0) + F(0.3275911) * az)
    poly = t * (F(0.2548296) + t * (F(-0.28449673) + t * (F(1.4214137) + t * (F(-1.453152) + t * F(1.0614054)))))
    erf = np.copysign(F(1.0) - poly * np.exp(-(z * z)).astype(F), z)
    return (F(0.5) * x * (F(1.0) + erf)).astype(F)


def layer_norm(x: np.ndarray, w: np.ndarray, b: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    x = x.astype(F)
    mean = x.mean(axis=-1, keepdims=True, dtype=np.float64)
    var = ((x - mean) ** 2).mean(axis=-1, keepdims=True, dtype=np.float64)
    return (((x - mean) / np.sqrt(var + eps)) * w + b).astype(F)


def attention(qkv: np.ndarray, hidden: int, scale: float) -> np.ndarray:
    """One document: qkv [S, 3H] -> ctx [S, H]."""
    s = qkv.shape[0]
    nh = hidden // ATTN_HEAD_DIM
    q = qkv[:, :hidden].reshape(s, nh, ATTN_HEAD_DIM).transpose(1, 0, 2)
    k = qkv[:, hidden:2 * hidden].reshape(s, nh, ATTN_HEAD_DIM).transpose(1, 0, 2)
    v = qkv[:, 2 * hidden:].reshape(s, nh, ATTN_HEAD_DIM).transpose(1, 0, 2)
    scores = np.matmul(q, k.transpose(0, 2, 1)).astype(F)  # [NH, S, S]
    m = scores.max(axis=-1, keepdims=True)
    e = np.exp(((scores - m) * F(scale)).astype(F)).astype(F)
    p = (e * (F(1.0) / e.sum(axis=-1, keepdims=True, dtype=F))).astype(F)
    ctx = np.matmul(p, v).astype(F)  # [NH, S, HD]
    return ctx.transpose(1, 0, 2).reshape(s, hidden)


def embed_forward(weights: Dict[str, np.ndarray], batch: Sequence[Sequence[int]], num_layers: int,
                  final_zero_guard: bool = True) -> np.ndarray:
    """batch of token-id sequences -> [n_docs, H] unit vectors (zeros for empty inputs)."""
    w = normalise_keys(weights)
    hidden = w["bert.embeddings.word_embeddings.weight"].shape[1]
    scale = F(0.17677669) if ATTN_HEAD_DIM == 32 else F(1.0 / np.sqrt(ATTN_HEAD_DIM))
    lens = [len(ids) for ids in batch]
    out = np.zeros((len(batch), hidden), dtype=F)
    if sum(lens) == 0:
        return out
    ids_flat = np.concatenate([np.asarray(ids, dtype=np.int64) for ids in batch if len(ids)])
    pos_flat = np.concatenate([np.arange(n, dtype=np.int64) for n in lens if n])
    x = (w["bert.embeddings.word_embeddings.weight"][ids_flat]
         + w["bert.embeddings.position_embeddings.weight"][pos_flat]).astype(F)
    x = layer_norm(x + w["bert.embeddings.token_type_embeddings.weight"][0],
                   w["bert.embeddings.LayerNorm.weight"], w["bert.embeddings.LayerNorm.bias"])
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(int)
    for layer in range(num_layers):
        p = f"bert.encoder.layer.{layer}"
        wq = np.concatenate([w[f"{p}.attention.self.query.weight"], w[f"{p}.attention.self.key.weight"],
                             w[f"{p}.attention.self.value.weight"]], axis=0)
        bq = np.concatenate([w[f"{p}.attention.self.query.bias"], w[f"{p}.attention.self.key.bias"],
                             w[f"{p}.attention.self.value.bias"]], axis=0)
        qkv = (x @ wq.T + bq).astype(F)
        ctx = np.zeros_like(x)
        for d, n in enumerate(lens):
            if n:
                a, b = offsets[d], offsets[d + 1]
                ctx[a:b] = attention(qkv[a:b], hidden, scale)
        attn = (ctx @ w[f"{p}.attention.output.dense.weight"].T + w[f"{p}.attention.output.dense.bias"]).astype(F)
        x = layer_norm(x + attn, w[f"{p}.attention.output.LayerNorm.weight"], w[f"{p}.attention.output.LayerNorm.bias"])
        inter = gelu((x @ w[f"{p}.intermediate.dense.weight"].T + w[f"{p}.intermediate.dense.bias"]).astype(F))
        ffn = (inter @ w[f"{p}.output.dense.weight"].T + w[f"{p}.output.dense.bias"]).astype(F)
        x = layer_norm(x + ffn, w[f"{p}.output.LayerNorm.weight"], w[f"{p}.output.LayerNorm.bias"])
    for d, n in enumerate(lens):
        if n == 0:
            continue
        acc = x[offsets[d]:offsets[d + 1]].sum(axis=0, dtype=F) * F(1.0 / n)
        norm_sq = F((acc * acc).sum(dtype=F))
        if final_zero_guard:
            if np.isfinite(norm_sq) and norm_sq > F(1.1920929e-7):
                acc = acc * F(1.0 / np.sqrt(norm_sq))
            else:
                acc = np.zeros_like(acc)
        else:
            norm = np.sqrt(norm_sq)
            if norm > 0:
                acc = acc * F(1.0 / norm)
        out[d] = acc
    return out


def random_weights(seed: int, vocab: int, hidden: int, layers: int, inter: int, max_pos: int = 512,
                   scale: float = 0.05) -> Dict[str, np.ndarray]:
    """Deterministic synthetic weights in the bare sentence-transformers key layout."""
    rng = np.random.default_rng(seed)

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


# ---- the same forward in C (oracle/bert_oracle_c.c): multi-threaded, the encoder cpu_baseline of bench.py -------------------
def _c_weights(weights: Dict[str, np.ndarray], num_layers: int, eps: float = 1e-12):
    """(struct, keepalive) for fso_bert_forward from HF-layout weights."""
    import ctypes as C

    w = normalise_keys(weights)
    fp = C.POINTER(C.c_float)

    class Layer(C.Structure):
        _fields_ = [(n, fp) for n in ("wqkv", "bqkv", "wo", "bo", "ln1_w", "ln1_b", "w1", "b1", "w2", "b2", "ln2_w", "ln2_b")]

    class Weights(C.Structure):
        _fields_ = [("vocab", C.c_int32), ("hidden", C.c_int32), ("layers", C.c_int32), ("inter", C.c_int32),
                    ("max_pos", C.c_int32), ("eps", C.c_float), ("word", fp), ("pos", fp), ("type0", fp),
                    ("emb_ln_w", fp), ("emb_ln_b", fp), ("layer", C.POINTER(Layer))]

    keep = []

    def ptr(a):
        a = np.ascontiguousarray(a, dtype=F)
        keep.append(a)
        return a.ctypes.data_as(fp)

    layers = (Layer * num_layers)()
    for i in range(num_layers):
        p = f"bert.encoder.layer.{i}"
        layers[i].wqkv = ptr(np.concatenate([w[f"{p}.attention.self.{n}.weight"] for n in ("query", "key", "value")], axis=0))
        layers[i].bqkv = ptr(np.concatenate([w[f"{p}.attention.self.{n}.bias"] for n in ("query", "key", "value")], axis=0))
        layers[i].wo, layers[i].bo = ptr(w[f"{p}.attention.output.dense.weight"]), ptr(w[f"{p}.attention.output.dense.bias"])
        layers[i].ln1_w, layers[i].ln1_b = ptr(w[f"{p}.attention.output.LayerNorm.weight"]), ptr(w[f"{p}.attention.output.LayerNorm.bias"])
        layers[i].w1, layers[i].b1 = ptr(w[f"{p}.intermediate.dense.weight"]), ptr(w[f"{p}.intermediate.dense.bias"])
        layers[i].w2, layers[i].b2 = ptr(w[f"{p}.output.dense.weight"]), ptr(w[f"{p}.output.dense.bias"])
        layers[i].ln2_w, layers[i].ln2_b = ptr(w[f"{p}.output.LayerNorm.weight"]), ptr(w[f"{p}.output.LayerNorm.bias"])
    word = w["bert.embeddings.word_embeddings.weight"]
    s = Weights()
    s.vocab, s.hidden, s.layers = word.shape[0], word.shape[1], num_layers
    s.inter = w["bert.encoder.layer.0.intermediate.dense.weight"].shape[0] if num_layers else 0
    s.max_pos = w["bert.embeddings.position_embeddings.weight"].shape[0]
    s.eps = eps
    s.word, s.pos = ptr(word), ptr(w["bert.embeddings.position_embeddings.weight"])
    s.type0 = ptr(w["bert.embeddings.token_type_embeddings.weight"][0])
    s.emb_ln_w, s.emb_ln_b = ptr(w["bert.embeddings.LayerNorm.weight"]), ptr(w["bert.embeddings.LayerNorm.bias"])
    s.layer = layers
    keep.append(layers)
    return s, keep


class CForward:
    """fso_bert_forward bound once to a weight set: `run(batch, nthreads)` -> [n_docs, H] f32."""

    def __init__(self, weights: Dict[str, np.ndarray], num_layers: int):
        import ctypes as C
        from oracle import oracle

        oracle.build()
        self._lib = oracle.lib()
        self._w, self._keep = _c_weights(weights, num_layers)
        self._hidden = int(self._w.hidden)
        self._lib.fso_bert_forward.restype = C.c_int
        self._lib.fso_bert_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]

    def run(self, batch: Sequence[Sequence[int]], nthreads: int = 1) -> np.ndarray:
        import ctypes as C

        lens = [len(b) for b in batch]
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
        ids = np.concatenate([np.asarray(b, dtype=np.int32) for b in batch if len(b)]) if sum(lens) else np.zeros(1, np.int32)
        out = np.zeros((len(batch), self._hidden), dtype=F)
        rc = self._lib.fso_bert_forward(C.byref(self._w), ids.ctypes.data, offsets.ctypes.data, len(batch), nthreads, out.ctypes.data)
        assert rc == 0, rc
        return out


def heavy_tailed_weights(seed: int, vocab: int, hidden: int, layers: int, inter: int, max_pos: int = 512) -> Dict[str, np.ndarray]:
    """Synthetic weights with the statistics trained MiniLM / BERT checkpoints are known for and Gaussian ones lack: a few OUTLIER
    CHANNELS of the residual stream (LayerNorm gains of 8-20 on 4 fixed hidden dimensions, offsets of +-2 there), log-normal
    LayerNorm gains elsewhere, Student-t (3 d.o.f.) weight entries, and down-scaled columns of the linears that read the outlier
    channels — the activations those channels carry are 10-50 x the rest, which is what an f16 operand path has to survive."""
    rng = np.random.default_rng(seed)
    w = random_weights(seed, vocab, hidden, layers, inter, max_pos)
    outl = rng.choice(hidden, 4, replace=False)

    def ln(prefix):
        g = np.exp(0.35 * rng.standard_normal(hidden)).astype(F)
        g[outl] = rng.uniform(8.0, 20.0, 4).astype(F) * rng.choice([-1.0, 1.0], 4).astype(F)
        b = (0.1 * rng.standard_normal(hidden)).astype(F)
        b[outl] = rng.choice([-2.0, 2.0], 4).astype(F)
        w[f"{prefix}.weight"], w[f"{prefix}.bias"] = g, b

    def heavy(name, s):
        shape = w[name].shape
        t = rng.standard_t(3, size=shape).astype(F) * F(s / np.sqrt(3.0))
        t[..., outl] *= F(0.12)     # the columns that read the outlier channels
        w[name] = t

    ln("embeddings.LayerNorm")
    for layer in range(layers):
        p = f"encoder.layer.{layer}"
        for name in ("query", "key", "value"):
            heavy(f"{p}.attention.self.{name}.weight", 0.07)
        heavy(f"{p}.attention.output.dense.weight", 0.05)
        ln(f"{p}.attention.output.LayerNorm")
        heavy(f"{p}.intermediate.dense.weight", 0.05)
        w[f"{p}.output.dense.weight"] = (rng.standard_t(3, size=(hidden, inter)).astype(F) * F(0.04 / np.sqrt(3.0)))
        ln(f"{p}.output.LayerNorm")
    return w
