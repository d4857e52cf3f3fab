"""Generates tests/golden/bert_golden.npz in the AUTHORING container (needs `transformers` + torch CPU).

A `transformers.BertModel` (the architecture all-MiniLM-L6-v2 uses: post-LN BERT, exact-erf GELU, eps 1e-12,
no pooler) is given the oracle's seeded synthetic weights; its mean-pooled, L2-normalised outputs for a few
token-id sequences are stored as the golden vectors.  The fixture holds only data (config numbers, seeds,
token ids, expected outputs) — the weights are regenerated from the seed by oracle/bert_oracle.random_weights.
"""
import os
import sys

import numpy as np
import torch
from transformers import BertConfig, BertModel

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bert_oracle  # noqa: E402

CASES = {
    # name: (seed, vocab, hidden, layers, inter)
    "tiny": (11, 200, 128, 2, 512),
    "minilm_shape": (12, 1000, 384, 6, 1536),
}
BATCH = [
    [101, 7, 8, 9, 102],
    [101, 55, 102],
    [101] + list(range(3, 40)) + [102],
    [101, 150, 151, 152, 153, 154, 155, 156, 102],
    [5],
]


def run(seed, vocab, hidden, layers, inter):
    w = bert_oracle.random_weights(seed, vocab, hidden, layers, inter)
    cfg = BertConfig(vocab_size=vocab, hidden_size=hidden, num_hidden_layers=layers,
                     num_attention_heads=hidden // 32, intermediate_size=inter, max_position_embeddings=512,
                     layer_norm_eps=1e-12, hidden_act="gelu", hidden_dropout_prob=0.0,
                     attention_probs_dropout_prob=0.0)
    model = BertModel(cfg, add_pooling_layer=False).eval()
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in m or "token_type_ids" in m for m in missing), missing
    outs = []
    with torch.no_grad():
        for ids in BATCH:  # one sequence at a time: no padding, every returned token pooled (native_embedder.rs:123-138)
            h = model(input_ids=torch.tensor([ids])).last_hidden_state[0]
            v = h.mean(dim=0)
            outs.append((v / v.norm()).numpy())
    return np.stack(outs).astype(np.float32)


if __name__ == "__main__":
    data = {"batch_lens": np.array([len(b) for b in BATCH]), "batch_ids": np.concatenate([np.array(b) for b in BATCH])}
    for name, cfg in CASES.items():
        data[f"{name}_config"] = np.array(cfg)
        data[f"{name}_expected"] = run(*cfg)
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "bert_golden.npz"), **data)
    print("wrote bert_golden.npz", {k: v.shape for k, v in data.items()})
