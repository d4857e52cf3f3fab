#!/usr/bin/env python3
"""Real-weights conformance of the MiniLM encoder: `python scripts/conformance_minilm.py <model_dir>` (needs a GPU).

<model_dir> holds the files the reference ships for all-MiniLM-L6-v2 (crates/frankensearch-embed/src/model_manifest.rs:1053-1062):
`model.safetensors` (or `model_f32.safetensors`), `tokenizer.json` and `config.json`.  The script

  1. tokenises the reference's pinned conformance corpus MODEL_CONFORMANCE_TEXTS_V1 (model_manifest.rs:65-70; the corpus the
     reference's GoldenVectorCertificateV1 is computed over, :308-314) with the Python `tokenizers` package — special tokens added,
     truncated to 512, no padding: what FastEmbed's adapter feeds the model (fastembed_embedder.rs:416-426);
  2. runs `fsgpu_bert_create_safetensors` + `fsgpu_bert_embed` on those token ids (every text alone, and the four as one batch);
  3. runs `transformers.BertModel` in f32 on the CPU over the same ids from the same file, mean-pools over ALL returned tokens and
     L2-normalises (native.rs:1142-1236, fastembed_embedder.rs:416-426);
  4. prints cosine and max-abs difference per text and exits non-zero outside the encoder's stated tolerance
     (cosine >= 0.999, max-abs <= 2e-3: tests/test_gpu_bert.py).

No weights exist in the authoring container or on the GPU boxes (no network): without <model_dir> the script says so and exits 0
with "SKIPPED".  It is the hook a maintainer — or the driver — points at the real files; tests/golden/make_bert_golden.py is the
same comparison on seeded random weights of the same architecture (committed goldens).
"""
from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MODEL_CONFORMANCE_TEXTS_V1 = ["hello world", "semantic search finds related ideas", "identifier fsvi_v2", "naive cafe Tokyo"]
MIN_COSINE, MAX_ABS = 0.999, 2e-3


def make_selftest_dir(path: str) -> None:
    """`--selftest`: a model directory of the real layout with SEEDED RANDOM weights of the MiniLM-L6 architecture and a small WordPiece
    tokenizer — exercises every step of this script where no real weights exist (tests/test_gpu_bert.py runs it on the GPU box)."""
    import torch
    from safetensors.torch import save_file
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors
    from transformers import BertConfig, BertModel

    torch.manual_seed(7)
    words = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list("abcdefghijklmnopqrstuvwxyz0123456789_") + \
            ["##" + c for c in "abcdefghijklmnopqrstuvwxyz0123456789_"] + ["hello", "world", "semantic", "search", "finds", "related", "ideas",
                                                                             "identifier", "naive", "cafe", "tokyo", "fsvi", "##v", "v2"]
    vocab = {w: i for i, w in enumerate(dict.fromkeys(words))}
    tok = Tokenizer(models.WordPiece(vocab, unk_token="[UNK]"))
    tok.normalizer = normalizers.BertNormalizer(lowercase=True)
    tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tok.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", special_tokens=[("[CLS]", vocab["[CLS]"]), ("[SEP]", vocab["[SEP]"])])
    tok.save(os.path.join(path, "tokenizer.json"))
    cfg = BertConfig(vocab_size=len(vocab), hidden_size=384, num_hidden_layers=6, num_attention_heads=12, intermediate_size=1536,
                     max_position_embeddings=512, layer_norm_eps=1e-12)
    model = BertModel(cfg, add_pooling_layer=False).eval()
    state = {k: v.contiguous() for k, v in model.state_dict().items() if "position_ids" not in k}
    save_file(state, os.path.join(path, "model.safetensors"))
    json.dump({"num_attention_heads": 12, "layer_norm_eps": 1e-12, "hidden_act": "gelu"}, open(os.path.join(path, "config.json"), "w"))


def main() -> int:
    if len(sys.argv) >= 2 and sys.argv[1] == "--selftest":
        import tempfile
        with tempfile.TemporaryDirectory(prefix="fsgpu_conformance_") as tmp:
            make_selftest_dir(tmp)
            sys.argv[1] = tmp
            return main()
    if len(sys.argv) < 2 or not os.path.isdir(sys.argv[1]):
        print("SKIPPED: usage: conformance_minilm.py <model_dir> — a directory with model.safetensors + tokenizer.json + config.json "
              "(none exists in this environment: no network, no weights)")
        return 0
    d = sys.argv[1]
    weights = next((os.path.join(d, n) for n in ("model.safetensors", "model_f32.safetensors") if os.path.exists(os.path.join(d, n))), None)
    tok_path = os.path.join(d, "tokenizer.json")
    if not weights or not os.path.exists(tok_path):
        print(f"SKIPPED: {d} lacks model.safetensors / tokenizer.json")
        return 0
    import numpy as np
    import torch
    from tokenizers import Tokenizer
    from transformers import BertConfig, BertModel
    from safetensors.torch import load_file

    tok = Tokenizer.from_file(tok_path)
    tok.no_padding()
    tok.enable_truncation(max_length=512)
    ids = [tok.encode(t, add_special_tokens=True).ids for t in MODEL_CONFORMANCE_TEXTS_V1]
    cfg_path = os.path.join(d, "config.json")
    cfg = json.load(open(cfg_path)) if os.path.exists(cfg_path) else {}
    eps = float(cfg.get("layer_norm_eps", 1e-12))

    # ---- reference arithmetic: transformers f32 on the CPU, the same file
    state = load_file(weights)
    state = {(k[5:] if k.startswith("bert.") else k): v.float() for k, v in state.items()}
    hidden = state["embeddings.word_embeddings.weight"].shape[1]
    layers = 1 + max(int(k.split(".")[2]) for k in state if k.startswith("encoder.layer."))
    inter = state["encoder.layer.0.intermediate.dense.weight"].shape[0]
    hf = BertConfig(vocab_size=state["embeddings.word_embeddings.weight"].shape[0], hidden_size=hidden, num_hidden_layers=layers,
                    num_attention_heads=int(cfg.get("num_attention_heads", hidden // 32)), intermediate_size=inter,
                    max_position_embeddings=state["embeddings.position_embeddings.weight"].shape[0], layer_norm_eps=eps,
                    hidden_act=cfg.get("hidden_act", "gelu"))
    model = BertModel(hf, add_pooling_layer=False).eval()
    missing, unexpected = model.load_state_dict(state, strict=False)
    missing = [m for m in missing if "position_ids" not in m]
    if missing:
        print("FAILED: tensors missing from the file:", missing[:5])
        return 2
    want = []
    with torch.no_grad():
        for t in ids:
            h = model(input_ids=torch.tensor([t])).last_hidden_state[0]
            v = h.mean(dim=0)
            n = float(v.norm())
            want.append((v / n if n > 0 else v).numpy())
    want = np.stack(want)

    # ---- the product: the file as it is through the C ABI
    import frankensearch_amd as fa
    if fa._lib.lib().fsgpu_device_count() < 1:
        print("SKIPPED: no GPU visible (the encoder has no CPU path)")
        return 0
    enc = fa.NativeEmbedder.from_safetensors(weights, device=0, ln_eps=eps)
    alone = np.stack([enc.embed_token_ids(t) for t in ids])
    batch = enc.embed_batch_token_ids(ids)
    ok = True
    for name, got in (("one text per call", alone), ("the four texts as one batch", batch)):
        for i, text in enumerate(MODEL_CONFORMANCE_TEXTS_V1):
            cos = float(np.sum(got[i] * want[i]))
            err = float(np.max(np.abs(got[i] - want[i])))
            good = cos >= MIN_COSINE and err <= MAX_ABS
            ok &= good
            print(f"{name:30s} {text!r:44s} tokens {len(ids[i]):3d}  cosine {cos:.6f}  max-abs {err:.2e}  {'ok' if good else 'OUT OF TOLERANCE'}")
    print(f"hidden {hidden}, layers {layers}, intermediate {inter}, layer_norm_eps {eps:g}; tolerance: cosine >= {MIN_COSINE}, max-abs <= {MAX_ABS}")
    print("PASSED" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
