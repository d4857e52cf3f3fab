// safetensors.cpp — NativeEmbedder::load's weight contract behind the C ABI (fsgpu_bert_create_safetensors): the blob is a
// safetensors file in HuggingFace key layout, parsed the way parse_weights does (crates/frankensearch-rerank/src/native.rs:
// 1359-1602): 8-byte little-endian header length, a JSON header {name: {dtype, shape, data_offsets}}, the tensor bytes; only F32
// tensors are read (I64 position_ids and the like are skipped, :1399-1407), bare `embeddings.*` / `encoder.*` keys count as
// `bert.`-prefixed (:1466-1476), pooler / classifier tensors are ignored.  The shape of the model comes from the tensors:
// vocab x hidden from the word embeddings, layers by counting, inter from intermediate.dense, heads = hidden / 32 (native.rs:36-45).
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <string>
#include <vector>

#include "bert_embedder.hpp"

namespace fsgpu {

namespace {

SearchError load_failed(const std::string& why) {
    SearchError e;
    e.code = FSGPU_ERR_MODEL_LOAD_FAILED;
    e.detail = why;
    return e;
}

// The subset of JSON a safetensors header uses; values are visited, not stored.
struct JsonCursor {
    const char* p;
    const char* end;
    bool ok = true;
    void ws() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
    }
    bool eat(char c) {
        ws();
        if (p < end && *p == c) {
            ++p;
            return true;
        }
        return false;
    }
    bool string(std::string* out) {
        ws();
        if (p >= end || *p != '"') return ok = false;
        ++p;
        out->clear();
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return ok = false;
                switch (*p) {
                    case 'n': out->push_back('\n'); break;
                    case 't': out->push_back('\t'); break;
                    case 'r': out->push_back('\r'); break;
                    case 'b': out->push_back('\b'); break;
                    case 'f': out->push_back('\f'); break;
                    case 'u': {   // tensor names are ASCII; a \uXXXX escape is kept as its low byte
                        if (end - p < 5) return ok = false;
                        unsigned v = 0;
                        for (int i = 1; i <= 4; ++i) {
                            const char c = p[i];
                            v = v * 16 + (c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : 0);
                        }
                        out->push_back((char)(v & 0xff));
                        p += 4;
                        break;
                    }
                    default: out->push_back(*p);
                }
                ++p;
            } else {
                out->push_back(*p++);
            }
        }
        if (p >= end) return ok = false;
        ++p;
        return true;
    }
    bool number(uint64_t* out) {
        ws();
        if (p >= end || *p < '0' || *p > '9') return ok = false;
        uint64_t v = 0;
        while (p < end && *p >= '0' && *p <= '9') v = v * 10 + (uint64_t)(*p++ - '0');
        *out = v;
        return true;
    }
    // skips any value
    bool skip() {
        ws();
        if (p >= end) return ok = false;
        if (*p == '"') {
            std::string s;
            return string(&s);
        }
        if (*p == '{' || *p == '[') {
            const char open = *p, close = open == '{' ? '}' : ']';
            ++p;
            if (eat(close)) return true;
            for (;;) {
                if (open == '{') {
                    std::string k;
                    if (!string(&k) || !eat(':')) return ok = false;
                }
                if (!skip()) return false;
                if (eat(',')) continue;
                if (eat(close)) return true;
                return ok = false;
            }
        }
        while (p < end && *p != ',' && *p != '}' && *p != ']' && *p != ' ' && *p != '\n') ++p;   // number / true / false / null
        return true;
    }
};

struct Tensor {
    const float* data = nullptr;
    std::vector<uint64_t> shape;
    uint64_t count = 0;
};

}  // namespace

SearchError NativeEmbedder::init_safetensors(int device, const void* blob, uint64_t blob_len, float ln_eps) {
    const unsigned char* bytes = static_cast<const unsigned char*>(blob);
    if (!bytes || blob_len < 8) return load_failed("safetensors file too small");
    uint64_t header_len = 0;
    std::memcpy(&header_len, bytes, 8);   // little-endian host
    if (header_len > blob_len - 8) return load_failed("safetensors header length out of range");
    const unsigned char* data = bytes + 8 + header_len;
    const uint64_t data_len = blob_len - 8 - header_len;
    // A tensor that does not start on a 4-byte address in the caller's blob (a header length that is not a multiple of 4: older
    // writers, hand-made files) is copied to an aligned staging area; the reference decodes with f32::from_le_bytes at any
    // alignment (native.rs parse_weights) and loads such files.
    std::deque<std::vector<float>> staged;
    JsonCursor c{reinterpret_cast<const char*>(bytes + 8), reinterpret_cast<const char*>(bytes + 8 + header_len)};
    if (!c.eat('{')) return load_failed("safetensors header is not an object");
    std::map<std::string, Tensor> raw;
    if (!c.eat('}')) {
        for (;;) {
            std::string name;
            if (!c.string(&name) || !c.eat(':')) return load_failed("safetensors header parse: expected a tensor name");
            if (name == "__metadata__") {
                if (!c.skip()) return load_failed("safetensors header parse: bad __metadata__");
            } else {
                if (!c.eat('{')) return load_failed("safetensors header parse: tensor " + name + " is not an object");
                std::string dtype;
                Tensor t;
                uint64_t start = 0, stop = 0;
                bool have_offsets = false;
                if (!c.eat('}')) {
                    for (;;) {
                        std::string key;
                        if (!c.string(&key) || !c.eat(':')) return load_failed("safetensors header parse: tensor " + name);
                        if (key == "dtype") {
                            if (!c.string(&dtype)) return load_failed("safetensors header parse: dtype of " + name);
                        } else if (key == "shape" || key == "data_offsets") {
                            std::vector<uint64_t> v;
                            if (!c.eat('[')) return load_failed("safetensors header parse: " + key + " of " + name);
                            if (!c.eat(']')) {
                                for (;;) {
                                    uint64_t x = 0;
                                    if (!c.number(&x)) return load_failed("safetensors header parse: " + key + " of " + name);
                                    v.push_back(x);
                                    if (c.eat(',')) continue;
                                    if (c.eat(']')) break;
                                    return load_failed("safetensors header parse: " + key + " of " + name);
                                }
                            }
                            if (key == "shape") {
                                t.shape = v;
                            } else {
                                have_offsets = true;
                                start = v.size() > 0 ? v[0] : 0;
                                stop = v.size() > 1 ? v[1] : 0;
                            }
                        } else if (!c.skip()) {
                            return load_failed("safetensors header parse: tensor " + name);
                        }
                        if (c.eat(',')) continue;
                        if (c.eat('}')) break;
                        return load_failed("safetensors header parse: tensor " + name);
                    }
                }
                if (dtype == "F32") {   // (everything else — I64 position_ids, F16 / BF16 exports — is skipped, as the reference does)
                    if (!have_offsets) return load_failed("safetensors tensor " + name + " missing data_offsets");
                    if (start > stop || stop > data_len) return load_failed("safetensors tensor " + name + " has out-of-range offsets");
                    if ((stop - start) & 3u) return load_failed("safetensors tensor " + name + " is not a whole number of f32 values");
                    t.count = (stop - start) / 4;
                    if ((reinterpret_cast<uintptr_t>(data + start) & 3u) != 0) {
                        staged.emplace_back((size_t)t.count);
                        std::memcpy(staged.back().data(), data + start, (size_t)t.count * 4);
                        t.data = staged.back().data();
                    } else {
                        t.data = reinterpret_cast<const float*>(data + start);
                    }
                    const std::string key = (name.rfind("embeddings.", 0) == 0 || name.rfind("encoder.", 0) == 0) ? "bert." + name : name;
                    raw[key] = t;
                }
            }
            if (c.eat(',')) continue;
            if (c.eat('}')) break;
            return load_failed("safetensors header parse: expected ',' or '}'");
        }
    }
    if (raw.empty()) return load_failed("no F32 tensors found in safetensors");
    auto need = [&](const std::string& key, uint64_t rows, uint64_t cols, const float** out) -> SearchError {
        auto it = raw.find(key);
        if (it == raw.end()) return load_failed("missing tensor " + key);
        const uint64_t want = cols ? rows * cols : rows;
        if (it->second.count != want)
            return load_failed("tensor " + key + " holds " + std::to_string(it->second.count) + " values, expected " + std::to_string(want));
        *out = it->second.data;
        return SearchError{};
    };
    auto shape_of = [&](const std::string& key, uint64_t* a, uint64_t* b) -> SearchError {
        auto it = raw.find(key);
        if (it == raw.end()) return load_failed("missing tensor " + key);
        // (both extents fit 32 bits — what fsgpu_bert_config holds — so their product cannot wrap 64)
        if (it->second.shape.size() != 2 || it->second.shape[0] == 0 || it->second.shape[1] == 0 ||
            it->second.shape[0] > UINT32_MAX || it->second.shape[1] > UINT32_MAX ||
            it->second.shape[0] * it->second.shape[1] != it->second.count)
            return load_failed("tensor " + key + " has a bad shape for its " + std::to_string(it->second.count) + " values");
        *a = it->second.shape[0];
        *b = it->second.shape[1];
        return SearchError{};
    };
    uint64_t vocab = 0, hidden = 0, max_pos = 0, ph = 0, inter = 0, ih = 0;
    SearchError e = shape_of("bert.embeddings.word_embeddings.weight", &vocab, &hidden);
    if (!e.ok()) return e;
    e = shape_of("bert.embeddings.position_embeddings.weight", &max_pos, &ph);
    if (!e.ok()) return e;
    uint32_t layers = 0;
    while (raw.count("bert.encoder.layer." + std::to_string(layers) + ".attention.self.query.weight")) ++layers;
    if (layers == 0) return load_failed("no encoder layers (bert.encoder.layer.0.attention.self.query.weight is missing)");
    e = shape_of("bert.encoder.layer.0.intermediate.dense.weight", &inter, &ih);
    if (!e.ok()) return e;
    if (ph != hidden || ih != hidden || hidden % 32 != 0) return load_failed("hidden size must be a multiple of 32 and agree across tensors (native.rs:36-45: 32-wide heads)");
    fsgpu_bert_config cfg{};
    cfg.vocab = (uint32_t)vocab;
    cfg.hidden = (uint32_t)hidden;
    cfg.layers = layers;
    cfg.heads = (uint32_t)(hidden / 32);
    cfg.inter = (uint32_t)inter;
    cfg.max_pos = (uint32_t)(max_pos < 512 ? max_pos : 512);   // DEFAULT_MAX_LENGTH (native.rs:41-51)
    cfg.ln_eps = ln_eps > 0.f ? ln_eps : 1e-12f;
    std::vector<fsgpu_bert_layer_weights> lw(layers);
    fsgpu_bert_weights w{};
    const uint64_t H = hidden, I = inter;
    uint64_t type_rows = 0, th = 0;
    e = shape_of("bert.embeddings.token_type_embeddings.weight", &type_rows, &th);
    if (!e.ok()) return e;
    if (th != hidden) return load_failed("token_type_embeddings has another hidden size");
    struct Want {
        const char* key;
        uint64_t rows, cols;
        const float** dst;
    };
    {
        const Want top[] = {{"bert.embeddings.word_embeddings.weight", vocab, H, &w.word_emb},
                            {"bert.embeddings.position_embeddings.weight", max_pos, H, &w.pos_emb},
                            {"bert.embeddings.token_type_embeddings.weight", type_rows, H, &w.type_emb},
                            {"bert.embeddings.LayerNorm.weight", H, 0, &w.emb_ln_w},
                            {"bert.embeddings.LayerNorm.bias", H, 0, &w.emb_ln_b}};
        for (const Want& t : top) {
            e = need(t.key, t.rows, t.cols, t.dst);
            if (!e.ok()) return e;
        }
    }
    for (uint32_t l = 0; l < layers; ++l) {
        const std::string p = "bert.encoder.layer." + std::to_string(l) + ".";
        fsgpu_bert_layer_weights& x = lw[l];
        const struct {
            const char* suffix;
            uint64_t rows, cols;
            const float** dst;
        } each[] = {{"attention.self.query.weight", H, H, &x.q_w},   {"attention.self.query.bias", H, 0, &x.q_b},
                    {"attention.self.key.weight", H, H, &x.k_w},     {"attention.self.key.bias", H, 0, &x.k_b},
                    {"attention.self.value.weight", H, H, &x.v_w},   {"attention.self.value.bias", H, 0, &x.v_b},
                    {"attention.output.dense.weight", H, H, &x.ao_w}, {"attention.output.dense.bias", H, 0, &x.ao_b},
                    {"attention.output.LayerNorm.weight", H, 0, &x.ln1_w}, {"attention.output.LayerNorm.bias", H, 0, &x.ln1_b},
                    {"intermediate.dense.weight", I, H, &x.i_w},     {"intermediate.dense.bias", I, 0, &x.i_b},
                    {"output.dense.weight", H, I, &x.o_w},           {"output.dense.bias", H, 0, &x.o_b},
                    {"output.LayerNorm.weight", H, 0, &x.ln2_w},     {"output.LayerNorm.bias", H, 0, &x.ln2_b}};
        for (const auto& t : each) {
            e = need(p + t.suffix, t.rows, t.cols, t.dst);
            if (!e.ok()) return e;
        }
    }
    w.layers = lw.data();
    if (device < 0) return SearchError{};   // parse only (callers that validate a blob without a device)
    return init(device, cfg, w);
}

}  // namespace fsgpu
