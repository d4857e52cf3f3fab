// bert_embedder.hpp — host side of the GPU MiniLM-class embedder; mirrors the reference's NativeEmbedder
// (crates/frankensearch-rerank/src/native_embedder.rs:40-50,173-255: embed_sync / embed_batch_sync over token
// ids) and the weight contract of parse_weights (crates/frankensearch-rerank/src/native.rs:1359-1602).
#pragma once

#include <cstdint>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "../../include/fsgpu.h"
#include "vector_index.hpp"

namespace fsgpu {

class NativeEmbedder {
  public:
    ~NativeEmbedder();
    SearchError init(int device, const fsgpu_bert_config& cfg, const fsgpu_bert_weights& w);
    // NativeEmbedder::load's weight contract: a safetensors blob in HuggingFace key layout (safetensors.cpp; parse_weights,
    // crates/frankensearch-rerank/src/native.rs:1359-1602).  device < 0 validates the blob only.
    SearchError init_safetensors(int device, const void* blob, uint64_t blob_len, float ln_eps);
    // ids: concatenated token ids; text i owns ids[offsets[i]..offsets[i+1]).  out: [n, hidden] f32.
    // out_dev (on this embedder's device, may be null): the pooled vectors are left in device memory — the hand-off to a search that
    // takes device queries; out (may then be null): the host copy.  Either way the call returns when the forward has finished.
    SearchError embed_batch(const int32_t* ids, const uint32_t* offsets, uint32_t n, float* out, float* out_dev = nullptr);
    uint32_t dimension() const { return cfg_.hidden; }
    int device() const { return device_; }

  private:
    struct Layer {
        DeviceBuffer qkv_w, ao_w, i_w, o_w;              // f16 [N,K]
        DeviceBuffer qkv_wp, ao_wp, i_wp, o_wp;          // the same weights in matrix-core fragment order (bert_gemm_w.hip)
        DeviceBuffer qkv_b, ao_b, ln1_w, ln1_b, i_b, o_b, ln2_w, ln2_b;  // f32
    };
    SearchError upload_f32(DeviceBuffer& dst, const float* src, size_t n);
    SearchError upload_f16(DeviceBuffer& dst, const float* src, size_t n, DeviceBuffer& staging);
    SearchError pack_weights(DeviceBuffer& dst, const DeviceBuffer& src, int N, int K);
    SearchError forward(uint32_t n_docs, uint32_t tokens, uint32_t max_seq);
    // the fragment-order batch path over texts [d0, d1) = tokens [t0, t1) on `stream` (bert_gemm_w.hip)
    SearchError forward_packed_range(uint32_t d0, uint32_t d1, uint32_t t0, uint32_t t1, uint32_t max_seq, hipStream_t stream);
    SearchError forward_query(uint32_t n_docs, uint32_t tokens);   // <= 32 tokens: 25 launches (bert_query_kernels.hip)
    bool query_path(uint32_t tokens) const;
    bool one_launch_path() const;                                  // ... as ONE launch with grid-wide barriers (experiments builds)
    bool docs_path(uint32_t tokens, uint32_t max_seq) const;       // every text <= 32 tokens: ONE launch (bert_docs_w.hip)
    SearchError embed_docs(const int32_t* ids, const std::vector<uint32_t>& offs, uint32_t n, uint32_t total, float* out, float* out_dev);
    SearchError reserve_workspaces(uint32_t tokens);
    void drop_graphs();

    std::mutex mu_;
    int device_ = -1;
    fsgpu_bert_config cfg_{};
    hipStream_t stream_ = nullptr;
    DeviceBuffer word_, pos_, type_, emb_ln_w_, emb_ln_b_;
    std::vector<Layer> layers_;
    // workspaces
    DeviceBuffer ids_, positions_, offsets_, x_f32_, x_h_, qkv_f32_, ctx_h_, tmp_f32_, inter_h_, out_, q_x_, q_parts_;
    // pinned staging for small calls (see embed_batch)
    static constexpr size_t kPinnedIoBytes = 512 * 1024;
    void* io_host_ = nullptr;
    bool io_failed_ = false;
    float* pooled_out_ = nullptr;  // where the pool kernel writes during a pinned call
    const int32_t *q_ids_ = nullptr, *q_positions_ = nullptr;  // query path of a pinned call: inputs read in place
    const uint32_t* q_offsets_ = nullptr;
    // the one-launch query forward: the stage table on the device (+ the host copies it was sent from / is compared with), the
    // barrier counter, how many launches have used it, the mapped word a block raises when a barrier wait gave up
    DeviceBuffer q_stages_, q_counter_;
    std::vector<BertQueryArgs> q_stages_host_[2];
    std::vector<unsigned char> q_kinds_host_[2];
    int q_stages_flip_ = 0;
    unsigned int q_launches_ = 0;
    unsigned int* q_status_ = nullptr;
    bool q_one_launch_ok_ = true;
    // Query-sized calls replay a captured hipGraph of the whole call (three H2D copies + the ~44 kernels of the forward):
    // the chain is launch-bound, and a replay costs one submission instead of one per kernel.  One graph per call shape
    // (texts, tokens, longest text), built the second time a shape is seen; every buffer a graph names is allocated at its
    // graph-eligible maximum up front, and the graphs are dropped if a larger call ever moves one.
    struct GraphEntry {
        hipGraphExec_t exec = nullptr;
        uint32_t seen = 0;
    };
    static constexpr uint32_t kGraphMaxTokens = 8192;
    static constexpr size_t kGraphMaxEntries = 128;
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, GraphEntry> graphs_;
    bool graphs_enabled_ = true;
    bool packed_ = false;  // every layer has its fragment-order copies: the batch path runs bert_gemm_w.hip
    DeviceBuffer docs_layers_, docs_in_, docs_out_;   // bert_docs_w.hip: the layers' pointer table; inputs of a call too large for the pinned block
    bool docs_ready_ = false;
    static constexpr size_t kDocsIoBytes = 2 * 1024 * 1024;   // pinned block of the one-launch path: row tables in, pooled vectors out
    void* docs_io_ = nullptr;
    bool docs_io_failed_ = false;
};

}  // namespace fsgpu
