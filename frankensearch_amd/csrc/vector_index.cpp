// vector_index.cpp — host side of the device-resident VectorIndex and Model2VecEmbedder.
//
// Follows the reference's orchestration and error behaviour:
//   search_top_k_internal  crates/frankensearch-index/src/search.rs:426-494
//   ensure_query_dimension src/search.rs:1602-1610
//   parse_header / record table / slab offsets  src/lib.rs:4049-4144, 3510-3537, 1780-1816
//   soft_delete (tombstone flag)  src/lib.rs:171-173
// The data path itself runs in the HIP kernels of scan_kernels.hip; there is no CPU fallback.
#include "vector_index.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <type_traits>

#include "../../include/fsgpu.h"
#include "lab_env.hpp"

namespace fsgpu {

namespace {

SearchError ok() { return SearchError{}; }

SearchError hip_fail(hipError_t e, const char* what) {
    SearchError err;
    err.code = FSGPU_ERR_DEVICE;
    err.detail = std::string(what) + ": " + hipGetErrorString(e);
    return err;
}

#define FSGPU_HIP(expr)                                   \
    do {                                                  \
        hipError_t _e = (expr);                           \
        if (_e != hipSuccess) return hip_fail(_e, #expr); \
    } while (0)

#define FSGPU_TRY(expr)            \
    do {                           \
        SearchError _s = (expr);   \
        if (!_s.ok()) return _s;   \
    } while (0)

// Switches read from the environment ONCE (getenv is not safe against concurrent setenv).  A default build reads three:
// FSGPU_WIDE, FSGPU_FILTER, FSGPU_DEBUG_BATCHED (documented in include/fsgpu.h).  Everything else is a tuning / A-B knob of the
// lab and exists only in builds with -DFSGPU_EXPERIMENTS (FSGPU_BUILD_DEFS, frankensearch_amd/build.py; scripts/exp_*).
struct Knobs {
    int grid_blocks = 0, ra = 0, rb = 0, mfma_shape = 0, mfma_shape_i8 = 0, round = 0, i8_per_cu = 0;
    int wide = -1;  // FSGPU_WIDE: 0 = never the register-resident-query main pass, 2 / 3 = its query tiles per wave
    int filter = 0;     // FSGPU_FILTER: "f16" (1) / "i8" (2) pin the filter of the exact batched search; unset = automatic
    int slots_b = 0, slots_main = 0;   // FSGPU_SLOTS_B / FSGPU_SLOTS_MAIN: list slots per (query, block) of the wide kernel's stages
    int wide_max = 0;   // FSGPU_WIDE_MAX: cap on the query tiles per wave of the wide main pass (default: what the registers hold)
    int i8f_growth = 0; // FSGPU_I8F_GROWTH: sample growth factor of the int8 filter (default 4)
    bool no_skip_b = false, use_160 = false, debug_batched = false, no_reverse = false, no_wide_b = false, no_anchor = false;
    bool no_big_pool = false, no_heur_b = false, no_group_sample = false;
    int rb_pct = 0;      // FSGPU_RB_PCT: the second sample's size in percent of what the plan chose (tuning experiments only)
    int heur_rank = 0;   // FSGPU_HEUR_RANK: rank of the first sample whose score gates the anchoring-only second sample (default 4)
    Knobs() {
        auto env = [](const char* name) { return std::getenv(name); };
        if (const char* w = env("FSGPU_WIDE")) wide = std::atoi(w);
        if (const char* f = env("FSGPU_FILTER")) filter = std::strcmp(f, "f16") == 0 ? 1 : std::strcmp(f, "i8") == 0 ? 2 : 0;
        debug_batched = env("FSGPU_DEBUG_BATCHED") != nullptr;
#ifdef FSGPU_EXPERIMENTS
        auto num = [&](const char* name) {
            const char* e = env(name);
            return e ? std::atoi(e) : 0;
        };
        grid_blocks = num("FSGPU_GRID_BLOCKS");
        ra = num("FSGPU_RA");
        rb = num("FSGPU_RB");
        rb_pct = num("FSGPU_RB_PCT");
        round = num("FSGPU_ROUND");
        i8_per_cu = num("FSGPU_I8_PER_CU");
        mfma_shape = num("FSGPU_MFMA_SHAPE");
        mfma_shape_i8 = num("FSGPU_MFMA_SHAPE_I8");
        i8f_growth = num("FSGPU_I8F_GROWTH");
        wide_max = num("FSGPU_WIDE_MAX");
        slots_b = std::min(num("FSGPU_SLOTS_B"), (int)kWideSlots);
        slots_main = std::min(num("FSGPU_SLOTS_MAIN"), (int)kWideSlots);
        no_skip_b = env("FSGPU_NO_SKIP_B") != nullptr;
        no_wide_b = env("FSGPU_NO_WIDE_B") != nullptr;
        no_anchor = env("FSGPU_NO_ANCHOR") != nullptr;
        no_big_pool = env("FSGPU_NO_BIG_POOL") != nullptr;
        no_heur_b = env("FSGPU_NO_HEUR_B") != nullptr;
        no_group_sample = env("FSGPU_NO_GROUP_SAMPLE") != nullptr;
        heur_rank = num("FSGPU_HEUR_RANK");
        no_reverse = env("FSGPU_NO_REVERSE") != nullptr;
        use_160 = env("FSGPU_USE_160") != nullptr;
#endif
    }
};
const Knobs& knobs() {
    static const Knobs k;
    return k;
}

SearchError make_error(int32_t code, std::string detail) {
    SearchError e;
    e.code = code;
    e.detail = std::move(detail);
    return e;
}

uint32_t crc32_ieee(const uint8_t* p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int b = 0; b < 8; ++b) c = (c & 1u) ? (0xedb88320u ^ (c >> 1)) : (c >> 1);
            table[i] = c;
        }
        init = true;
    }
    uint32_t c = 0xffffffffu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xffu] ^ (c >> 8);
    return ~c;
}

uint64_t fnv1a(const char* p, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) {
        h ^= (uint8_t)p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

template <typename T>
T read_le(const uint8_t* p) {
    T v = 0;
    for (size_t i = 0; i < sizeof(T); ++i) v |= (T)p[i] << (8 * i);
    return v;
}

}  // namespace

SearchError DeviceBuffer::reserve(size_t want) {
    if (want <= bytes && ptr) return ok();
    release();
    size_t alloc = want < 256 ? 256 : want;
    FSGPU_HIP(hipMalloc(&ptr, alloc));
    bytes = alloc;
    return ok();
}

void DeviceBuffer::release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
}

// ------------------------------------------------------------------------------------------------
// VectorIndex
// ------------------------------------------------------------------------------------------------

VectorIndex::~VectorIndex() {
    if (device_ >= 0) (void)hipSetDevice(device_);
    for (auto& ev : events_) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    for (hipEvent_t& e : async_ev_)
        if (e) (void)hipEventDestroy(e);
    if (stream_) (void)hipStreamDestroy(stream_);
    for (DeviceBuffer* b : {&slab_own_, &live_own_, &ws_partial_, &ws_queries_, &ws_allow_, &ws_rows_, &ws_scores_,
                            &ws_counts_, &ws_keys_a_, &ws_keys_b_, &ws_sort_tmp_, &ws_gather_rows_, &ws_gather_out_,
                            &i8_slab_, &n4_slab_, &i8_max_, &ws_i8_query_, &ws_cand_packed_, &ws_cand_rows_, &ws_cand_scores_,
                            &mf_max_norm_, &mf_qh_, &mf_delta_, &mf_tau_, &mf_cand_, &mf_dense_, &mf_sel_,
                            &mf_fallback_, &mf_fallback2_, &mf_spill_, &mf_io_, &mf_io2_, &i8_stats_, &n4u_slab_, &mf_cand_count_, &ws_pairs_,
                            &i8f_slab_, &i8f_max_, &i8f_stats_, &rot_mat_, &rot_q_})
        b->release();
    if (mf_flags_host_) (void)hipHostFree(mf_flags_host_);
    if (io_host_) (void)hipHostFree(io_host_);
}

SearchError VectorIndex::common_init(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return make_error(FSGPU_ERR_NO_DEVICE, "no HIP device visible (libfsgpu has no CPU fallback)");
    if (device < 0 || device >= count) return make_error(FSGPU_ERR_INVALID_CONFIG, "device ordinal out of range");
    FSGPU_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    FSGPU_HIP(hipGetDeviceProperties(&prop, device));
    num_cus_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    device_ = device;
    FSGPU_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    return ok();
}

SearchError VectorIndex::init_host(int device, uint32_t dim, uint64_t nrows, const void* slab, const uint64_t* live,
                                   uint64_t row_base, bool f32_rows) {
    if (dim == 0) return make_error(FSGPU_ERR_INVALID_CONFIG, "dimension must be greater than zero");
    if (nrows + row_base >= 0xffffffffull)
        return make_error(FSGPU_ERR_INVALID_CONFIG, "row ids must fit in u32 (VectorHit.index)");
    if (nrows > 0 && !slab) return make_error(FSGPU_ERR_NULL_ARGUMENT, "slab is null");
    FSGPU_TRY(common_init(device));
    dim_ = dim;
    nrows_ = nrows;
    row_base_ = row_base;
    f32_ = f32_rows;
    const size_t bytes = (size_t)nrows * dim * (f32_rows ? 4 : 2);
    FSGPU_TRY(slab_own_.reserve(bytes));
    if (bytes) FSGPU_HIP(hipMemcpy(slab_own_.ptr, slab, bytes, hipMemcpyHostToDevice));
    slab_dev_ = slab_own_.ptr;
    owns_slab_ = true;
    return set_live_bitmap(live);
}

SearchError VectorIndex::init_device(int device, uint32_t dim, uint64_t nrows, const void* slab_dev,
                                     const uint64_t* live_dev, uint64_t row_base) {
    if (dim == 0) return make_error(FSGPU_ERR_INVALID_CONFIG, "dimension must be greater than zero");
    if (nrows + row_base >= 0xffffffffull)
        return make_error(FSGPU_ERR_INVALID_CONFIG, "row ids must fit in u32 (VectorHit.index)");
    if (nrows > 0 && !slab_dev) return make_error(FSGPU_ERR_NULL_ARGUMENT, "slab is null");
    FSGPU_TRY(common_init(device));
    dim_ = dim;
    nrows_ = nrows;
    row_base_ = row_base;
    slab_dev_ = slab_dev;
    live_dev_ = live_dev;
    owns_slab_ = false;
    return ok();
}

SearchError VectorIndex::set_live_bitmap(const uint64_t* live) {
    if (async_state_[0] == 1 || async_state_[1] == 1)   // (its kernels read the live bitmap this call would rewrite)
        return make_error(FSGPU_ERR_INVALID_CONFIG, "a begun batched search is outstanding on this index: end it first");
    if (!live) {
        live_dev_ = nullptr;
        live_host_.clear();
        return ok();
    }
    if (catalog_only_) {   // (the shards hold the device copies: sharded_index.cpp pushes the slices)
        live_host_.assign(live, live + (size_t)((nrows_ + 63) / 64));
        return ok();
    }
    FSGPU_HIP(hipSetDevice(device_));
    const size_t words = (size_t)((nrows_ + 63) / 64);
    live_host_.assign(live, live + words);
    FSGPU_TRY(live_own_.reserve(words * 8));
    if (words) FSGPU_HIP(hipMemcpy(live_own_.ptr, live_host_.data(), words * 8, hipMemcpyHostToDevice));
    live_dev_ = static_cast<const uint64_t*>(live_own_.ptr);
    return ok();
}

// VectorIndexWriter::write_record + finish for FSVI v1 (crates/frankensearch-index/src/lib.rs:3637-3672, 3752-3943;
// header :5714-5768): records are validated (finite, usable signal, doc id <= u16 bytes), STABLY sorted by
// (FNV-1a(doc_id), doc_id) (:3753-3762), and written as header | 16-byte records | string table | pad to 64 | f16 slab.
// The f32 -> f16 conversion (round to nearest even, simd.rs:2245-2305) of the slab runs on the GPU, row-permuted into file
// order; everything else is host bookkeeping.
SearchError write_fsvi_v1(const char* path, const char* embedder_id, const char* embedder_revision, uint32_t dim, uint64_t n,
                          const char* const* doc_ids, const uint32_t* doc_id_lens, const float* vectors,
                          uint8_t compaction_gen, int device, uint8_t quantization) {
    if (quantization > 1) return make_error(FSGPU_ERR_INVALID_CONFIG, "quantization must be 0 (F32) or 1 (F16)");
    if (!path || !embedder_id || !embedder_revision || (n && (!doc_ids || !vectors)))
        return make_error(FSGPU_ERR_NULL_ARGUMENT, "null argument");
    if (dim == 0) return make_error(FSGPU_ERR_INVALID_CONFIG, "dimension must be greater than zero");
    const size_t idl = std::strlen(embedder_id), rvl = std::strlen(embedder_revision);
    if (idl > 0xffff || rvl > 0xffff) return make_error(FSGPU_ERR_INVALID_CONFIG, "embedder id / revision must fit in u16");
    struct Pending {
        uint64_t hash;
        const char* id;
        uint32_t len;
        uint64_t seq;
    };
    std::vector<Pending> recs((size_t)n);
    uint64_t strings_len = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const float* v = vectors + (size_t)i * dim;
        float norm_sq = 0.f;
        for (uint32_t d = 0; d < dim; ++d) {
            if (!std::isfinite(v[d])) return make_error(FSGPU_ERR_INVALID_CONFIG, "all embedding values must be finite");
            const float pq = v[d] * v[d];
            norm_sq = norm_sq + pq;
        }
        if (!(norm_sq > 0.0f) || !std::isfinite(norm_sq))
            return make_error(FSGPU_ERR_INVALID_CONFIG, "embedding norm must be non-zero and finite");
        const size_t len = doc_id_lens ? doc_id_lens[i] : std::strlen(doc_ids[i]);
        if (len > 0xffff) return make_error(FSGPU_ERR_INVALID_CONFIG, "doc_id byte length must fit in u16");
        recs[(size_t)i] = Pending{fnv1a(doc_ids[i], len), doc_ids[i], (uint32_t)len, i};
        strings_len += len;
    }
    if (strings_len > 0xffffffffull) return make_error(FSGPU_ERR_INVALID_CONFIG, "string table exceeds u32 offsets");
    std::stable_sort(recs.begin(), recs.end(), [](const Pending& a, const Pending& b) {
        if (a.hash != b.hash) return a.hash < b.hash;
        const int c = std::memcmp(a.id, b.id, std::min(a.len, b.len));
        if (c != 0) return c < 0;
        return a.len < b.len;
    });
    const size_t header_len = 4 + 2 + 2 + idl + 2 + rvl + 4 + 1 + 3 + 8 + 8 + 4;
    const uint64_t pre = (uint64_t)header_len + n * 16 + strings_len;
    const uint64_t vectors_offset = (pre + 63) / 64 * 64;
    const size_t slab_bytes = (size_t)n * dim * (quantization == 1 ? 2 : 4);
    std::vector<uint8_t> buf((size_t)vectors_offset + slab_bytes, 0);
    auto put = [&](size_t at, uint64_t v, int bytes) {
        for (int b = 0; b < bytes; ++b) buf[at + b] = (uint8_t)(v >> (8 * b));
    };
    size_t c = 0;
    std::memcpy(buf.data(), "FSVI", 4);
    c += 4;
    put(c, 1, 2);
    c += 2;
    put(c, idl, 2);
    c += 2;
    std::memcpy(buf.data() + c, embedder_id, idl);
    c += idl;
    put(c, rvl, 2);
    c += 2;
    std::memcpy(buf.data() + c, embedder_revision, rvl);
    c += rvl;
    put(c, dim, 4);
    c += 4;
    buf[c++] = quantization;  // Quantization::{F32 = 0, F16 = 1} (lib.rs:203-208)
    buf[c++] = compaction_gen;
    put(c, 0, 2);  // publication nonce
    c += 2;
    put(c, n, 8);
    c += 8;
    put(c, vectors_offset, 8);
    c += 8;
    put(c, crc32_ieee(buf.data(), c), 4);
    c += 4;
    size_t str_off = 0;
    const size_t str_base = c + (size_t)n * 16;
    std::vector<uint32_t> perm((size_t)n);
    for (uint64_t i = 0; i < n; ++i) {
        const Pending& r = recs[(size_t)i];
        put(c + (size_t)i * 16, r.hash, 8);
        put(c + (size_t)i * 16 + 8, str_off, 4);
        put(c + (size_t)i * 16 + 12, r.len, 2);
        put(c + (size_t)i * 16 + 14, 0, 2);
        std::memcpy(buf.data() + str_base + str_off, r.id, r.len);
        str_off += r.len;
        perm[(size_t)i] = (uint32_t)r.seq;
    }
    if (n && quantization == 0) {
        // Quantization::F32: the rows as they are, little-endian (write_vector_slab, lib.rs:6017-6024), in sorted order
        for (uint64_t i = 0; i < n; ++i)
            std::memcpy(buf.data() + vectors_offset + (size_t)i * dim * 4, vectors + (size_t)perm[(size_t)i] * dim, (size_t)dim * 4);
    } else if (n) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            return make_error(FSGPU_ERR_NO_DEVICE, "no HIP device visible (libfsgpu has no CPU fallback)");
        if (device < 0 || device >= count) return make_error(FSGPU_ERR_INVALID_CONFIG, "device ordinal out of range");
        FSGPU_HIP(hipSetDevice(device));
        DeviceBuffer src, pidx, dst;
        SearchError e = src.reserve((size_t)n * dim * 4);
        if (e.ok()) e = pidx.reserve((size_t)n * 4);
        if (e.ok()) e = dst.reserve(slab_bytes);
        hipError_t he = hipSuccess;
        if (e.ok()) {
            he = hipMemcpy(src.ptr, vectors, (size_t)n * dim * 4, hipMemcpyHostToDevice);
            if (he == hipSuccess) he = hipMemcpy(pidx.ptr, perm.data(), (size_t)n * 4, hipMemcpyHostToDevice);
            if (he == hipSuccess)
                he = launch_encode_rows_f16(static_cast<const float*>(src.ptr), static_cast<const uint32_t*>(pidx.ptr), n, dim,
                                            static_cast<unsigned short*>(dst.ptr), nullptr);
            if (he == hipSuccess) he = hipMemcpy(buf.data() + vectors_offset, dst.ptr, slab_bytes, hipMemcpyDeviceToHost);
        }
        src.release();
        pidx.release();
        dst.release();
        if (!e.ok()) return e;
        if (he != hipSuccess) return make_error(FSGPU_ERR_DEVICE, hipGetErrorString(he));
    }
    FILE* f = std::fopen(path, "wb");
    if (!f) return make_error(FSGPU_ERR_IO, std::string("cannot create ") + path);
    const size_t w = std::fwrite(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    if (w != buf.size()) return make_error(FSGPU_ERR_IO, std::string("short write to ") + path);
    return ok();
}

// VectorIndex::open for FSVI v1 (lib.rs:1747-1816, parse_header :4049-4144).
namespace {
// std::str::from_utf8 (the header strings, lib.rs:4073-4095): well-formed UTF-8 only — no overlongs, surrogates or > U+10FFFF.
bool valid_utf8(const uint8_t* p, size_t n) {
    size_t i = 0;
    while (i < n) {
        const uint8_t b = p[i];
        if (b < 0x80) {
            ++i;
            continue;
        }
        size_t extra;
        uint32_t cp, min;
        if ((b & 0xe0) == 0xc0) {
            extra = 1, cp = b & 0x1f, min = 0x80;
        } else if ((b & 0xf0) == 0xe0) {
            extra = 2, cp = b & 0x0f, min = 0x800;
        } else if ((b & 0xf8) == 0xf0) {
            extra = 3, cp = b & 0x07, min = 0x10000;
        } else {
            return false;
        }
        if (i + extra >= n) return false;  // truncated sequence
        for (size_t j = 1; j <= extra; ++j) {
            if ((p[i + j] & 0xc0) != 0x80) return false;
            cp = (cp << 6) | (p[i + j] & 0x3f);
        }
        if (cp < min || cp > 0x10ffff || (cp >= 0xd800 && cp <= 0xdfff)) return false;
        i += extra + 1;
    }
    return true;
}
}  // namespace

SearchError VectorIndex::open_fsvi(const char* path, int device) { return open_fsvi_impl(path, device, nullptr); }

// The same reader for a row-SHARDED index (sharded_index.cpp): this object keeps the record table, the doc-id strings, the
// tombstone bitmap and the WAL of the whole file — it resolves, deduplicates and shadows hits — while the slab goes to the shards.
SearchError VectorIndex::open_fsvi_catalog(const char* path, FsviImage* image) {
    if (!image) return make_error(FSGPU_ERR_NULL_ARGUMENT, "image is null");
    return open_fsvi_impl(path, -1, image);
}

SearchError VectorIndex::open_fsvi_impl(const char* path, int device, FsviImage* image) {
    if (!path) return make_error(FSGPU_ERR_NULL_ARGUMENT, "path is null");
    FILE* f = std::fopen(path, "rb");
    if (!f) return make_error(FSGPU_ERR_IO, std::string("cannot open ") + path);
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> data((size_t)(sz > 0 ? sz : 0));
    const size_t rd = data.empty() ? 0 : std::fread(data.data(), 1, data.size(), f);
    std::fclose(f);
    if (rd != data.size()) return make_error(FSGPU_ERR_IO, std::string("short read on ") + path);

    auto corrupt = [&](const std::string& detail) {
        return make_error(FSGPU_ERR_INDEX_CORRUPTED, std::string(path) + ": " + detail);
    };
    size_t c = 0;
    auto need = [&](size_t n) { return c + n <= data.size(); };
    if (!need(4) || std::memcmp(data.data(), "FSVI", 4) != 0) return corrupt("bad magic bytes");
    c = 4;
    if (!need(2)) return corrupt("truncated header (version)");
    const uint16_t version = read_le<uint16_t>(&data[c]);
    c += 2;
    if (version != 1)
        return make_error(FSGPU_ERR_INDEX_VERSION_MISMATCH,
                          "FSVI version expected 1, found " + std::to_string(version));
    for (const char* field : {"embedder_id", "embedder_revision"}) {
        if (!need(2)) return corrupt(std::string("truncated header (") + field + "_len)");
        const size_t len = read_le<uint16_t>(&data[c]);
        c += 2;
        if (!need(len)) return corrupt(std::string("truncated header (") + field + ")");
        if (!valid_utf8(&data[c], len)) return corrupt(std::string("invalid UTF-8 in ") + field);  // lib.rs:4073-4095
        c += len;
    }
    if (!need(4)) return corrupt("truncated header (dimension)");
    const uint32_t dim = read_le<uint32_t>(&data[c]);
    c += 4;
    if (dim == 0) return corrupt("dimension must be greater than zero");
    if (!need(1)) return corrupt("truncated header (quantization)");
    const uint8_t quant = data[c++];
    if (quant > 1) return corrupt("unknown quantization byte");
    if (!need(3)) return corrupt("truncated header (reserved)");
    c += 3;
    if (!need(16)) return corrupt("truncated header (record_count / vectors_offset)");
    const uint64_t record_count = read_le<uint64_t>(&data[c]);
    c += 8;
    const uint64_t vectors_offset = read_le<uint64_t>(&data[c]);
    c += 8;
    if (!need(4)) return corrupt("truncated header (crc)");
    const uint32_t want_crc = read_le<uint32_t>(&data[c]);
    if (crc32_ieee(data.data(), c) != want_crc) return corrupt("header CRC mismatch");
    c += 4;
    const uint64_t elem = quant == 1 ? 2 : 4;  // Quantization::{F32 = 0, F16 = 1} (lib.rs:203-208)
    const size_t records_offset = c;
    // checked arithmetic as in VectorIndex::open (lib.rs:1782-1816): a crafted header with a valid CRC must end in
    // IndexCorrupted, not in a wrapped bound that passes.  (The v1 reader does not require an aligned vectors_offset;
    // the slab is copied into a fresh device allocation, so the kernels' 16-byte loads do not depend on it.)
    uint64_t records_bytes = 0, strings_offset = 0, vector_bytes = 0, required_len = 0;
    if (__builtin_mul_overflow(record_count, (uint64_t)16, &records_bytes)) return corrupt("record table size overflow");
    if (__builtin_add_overflow((uint64_t)records_offset, records_bytes, &strings_offset))
        return corrupt("record table offset overflow");
    if (vectors_offset < strings_offset)
        return corrupt("vectors_offset points inside the record table/string table region");
    if (__builtin_mul_overflow(record_count, (uint64_t)dim, &vector_bytes) ||
        __builtin_mul_overflow(vector_bytes, elem, &vector_bytes))
        return corrupt("vector slab size overflow");
    if (__builtin_add_overflow(vectors_offset, vector_bytes, &required_len)) return corrupt("vector slab end overflow");
    if (data.size() < required_len)
        return corrupt("truncated file: have " + std::to_string(data.size()) + " bytes, need at least " +
                       std::to_string(required_len) + " bytes");

    std::vector<uint64_t> live((size_t)((record_count + 63) / 64), 0);
    doc_hashes_.resize((size_t)record_count);
    doc_offsets_.assign((size_t)record_count + 1, 0);
    doc_blob_.clear();
    for (uint64_t r = 0; r < record_count; ++r) {
        const uint8_t* rec = &data[records_offset + r * 16];
        const uint64_t off = read_le<uint32_t>(rec + 8), len = read_le<uint16_t>(rec + 12);
        const uint16_t flags = read_le<uint16_t>(rec + 14);
        if (strings_offset + off + len > vectors_offset) return corrupt("doc_id string out of bounds");
        doc_hashes_[(size_t)r] = read_le<uint64_t>(rec);
        doc_offsets_[(size_t)r] = doc_blob_.size();
        doc_blob_.append(reinterpret_cast<const char*>(&data[strings_offset + off]), (size_t)len);
        if ((flags & 0x0001u) == 0) live[(size_t)(r >> 6)] |= 1ull << (r & 63);
    }
    doc_offsets_[(size_t)record_count] = doc_blob_.size();
    if (image) {   // catalog of a sharded index: no device copy here
        catalog_only_ = true;
        dim_ = dim;
        nrows_ = record_count;
        row_base_ = 0;
        f32_ = quant == 0;
        live_host_ = live;
        image->dim = dim;
        image->nrows = record_count;
        image->f32_rows = quant == 0;
        image->slab_offset = (size_t)vectors_offset;
        image->bytes = std::move(data);
        return ok();
    }
    return init_host(device, dim, record_count, data.data() + vectors_offset, live.data(), 0, quant == 0);
}

SearchError VectorIndex::doc_id_at(uint32_t row, const char** ptr, uint32_t* len) const {
    if (doc_offsets_.empty()) return make_error(FSGPU_ERR_INVALID_CONFIG, "index has no doc-id table");
    if (row >= nrows_) {  // WAL virtual row = record_count + wal index (search.rs:1579-1596)
        const uint64_t wi = row - nrows_;
        if (wi >= wal_.size()) return make_error(FSGPU_ERR_INDEX_CORRUPTED, "WAL index out of bounds");
        *ptr = wal_[wi].doc_id.data();
        *len = (uint32_t)wal_[wi].doc_id.size();
        return ok();
    }
    *ptr = doc_blob_.data() + doc_offsets_[row];
    *len = (uint32_t)(doc_offsets_[row + 1] - doc_offsets_[row]);
    return ok();
}

namespace {

// dot_product_f32_f32 (crates/frankensearch-index/src/simd.rs:134-222), host side (WAL rows stay on the CPU
// in the reference too): groups of 32 into four 8-lane accumulators, (a0+a1)+(a2+a3), leftover chunks into
// the sum, horizontal add, scalar tail with separate multiply and add.  Built with -ffp-contract=off.
float dot_f32_f32(const float* a, const float* b, size_t n, int hreduce) {
    const size_t groups = n / 32, chunks = n / 8;
    float acc[4][8] = {};
    for (size_t g = 0; g < groups; ++g)
        for (int x = 0; x < 4; ++x)
            for (int j = 0; j < 8; ++j) {
                const size_t o = g * 32 + (size_t)x * 8 + (size_t)j;
                const float p = a[o] * b[o];
                acc[x][j] = acc[x][j] + p;
            }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = (acc[0][j] + acc[1][j]) + (acc[2][j] + acc[3][j]);
    for (size_t c = groups * 4; c < chunks; ++c)
        for (int j = 0; j < 8; ++j) {
            const float p = a[c * 8 + (size_t)j] * b[c * 8 + (size_t)j];
            v[j] = v[j] + p;
        }
    float result;
    if (hreduce == FSGPU_HREDUCE_SEQ) {
        const float lo = ((v[0] + v[1]) + v[2]) + v[3];
        const float hi = ((v[4] + v[5]) + v[6]) + v[7];
        result = lo + hi;
    } else if (hreduce == FSGPU_HREDUCE_AVX) {
        const float s0 = v[0] + v[4], s1 = v[1] + v[5], s2 = v[2] + v[6], s3 = v[3] + v[7];
        const float lo = s0 + s2, hi = s1 + s3;
        result = lo + hi;
    } else {
        const float lo = (v[0] + v[2]) + (v[1] + v[3]);
        const float hi = (v[4] + v[6]) + (v[5] + v[7]);
        result = lo + hi;
    }
    for (size_t i = chunks * 8; i < n; ++i) {
        const float p = a[i] * b[i];
        result = result + p;
    }
    return result;
}

// monotone image of score_key + f32::total_cmp (search.rs:1655-1686); larger = ranks earlier
uint32_t host_score_ord(float score) {
    uint32_t bits;
    std::memcpy(&bits, &score, 4);
    if ((bits & 0x7fffffffu) > 0x7f800000u) bits = 0xff800000u;
    return (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
}

}  // namespace

SearchError VectorIndex::wal_append(const char* doc_id, uint32_t len, const float* vector, uint32_t vector_len) {
    if (doc_offsets_.empty()) return make_error(FSGPU_ERR_INVALID_CONFIG, "index has no doc-id table");
    if (async_state_[0] == 1 || async_state_[1] == 1)   // (its kernels read the live bitmap this call would rewrite)
        return make_error(FSGPU_ERR_INVALID_CONFIG, "a begun batched search is outstanding on this index: end it first");
    if (vector_len != dim_)
        return make_error(FSGPU_ERR_DIMENSION_MISMATCH,
                          "expected " + std::to_string(dim_) + ", found " + std::to_string(vector_len));
    float norm_sq = 0.f;
    for (uint32_t i = 0; i < vector_len; ++i) {
        if (!std::isfinite(vector[i]))
            return make_error(FSGPU_ERR_INVALID_CONFIG, "all embedding values must be finite");
        const float p = vector[i] * vector[i];
        norm_sq = norm_sq + p;
    }
    if (!(norm_sq > 0.0f) || !std::isfinite(norm_sq))
        return make_error(FSGPU_ERR_INVALID_CONFIG, "embedding norm must be non-zero and finite");
    if (len > 0xffffu) return make_error(FSGPU_ERR_INVALID_CONFIG, "doc_id byte length must fit in u16");
    const std::string id(doc_id, len);
    // supersede older resident copies (lib.rs:2641-2647), then admit the new entry
    wal_.erase(std::remove_if(wal_.begin(), wal_.end(), [&](const WalEntry& e) { return e.doc_id == id; }),
               wal_.end());
    wal_.push_back(WalEntry{id, std::vector<float>(vector, vector + vector_len)});
    // tombstone the first live main row with this doc id so it cannot take a top-k slot (lib.rs:2665-2710)
    const uint64_t h = fnv1a(doc_id, len);
    auto lo = std::lower_bound(doc_hashes_.begin(), doc_hashes_.end(), h);
    for (auto it = lo; it != doc_hashes_.end() && *it == h; ++it) {
        const size_t r = (size_t)(it - doc_hashes_.begin());
        const size_t dl = (size_t)(doc_offsets_[r + 1] - doc_offsets_[r]);
        if (dl != len || std::memcmp(doc_blob_.data() + doc_offsets_[r], doc_id, len) != 0) continue;
        if (live_host_.empty()) live_host_.assign((size_t)((nrows_ + 63) / 64), ~0ull);
        if ((live_host_[r >> 6] >> (r & 63)) & 1ull) {
            live_host_[r >> 6] &= ~(1ull << (r & 63));
            std::vector<uint64_t> copy = live_host_;
            FSGPU_TRY(set_live_bitmap(copy.data()));
            break;
        }
    }
    return ok();
}

SearchError VectorIndex::search_hits(const float* query, uint32_t query_len, uint32_t k, uint32_t* out_rows,
                                     float* out_scores, uint32_t* out_count) {
    *out_count = 0;
    if (doc_offsets_.empty()) return make_error(FSGPU_ERR_INVALID_CONFIG, "index has no doc-id table");
    FSGPU_TRY(ensure_query_dimension(query_len));
    if (k == 0 || (nrows_ == 0 && wal_.empty())) return ok();
    struct Cand {
        uint64_t index;  // main row, or WAL-tagged (top bit) like wal.rs:557-569
        float score;
    };
    const uint64_t wal_tag = 1ull << 63;
    std::vector<Cand> cand;
    if (nrows_ > 0) {
        std::vector<uint32_t> rows(k);
        std::vector<float> scores(k);
        uint32_t count = 0;
        if (topk_override) FSGPU_TRY(topk_override(query, k, rows.data(), scores.data(), &count));   // the shards' merged top-k
        else FSGPU_TRY(search_top_k(query, 1, query_len, k, nullptr, rows.data(), scores.data(), &count));
        for (uint32_t i = 0; i < count; ++i) cand.push_back(Cand{rows[i], scores[i]});
    }
    for (size_t w = 0; w < wal_.size(); ++w) {
        const float s = dot_f32_f32(wal_[w].embedding.data(), query, dim_, hreduce);
        if (!std::isfinite(s)) continue;  // search.rs:1466-1470
        cand.push_back(Cand{wal_tag | w, s});
    }
    std::sort(cand.begin(), cand.end(), [](const Cand& a, const Cand& b) {
        const uint32_t ka = host_score_ord(a.score), kb = host_score_ord(b.score);
        if (ka != kb) return ka > kb;
        return a.index < b.index;
    });
    if (cand.size() > k) cand.resize(k);  // the size-k heap holds exactly the k best of main U wal
    uint32_t n = 0;
    std::vector<std::pair<const char*, uint32_t>> seen;
    for (const Cand& c : cand) {
        const char* di = nullptr;
        uint32_t dl = 0;
        uint32_t index;
        if (c.index & wal_tag) {
            const size_t w = (size_t)(c.index & ~wal_tag);
            di = wal_[w].doc_id.data();
            dl = (uint32_t)wal_[w].doc_id.size();
            index = (uint32_t)(nrows_ + w);
        } else {
            const size_t r = (size_t)(c.index - row_base_);
            if (!live_host_.empty() && !((live_host_[r >> 6] >> (r & 63)) & 1ull)) continue;
            di = doc_blob_.data() + doc_offsets_[r];
            dl = (uint32_t)(doc_offsets_[r + 1] - doc_offsets_[r]);
            bool shadowed = false;
            for (const WalEntry& e : wal_)
                if (e.doc_id.size() == dl && std::memcmp(e.doc_id.data(), di, dl) == 0) shadowed = true;
            if (shadowed) continue;
            index = (uint32_t)c.index;
        }
        bool dup = false;
        for (auto& sd : seen)
            if (sd.second == dl && std::memcmp(sd.first, di, dl) == 0) dup = true;
        if (dup) continue;
        seen.emplace_back(di, dl);
        out_rows[n] = index;
        out_scores[n] = c.score;
        ++n;
    }
    *out_count = n;
    return ok();
}

int64_t VectorIndex::find_index_by_doc_id(const char* doc_id, uint32_t len) const {
    if (doc_offsets_.empty()) return -1;
    const uint64_t h = fnv1a(doc_id, len);
    auto lo = std::lower_bound(doc_hashes_.begin(), doc_hashes_.end(), h);
    for (auto it = lo; it != doc_hashes_.end() && *it == h; ++it) {
        const size_t r = (size_t)(it - doc_hashes_.begin());
        if (row_tombstoned(r)) continue;
        const size_t dl = (size_t)(doc_offsets_[r + 1] - doc_offsets_[r]);
        if (dl == len && std::memcmp(doc_blob_.data() + doc_offsets_[r], doc_id, len) == 0) return (int64_t)r;
    }
    return -1;
}

int64_t VectorIndex::wal_latest(const char* doc_id, uint32_t len) const {
    for (size_t i = wal_.size(); i-- > 0;)
        if (wal_[i].doc_id.size() == len && std::memcmp(wal_[i].doc_id.data(), doc_id, len) == 0) return (int64_t)i;
    return -1;
}

float VectorIndex::wal_dot(size_t wal_index, const float* query) const {
    return dot_f32_f32(wal_[wal_index].embedding.data(), query, dim_, hreduce);
}

SearchError VectorIndex::soft_delete(const char* doc_id, uint32_t len, int32_t* deleted) {
    *deleted = 0;
    if (doc_offsets_.empty()) return make_error(FSGPU_ERR_INVALID_CONFIG, "index has no doc-id table");
    if (async_state_[0] == 1 || async_state_[1] == 1)   // (its kernels read the live bitmap this call would rewrite)
        return make_error(FSGPU_ERR_INVALID_CONFIG, "a begun batched search is outstanding on this index: end it first");
    const uint64_t h = fnv1a(doc_id, len);
    // rows are sorted by (hash, doc_id) (lib.rs:3758-3762): binary-search the hash run
    auto lo = std::lower_bound(doc_hashes_.begin(), doc_hashes_.end(), h);
    bool changed = false;
    for (auto it = lo; it != doc_hashes_.end() && *it == h; ++it) {
        const size_t r = (size_t)(it - doc_hashes_.begin());
        const size_t dl = (size_t)(doc_offsets_[r + 1] - doc_offsets_[r]);
        if (dl == len && std::memcmp(doc_blob_.data() + doc_offsets_[r], doc_id, len) == 0) {
            if (live_host_.empty()) live_host_.assign((size_t)((nrows_ + 63) / 64), ~0ull);
            if ((live_host_[r >> 6] >> (r & 63)) & 1ull) {
                live_host_[r >> 6] &= ~(1ull << (r & 63));
                changed = true;
            }
        }
    }
    if (changed) {
        std::vector<uint64_t> copy = live_host_;
        FSGPU_TRY(set_live_bitmap(copy.data()));
        *deleted = 1;
    }
    // step 2 of soft_delete_batch (lib.rs:2358-2373): resident WAL versions of the document go too and count as deleted —
    // after wal_append the main row is already tombstoned, and the WAL entry is what keeps the document searchable
    const size_t before = wal_.size();
    wal_.erase(std::remove_if(wal_.begin(), wal_.end(),
                              [&](const WalEntry& e) { return e.doc_id.size() == len && std::memcmp(e.doc_id.data(), doc_id, len) == 0; }),
               wal_.end());
    if (wal_.size() != before) *deleted = 1;
    return ok();
}

// gather_positions_for_hashes (search.rs:1146-1164) as a row bitmap: the rows of each hash are one run of the
// (hash, doc_id)-sorted record table (hash_range, search.rs:1166-1198).
SearchError VectorIndex::allow_bitmap_for_hashes(const uint64_t* hashes, uint32_t n, uint64_t* bitmap_out,
                                                 uint64_t* matched) const {
    if (doc_hashes_.empty() && nrows_ != 0) return make_error(FSGPU_ERR_INVALID_CONFIG, "index has no record table");
    const size_t words = (size_t)((nrows_ + 63) / 64);
    std::memset(bitmap_out, 0, words * 8);
    uint64_t count = 0;
    for (uint32_t i = 0; i < n; ++i) {
        auto lo = std::lower_bound(doc_hashes_.begin(), doc_hashes_.end(), hashes[i]);
        for (auto it = lo; it != doc_hashes_.end() && *it == hashes[i]; ++it) {
            const size_t r = (size_t)(it - doc_hashes_.begin());
            const uint64_t bit = 1ull << (r & 63);
            if (!(bitmap_out[r >> 6] & bit)) ++count;  // a hash may be listed twice
            bitmap_out[r >> 6] |= bit;
        }
    }
    if (matched) *matched = count;
    return ok();
}

// 256 KB of pinned, device-visible host memory per index for the latency paths (allocated on first use).
void* VectorIndex::pinned_io() {
    if (!io_host_ && !io_failed_) {
        if (hipHostMalloc(&io_host_, kPinnedIoBytes, hipHostMallocMapped) != hipSuccess) {
            io_host_ = nullptr;
            io_failed_ = true;
            (void)hipGetLastError();
        }
    }
    return io_host_;
}

SearchError VectorIndex::ensure_query_dimension(uint32_t query_len) const {
    if (query_len != dim_)
        return make_error(FSGPU_ERR_DIMENSION_MISMATCH,
                          "expected " + std::to_string(dim_) + ", found " + std::to_string(query_len));
    return ok();
}

ScanArgs VectorIndex::base_args(const float* queries_dev, const uint64_t* allow_dev) const {
    ScanArgs a;
    a.slab = slab_dev_;
    a.live = reinterpret_cast<const u64*>(live_dev_);
    a.allow = reinterpret_cast<const u64*>(allow_dev);
    a.queries = queries_dev;
    a.partial = nullptr;
    a.nrows = (uint32_t)nrows_;
    a.dim = dim_;
    a.k = 0;
    a.row_base = (uint32_t)row_base_;
    a.hreduce = hreduce;
    a.row_stride = row_stride_ ? row_stride_ : dim_ * (f32_ ? 4 : 2);
    return a;
}

SearchError VectorIndex::fused_search(const float* queries_dev, uint32_t nq, uint32_t k_out, uint32_t k_eff,
                                      const uint64_t* allow_dev, uint32_t* out_rows_dev, float* out_scores_dev,
                                      uint32_t* out_counts_dev, u64* out_packed_dev, hipStream_t stream) {
    const int kcap = k_eff <= 64 ? 64 : 256;
    uint32_t done = 0;
    while (done < nq) {
        const uint32_t left = nq - done;
        // queries per pass: 8 / 4 through the multi-query kernel, else 2 / 1 through the register-resident kernel
        int pass = left >= 2 && !f32_ ? 2 : 1;   // (F32 slabs: one fused kernel, one query per pass)
        bool mq = false;
        if (!f32_ && variant != 3 && variant != 1 && (!row_stride_ || row_stride_ == dim_ * 2)) {
            if (left >= 8 && kcap == 64 && scan_mq_supported((int)dim_, 8, kcap)) {
                pass = 8;
                mq = true;
            } else if (left >= 4 && scan_mq_supported((int)dim_, 4, kcap)) {
                pass = 4;
                mq = true;
            }
        }
        if (!mq && left >= 4 && scan_lds_bytes((int)dim_, 4, kcap) <= 150 * 1024 && (variant == 3)) pass = 4;
        int per_cu = 1;
        if (f32_) {
            ScanArgs probe = base_args(queries_dev, allow_dev);
            FSGPU_HIP(launch_scan_topk_f32(probe, kcap, 1, stream, &per_cu));
            per_cu = std::min(per_cu, 4);
        } else if (mq) {
            ScanArgs probe = base_args(queries_dev, allow_dev);
            FSGPU_HIP(launch_scan_mq(probe, pass, kcap, 1, stream, &per_cu));
        } else {
            per_cu = scan_occupancy_blocks_per_cu((int)dim_, pass, kcap, variant == 1);
        }
        // strided views (the MRL truncated scan reads a short prefix of every row): the small-dimension kernels fit 8
        // blocks per CU, but that many waves thrash — 4 per CU up to 64 dims and 2 beyond measured best (10M x 384 slab,
        // search_dims 64: 0.57 -> 0.44 ms per query; 32: 0.41 -> 0.33 ms; 128: 0.58 -> 0.55 ms)
        if (!f32_ && row_stride_ && row_stride_ != dim_ * 2) per_cu = std::min(per_cu, dim_ <= 64 ? 4 : 2);
        int grid = num_cus_ * per_cu;
        if (knobs().grid_blocks > 0) grid = knobs().grid_blocks;  // tuning experiments only
        const uint32_t ntiles_pass = (uint32_t)((nrows_ + (16 / pass) - 1) / (16 / pass));
        const int max_useful = (int)((ntiles_pass + 3) / 4);
        if (grid > max_useful) grid = max_useful;
        if (grid < 1) grid = 1;
        FSGPU_TRY(ws_partial_.reserve((size_t)pass * grid * k_eff * 8));
        ScanArgs a = base_args(queries_dev + (size_t)done * dim_, allow_dev);
        a.partial = static_cast<u64*>(ws_partial_.ptr);
        a.k = k_eff;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (profiling) {
            FSGPU_HIP(hipEventCreate(&e0));
            FSGPU_HIP(hipEventCreate(&e1));
            FSGPU_HIP(hipEventRecord(e0, stream));
        }
        if (f32_) FSGPU_HIP(launch_scan_topk_f32(a, kcap, grid, stream, nullptr));
        else if (mq) FSGPU_HIP(launch_scan_mq(a, pass, kcap, grid, stream, nullptr));
        else if (host_query_hint_ && nq == 1) FSGPU_HIP(launch_scan_topk_host_query(a, host_query_hint_, kcap, grid, stream));
        else FSGPU_HIP(launch_scan_topk(a, pass, kcap, grid, stream, variant == 1, variant == 2));
        if (profiling) {
            FSGPU_HIP(hipEventRecord(e1, stream));
            events_.emplace_back(e0, e1);
            profiled_rows_ += nrows_;
        }
        MergeArgs m;
        m.lists = a.partial;
        m.q_stride = (uint64_t)grid * k_eff;
        m.l_stride = (uint64_t)k_eff;
        m.out_packed = out_packed_dev ? out_packed_dev + (size_t)done * k_out : nullptr;
        m.nlists = (uint32_t)grid;
        m.list_len = k_eff;
        m.k = k_eff;
        m.out_stride = k_out;
        m.out_rows = out_rows_dev ? out_rows_dev + (size_t)done * k_out : nullptr;
        m.out_scores = out_scores_dev ? out_scores_dev + (size_t)done * k_out : nullptr;
        m.out_counts = out_counts_dev ? out_counts_dev + done : nullptr;
        FSGPU_HIP(launch_merge_topk(m, pass, stream));
        done += (uint32_t)pass;
    }
    return ok();
}

// Large-k / collect-all (search.rs:449-473) and dims that are not a multiple of 8: score every row,
// radix-sort the integer sortkeys, re-score the winners for their exact f32 bits (keeps NaN scores).
hipError_t VectorIndex::gather_dot_any(const ScanArgs& a, const uint32_t* rows, uint32_t n, float* out, hipStream_t stream) const {
    return f32_ ? launch_gather_dot_f32(a, rows, n, out, stream) : launch_gather_dot(a, rows, n, out, stream);
}

SearchError VectorIndex::general_search(const float* queries_dev, uint32_t nq, uint32_t k_out, uint32_t k_eff,
                                        const uint64_t* allow_dev, uint32_t* out_rows_dev, float* out_scores_dev,
                                        uint32_t* out_counts_dev, hipStream_t stream) {
    const size_t n = (size_t)nrows_;
    FSGPU_TRY(ws_keys_a_.reserve(n * 8));
    FSGPU_TRY(ws_keys_b_.reserve(n * 8));
    size_t tmp_bytes = 0;
    FSGPU_HIP(sort_keys_desc_temp_bytes(n, &tmp_bytes));
    FSGPU_TRY(ws_sort_tmp_.reserve(tmp_bytes));
    const uint32_t ntiles = (uint32_t)((nrows_ + 15) / 16);
    int grid = num_cus_ * 4;
    if (grid > (int)((ntiles + 3) / 4)) grid = (int)((ntiles + 3) / 4);
    if (grid < 1) grid = 1;
    FSGPU_HIP(hipMemsetAsync(out_rows_dev, 0xff, (size_t)nq * k_out * 4, stream));
    FSGPU_HIP(hipMemsetAsync(out_scores_dev, 0xff, (size_t)nq * k_out * 4, stream));
    for (uint32_t q = 0; q < nq; ++q) {
        ScanArgs a = base_args(queries_dev, allow_dev);
        u64* keys_a = static_cast<u64*>(ws_keys_a_.ptr);
        u64* keys_b = static_cast<u64*>(ws_keys_b_.ptr);
        if (f32_) FSGPU_HIP(launch_score_rows_f32(a, keys_a, (int)q, stream));
        else FSGPU_HIP(launch_score_rows(a, keys_a, (int)q, grid, stream));
        FSGPU_HIP(launch_packed_to_sortkey(keys_a, n, stream));
        FSGPU_HIP(sort_keys_desc(ws_sort_tmp_.ptr, ws_sort_tmp_.bytes, keys_a, keys_b, n, stream));
        uint32_t* rows_q = out_rows_dev + (size_t)q * k_out;
        FSGPU_HIP(launch_sorted_keys_to_rows(keys_b, k_eff, rows_q, out_counts_dev + q, stream));
        ScanArgs g = base_args(queries_dev + (size_t)q * dim_, nullptr);
        // rows beyond the count are 0xffffffff -> outside the shard -> left as padding
        FSGPU_HIP(gather_dot_any(g, rows_q, k_eff, out_scores_dev + (size_t)q * k_out, stream));
    }
    return ok();
}

// try_gather_filtered / scan_gather_positions (crates/frankensearch-index/src/search.rs:1114-1255): when a filter lets
// through fewer than 1/GATHER_SELECTIVITY_DIVISOR (= 50) of the rows, only those rows are scored (same dot, same order,
// same (score, row) selection), so the bytes read are cnt * dim * 2 instead of N * dim * 2.  `rows_dev` holds the
// allowed, live rows (global ids, ascending).
SearchError VectorIndex::gather_search(const float* queries_dev, uint32_t nq, uint32_t k, const uint32_t* rows_dev,
                                       uint32_t n, uint32_t* out_rows_dev, float* out_scores_dev,
                                       uint32_t* out_counts_dev, hipStream_t stream) {
    const uint32_t k_eff = std::min<uint32_t>(k, n);
    FSGPU_TRY(ws_gather_out_.reserve((size_t)n * 4));
    FSGPU_TRY(ws_keys_a_.reserve((size_t)n * 8));
    float* scores = static_cast<float*>(ws_gather_out_.ptr);
    u64* packed = static_cast<u64*>(ws_keys_a_.ptr);
    FSGPU_HIP(hipMemsetAsync(out_rows_dev, 0xff, (size_t)nq * k * 4, stream));
    FSGPU_HIP(hipMemsetAsync(out_scores_dev, 0xff, (size_t)nq * k * 4, stream));
    for (uint32_t q = 0; q < nq; ++q) {
        ScanArgs g = base_args(queries_dev + (size_t)q * dim_, nullptr);
        FSGPU_HIP(gather_dot_any(g, rows_dev, n, scores, stream));
        FSGPU_HIP(launch_pack_hits(rows_dev, scores, n, packed, stream));
        if (n <= 8192 && k_eff <= 256) {
            MergeArgs m;
            m.lists = packed;
            m.q_stride = n;
            m.l_stride = n;
            m.nlists = 1;
            m.list_len = n;
            m.k = k_eff;
            m.out_stride = k;
            m.out_rows = out_rows_dev + (size_t)q * k;
            m.out_scores = out_scores_dev + (size_t)q * k;
            m.out_counts = out_counts_dev + q;
            m.out_packed = nullptr;
            m.lists_sorted = 0;
            FSGPU_HIP(launch_merge_topk(m, 1, stream));
        } else {
            FSGPU_TRY(ws_keys_b_.reserve((size_t)n * 8));
            size_t tmp_bytes = 0;
            FSGPU_HIP(sort_keys_desc_temp_bytes(n, &tmp_bytes));
            FSGPU_TRY(ws_sort_tmp_.reserve(tmp_bytes));
            u64* keys_b = static_cast<u64*>(ws_keys_b_.ptr);
            FSGPU_HIP(launch_packed_to_sortkey(packed, n, stream));
            FSGPU_HIP(sort_keys_desc(ws_sort_tmp_.ptr, ws_sort_tmp_.bytes, packed, keys_b, n, stream));
            uint32_t* rows_q = out_rows_dev + (size_t)q * k;
            FSGPU_HIP(launch_sorted_keys_to_rows(keys_b, k_eff, rows_q, out_counts_dev + q, stream));
            FSGPU_HIP(gather_dot_any(g, rows_q, k_eff, out_scores_dev + (size_t)q * k, stream));
        }
    }
    return ok();
}

SearchError VectorIndex::search_top_k_device(const float* queries_dev, uint32_t nq, uint32_t query_len, uint32_t k,
                                             const uint64_t* allow_dev, uint32_t* out_rows_dev,
                                             float* out_scores_dev, uint32_t* out_counts_dev, hipStream_t stream) {
    FSGPU_TRY(ensure_query_dimension(query_len));
    if (nq == 0) return ok();
    FSGPU_HIP(hipSetDevice(device_));
    if (k == 0 || nrows_ == 0) {  // search.rs:437-439
        FSGPU_HIP(hipMemsetAsync(out_counts_dev, 0, (size_t)nq * 4, stream));
        return ok();
    }
    const uint32_t k_eff = (uint64_t)k < nrows_ ? k : (uint32_t)nrows_;
    if (dim_ % 8 == 0 && k_eff <= 256 && (!f32_ || dim_ <= 8192))
        return fused_search(queries_dev, nq, k, k_eff, allow_dev, out_rows_dev, out_scores_dev, out_counts_dev, nullptr,
                            stream);
    return general_search(queries_dev, nq, k, k_eff, allow_dev, out_rows_dev, out_scores_dev, out_counts_dev, stream);
}

SearchError VectorIndex::search_top_k(const float* queries, uint32_t nq, uint32_t query_len, uint32_t k,
                                      const uint64_t* allow, uint32_t* out_rows, float* out_scores,
                                      uint32_t* out_counts, const uint64_t* allow_resident_dev) {
    FSGPU_TRY(ensure_query_dimension(query_len));
    if (nq == 0) return ok();
    if (k == 0 || nrows_ == 0) {
        for (uint32_t q = 0; q < nq; ++q) out_counts[q] = 0;
        return ok();
    }
    FSGPU_HIP(hipSetDevice(device_));
    const size_t qbytes = (size_t)nq * dim_ * 4;
    FSGPU_TRY(ws_queries_.reserve(qbytes));
    // A lone query without a filter: the two halves below (certified int8 pass / staged filter path / exact kernels with the query in
    // the argument block), begun and ended at once.
    if (nq == 1 && !allow) {
        FSGPU_TRY(lone_exact_begin(queries, k));
        return lone_exact_end(out_rows, out_scores, out_counts);
    }
    // Latency path (a few queries, no filter): the queries go through a pinned staging block (true DMA instead of the
    // runtime's pageable-copy staging) and the last merge writes the hits straight into pinned host memory, so the call is
    // one H2D copy, the kernels and one stream synchronisation — no D2H copies (they cost ~90 us per call, measured).
    const size_t io_need = qbytes + (size_t)nq * k * 8 + (size_t)nq * 4 + 256;
    if (!allow && io_need <= kPinnedIoBytes && pinned_io() != nullptr) {
        unsigned char* io = static_cast<unsigned char*>(io_host_);
        float* q_pin = reinterpret_cast<float*>(io);
        uint32_t* rows_pin = reinterpret_cast<uint32_t*>(io + ((qbytes + 63) & ~(size_t)63));
        float* scores_pin = reinterpret_cast<float*>(rows_pin + (size_t)nq * k);
        uint32_t* counts_pin = reinterpret_cast<uint32_t*>(scores_pin + (size_t)nq * k);
        // opted in (fsgpu_index_set_int8_latency): the same hits through the int8 filter + exact re-score — half the bytes of the
        // exact kernel's pass; anything that path does not cover falls through to the exact kernels inside it
        const bool via_filter = int8_latency && batched_filter != 1 && !i8f_disabled_ && k <= 64 && nq <= 16 && !f32_ &&
                                !(row_stride_ && row_stride_ != dim_ * 2) && nrows_ >= 4 * 8192ull;
        std::memcpy(q_pin, queries, qbytes);
        FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, q_pin, qbytes, hipMemcpyHostToDevice, stream_));
        if (via_filter) {
            uint32_t fb = 0;
            FSGPU_TRY(search_top_k_batched_device(static_cast<const float*>(ws_queries_.ptr), nq, query_len, k, nullptr, rows_pin,
                                                  scores_pin, counts_pin, stream_, &fb));
        } else {
            FSGPU_TRY(search_top_k_device(static_cast<const float*>(ws_queries_.ptr), nq, query_len, k, nullptr, rows_pin,
                                          scores_pin, counts_pin, stream_));
        }
        FSGPU_HIP(hipStreamSynchronize(stream_));
        std::memcpy(out_rows, rows_pin, (size_t)nq * k * 4);
        std::memcpy(out_scores, scores_pin, (size_t)nq * k * 4);
        std::memcpy(out_counts, counts_pin, (size_t)nq * 4);
        return ok();
    }
    FSGPU_TRY(ws_rows_.reserve((size_t)nq * k * 4));
    FSGPU_TRY(ws_scores_.reserve((size_t)nq * k * 4));
    FSGPU_TRY(ws_counts_.reserve((size_t)nq * 4));
    FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, queries, qbytes, hipMemcpyHostToDevice, stream_));
    const uint64_t* allow_dev = nullptr;
    bool gathered = false;
    if (allow) {
        const size_t words = (size_t)((nrows_ + 63) / 64);
        // selective filter: score only the allowed rows (GATHER_SELECTIVITY_DIVISOR = 50, search.rs:1667,1136-1139);
        // needs the tombstone bitmap on the host (absent only for indexes created over a device-resident bitmap)
        const bool live_known = live_dev_ == nullptr || !live_host_.empty();
        uint64_t cnt = 0;
        if (live_known && variant != 6) {
            for (size_t w = 0; w < words; ++w) {
                uint64_t bitsw = allow[w];
                if (!live_host_.empty()) bitsw &= live_host_[w];
                if (w + 1 == words && (nrows_ & 63)) bitsw &= (1ull << (nrows_ & 63)) - 1ull;
                cnt += (uint64_t)__builtin_popcountll(bitsw);
            }
        }
        if (live_known && variant != 6 && cnt > 0 && cnt * 50 < nrows_) {
            std::vector<uint32_t> rows_host;
            rows_host.reserve((size_t)cnt);
            for (size_t w = 0; w < words; ++w) {
                uint64_t bitsw = allow[w];
                if (!live_host_.empty()) bitsw &= live_host_[w];
                if (w + 1 == words && (nrows_ & 63)) bitsw &= (1ull << (nrows_ & 63)) - 1ull;
                while (bitsw) {
                    const int b = __builtin_ctzll(bitsw);
                    rows_host.push_back((uint32_t)(row_base_ + w * 64 + (size_t)b));
                    bitsw &= bitsw - 1;
                }
            }
            FSGPU_TRY(ws_gather_rows_.reserve(rows_host.size() * 4));
            FSGPU_HIP(hipMemcpyAsync(ws_gather_rows_.ptr, rows_host.data(), rows_host.size() * 4, hipMemcpyHostToDevice, stream_));
            FSGPU_HIP(hipStreamSynchronize(stream_));  // rows_host is freed at the end of this block
            FSGPU_TRY(gather_search(static_cast<const float*>(ws_queries_.ptr), nq, k,
                                    static_cast<const uint32_t*>(ws_gather_rows_.ptr), (uint32_t)rows_host.size(),
                                    static_cast<uint32_t*>(ws_rows_.ptr), static_cast<float*>(ws_scores_.ptr),
                                    static_cast<uint32_t*>(ws_counts_.ptr), stream_));
            gathered = true;
            ++filter_gathered;
        } else if (allow_resident_dev) {
            allow_dev = allow_resident_dev;   // uploaded once, when the filter was made resident
            ++filter_scanned;
        } else {
            FSGPU_TRY(ws_allow_.reserve(words * 8));
            FSGPU_HIP(hipMemcpyAsync(ws_allow_.ptr, allow, words * 8, hipMemcpyHostToDevice, stream_));
            allow_dev = static_cast<const uint64_t*>(ws_allow_.ptr);
            ++filter_scanned;
        }
    }
    if (!gathered)
        FSGPU_TRY(search_top_k_device(static_cast<const float*>(ws_queries_.ptr), nq, query_len, k, allow_dev,
                                      static_cast<uint32_t*>(ws_rows_.ptr), static_cast<float*>(ws_scores_.ptr),
                                      static_cast<uint32_t*>(ws_counts_.ptr), stream_));
    FSGPU_HIP(hipMemcpyAsync(out_rows, ws_rows_.ptr, (size_t)nq * k * 4, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipMemcpyAsync(out_scores, ws_scores_.ptr, (size_t)nq * k * 4, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipMemcpyAsync(out_counts, ws_counts_.ptr, (size_t)nq * 4, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipStreamSynchronize(stream_));
    return ok();
}

// See vector_index.hpp.  Same results as the exact kernels, bit for bit: the candidates are re-scored in the reference's order
// (gather_dot_kernel) and the certificate is the int8 filter's proven bound (prepare_queries_i8_filter_kernel: the quantised query IS
// quantize_i8_query's, the slab IS quantize_f16_le_bytes_to_i8's) applied to the list's own scores: every true top-k row has
// idot >= idot_k - 2 delta, and the kept list is exactly the 256 largest idot.
SearchError VectorIndex::certified_i8_lone_query(const float* query, uint32_t k, uint32_t* out_rows, float* out_scores,
                                                 uint32_t* out_count, bool* certified) {
    *certified = false;
    bool enqueued = false;
    FSGPU_TRY(certified_i8_enqueue(query, k, &enqueued));
    if (!enqueued) return ok();
    return certified_i8_check(out_rows, out_scores, out_count, certified);
}

// Four launches behind one another, no copy (the query and every result live in the pinned staging block, which the kernels address
// directly); nothing is waited for:
//   prepare   the query quantised as the filter does + its proven bound delta
//   scan      the int8 copy, every block keeps its LK best (integer score, row) entries
//   cut       the best score any block may have DROPPED: the maximum over the full lists' last entries
//   finish    select_kernel: tau = (k-th best approximate score) - 2 delta, the entries at or above it re-scored in the reference's
//             order from the f16 slab, the k best exact entries out
// *enqueued = false: a shape the pass does not cover, nothing was launched.
SearchError VectorIndex::certified_i8_enqueue(const float* query, uint32_t k, bool* enqueued) {
    *enqueued = false;
    constexpr uint32_t LK = 32;
    const uint32_t k_eff = (uint32_t)std::min<uint64_t>(k, nrows_);
    if (k_eff == 0 || k_eff > LK || nrows_ < 4096 || (dim_ & 7) || !scan_i8_fused_supported((int)dim_, 64) || pinned_io() == nullptr) return ok();
    const size_t qbytes = (size_t)dim_ * 4;
    const size_t o_out = (qbytes + 255) & ~(size_t)255, o_flags = (o_out + (size_t)k * 8 + 4 + 255) & ~(size_t)255;
    if (o_flags + 64 > kPinnedIoBytes) return ok();
    unsigned char* io = static_cast<unsigned char*>(io_host_);
    FSGPU_TRY(ws_i8_query_.reserve(dim_));
    std::memcpy(io, query, qbytes);
    const float* q_pin = reinterpret_cast<const float*>(io);
    float* delta_pin = reinterpret_cast<float*>(io + o_flags);
    float* tau_pin = delta_pin + 1;
    float* cut_pin = delta_pin + 2;
    uint32_t* ncand_pin = reinterpret_cast<uint32_t*>(delta_pin + 3);
    uint32_t* overflow_pin = reinterpret_cast<uint32_t*>(delta_pin + 4);
    *overflow_pin = 0;
    *ncand_pin = 0;
    FSGPU_TRY(prepare_filter_queries(q_pin, 1, 1, dim_, ws_i8_query_.ptr, delta_pin, nullptr, stream_));
    ScanArgs a = base_args(q_pin, nullptr);
    int per_cu = 1;
    FSGPU_HIP(launch_scan_i8(a, filter_slab(), ws_i8_query_.ptr, 64, 1, stream_, &per_cu));
    (void)per_cu;   // one block per CU: 256 lists x 32 entries are ONE pass of the finish (8,192 entries)
    int grid = num_cus_;
    const int max_useful = (int)(((nrows_ + 15) / 16 + 3) / 4);
    grid = std::max(1, std::min(grid, max_useful));
    FSGPU_TRY(ws_partial_.reserve((size_t)grid * LK * 8));
    a.partial = static_cast<u64*>(ws_partial_.ptr);
    a.k = LK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (profiling) {
        FSGPU_HIP(hipEventCreate(&e0));
        FSGPU_HIP(hipEventCreate(&e1));
        FSGPU_HIP(hipEventRecord(e0, stream_));
    }
    FSGPU_HIP(launch_scan_i8(a, filter_slab(), ws_i8_query_.ptr, 64, grid, stream_, nullptr));
    if (profiling) {
        FSGPU_HIP(hipEventRecord(e1, stream_));
        events_.emplace_back(e0, e1);
        profiled_rows_ += nrows_;
        profiled_elem_bytes_ = 1;
    }
    FSGPU_HIP(launch_list_cut(a.partial, (uint32_t)grid, LK, cut_pin, stream_));
    SelectArgs f{};
    f.lists = a.partial;
    f.q_stride = (uint64_t)grid * LK;
    f.l_stride = LK;
    f.nlists = (uint32_t)grid;
    f.list_len = LK;
    f.k = k_eff;
    f.delta = delta_pin;
    f.tau_out = tau_pin;
    f.cand_counts = ncand_pin;
    f.overflow = overflow_pin;
    f.slab = slab_dev_;
    f.queries = q_pin;
    f.dim = dim_;
    f.nrows = (uint32_t)nrows_;
    f.row_base = (uint32_t)row_base_;
    f.row_stride = (row_stride_ && row_stride_ != dim_ * 2) ? row_stride_ : 0;
    f.hreduce = hreduce;
    f.k_out = k_eff;
    f.out_stride = k;
    f.out_rows = reinterpret_cast<uint32_t*>(io + o_out);
    f.out_scores = reinterpret_cast<float*>(io + o_out + (size_t)k * 4);
    f.out_counts = reinterpret_cast<uint32_t*>(io + o_out + (size_t)k * 8);
    FSGPU_HIP(launch_select(f, 1, stream_));
    cert_k_ = k;
    *enqueued = true;
    return ok();
}

// The other half: ONE synchronisation, then the certificate.  The answer is the exact search's when every row whose approximate score
// reaches tau was in some list: cut < tau (a list that is not full dropped nothing), no more candidates than the finish holds,
// delta >= 0.  Otherwise nothing is written and the caller's staged path answers.
SearchError VectorIndex::certified_i8_check(uint32_t* out_rows, float* out_scores, uint32_t* out_count, bool* certified) {
    *certified = false;
    const uint32_t k = cert_k_;
    const uint32_t k_eff = (uint32_t)std::min<uint64_t>(k, nrows_);
    const size_t qbytes = (size_t)dim_ * 4;
    const size_t o_out = (qbytes + 255) & ~(size_t)255, o_flags = (o_out + (size_t)k * 8 + 4 + 255) & ~(size_t)255;
    unsigned char* io = static_cast<unsigned char*>(io_host_);
    const float* delta_pin = reinterpret_cast<const float*>(io + o_flags);
    const uint32_t* ncand_pin = reinterpret_cast<const uint32_t*>(delta_pin + 3);
    const uint32_t* overflow_pin = reinterpret_cast<const uint32_t*>(delta_pin + 4);
    const uint32_t* rows_pin = reinterpret_cast<const uint32_t*>(io + o_out);
    const float* scores_pin = reinterpret_cast<const float*>(io + o_out + (size_t)k * 4);
    const uint32_t* count_pin = reinterpret_cast<const uint32_t*>(io + o_out + (size_t)k * 8);
    FSGPU_HIP(hipSetDevice(device_));
    FSGPU_HIP(hipStreamSynchronize(stream_));
    const float delta = delta_pin[0], tau = delta_pin[1], cut = delta_pin[2];
    if (!(delta >= 0.f)) return ok();   // a query the bound cannot cover (zero, non-finite, a slab with non-finite values)
    if (*overflow_pin != 0 || *ncand_pin > kSelectPool) return ok();   // more rows within the margin than the finish re-scores
    if (!(cut < tau)) return ok();      // a block may have dropped a row within the margin (NaN compares false: not certified)
    if (*count_pin < k_eff) return ok();
    std::memcpy(out_rows, rows_pin, (size_t)k * 4);
    std::memcpy(out_scores, scores_pin, (size_t)k * 4);
    *out_count = *count_pin;
    ++i8f_queries;
    *certified = true;
    return ok();
}

// ---- a lone query in two halves (vector_index.hpp) ------------------------------------------------------------------------
//
// search_top_k for ONE host query without a filter: begin enqueues on the index's own stream and returns, end waits and writes the
// hits.  What begin picks — the certified int8 pass, the staged filter path, the exact kernels — is what search_top_k always picked
// for a lone caller; a row-sharded handle begins the query on every shard before it ends any.
SearchError VectorIndex::lone_exact_begin(const float* query, uint32_t k) {
    lone_ = LoneState{};
    lone_.query = query;
    lone_.k = k;
    if (k == 0 || nrows_ == 0) {
        lone_.kind = kLoneEmpty;
        return ok();
    }
    FSGPU_HIP(hipSetDevice(device_));
    const size_t qbytes = (size_t)dim_ * 4;
    FSGPU_TRY(ws_queries_.reserve(qbytes));
    const size_t io_need = qbytes + (size_t)k * 8 + 4 + 256;
    if (io_need > kPinnedIoBytes || pinned_io() == nullptr) {
        lone_.kind = kLoneUnpinned;
        return ok();
    }
    unsigned char* io = static_cast<unsigned char*>(io_host_);
    float* q_pin = reinterpret_cast<float*>(io);
    uint32_t* rows_pin = reinterpret_cast<uint32_t*>(io + ((qbytes + 63) & ~(size_t)63));
    float* scores_pin = reinterpret_cast<float*>(rows_pin + k);
    uint32_t* counts_pin = reinterpret_cast<uint32_t*>(scores_pin + k);
    // opted in (fsgpu_index_set_int8_latency): the same hits through the int8 filter + exact re-score — half the bytes of the
    // exact kernel's pass; anything that path does not cover falls through to the exact kernels inside it
    const bool i8_shape = batched_filter != 1 && !i8f_disabled_ && !f32_ && !(row_stride_ && row_stride_ != dim_ * 2) && nrows_ >= 4 * 8192ull;
    const bool via_filter = int8_latency && !exact_only_ && i8_shape && k <= 64;
    // By default (round 5): an index that already HOLDS the int8 copy and its statistics — some batched search built them — answers a
    // lone query with the certified pass over that copy too: the rows and score bits of the exact kernels from half the bytes
    // (10M x 384: p50 0.67 against 1.27 ms; 1M: 0.12 against 0.17).  Nothing is built for it, an uncertified query goes to the exact
    // kernels, and fsgpu_search_topk_exact keeps those kernels reachable as they are.
    const bool by_default = !via_filter && !exact_only_ && i8_shape && k <= 32 && filter_ready() && variant == 0;
    // a lone query of a fused-kernel shape travels in the scan kernel's argument block: no H2D copy in front of the scan
    const uint32_t k_lat = (uint64_t)k < nrows_ ? k : (uint32_t)nrows_;
    const bool in_kernarg = !via_filter && !f32_ && dim_ % 8 == 0 && k_lat <= 256 && variant == 0 &&
                            scan_kernarg_query_supported((int)dim_, k_lat <= 64 ? 64 : 256);
    if ((via_filter && k <= 32 && filter_ready() && variant == 0) || by_default) {
        // A failed certificate costs a whole pass over the int8 copy (a query with more rows inside the margin than the finish
        // holds, or so many in one block's share that its list dropped one), so the single pass backs off: after a failure the
        // next 1, 2, 4 ... 64 lone queries go straight to the staged path (the exact kernels when the pass is the default); a
        // success resets it.
        // (the pass reads the query from the pinned staging block itself: no H2D copy in front of it)
        if (cert_skip_ > 0) {
            --cert_skip_;
        } else {
            bool enqueued = false;
            FSGPU_TRY(certified_i8_enqueue(query, k, &enqueued));
            if (enqueued) {
                lone_.kind = kLoneCertified;
                lone_.staged_behind = via_filter;
                return ok();
            }
        }
    }
    if (via_filter && async_state_[0] != 0 && async_state_[1] != 0) {   // both tickets of the staged path are out: end() answers, blocking
        lone_.kind = kLoneStagedBlocking;
        return ok();
    }
    if (!in_kernarg) {
        std::memcpy(q_pin, query, qbytes);
        FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, q_pin, qbytes, hipMemcpyHostToDevice, stream_));
    }
    if (via_filter) {
        FSGPU_TRY(search_top_k_batched_device_begin(static_cast<const float*>(ws_queries_.ptr), 1, dim_, k, nullptr, rows_pin, scores_pin,
                                                    counts_pin, stream_, nullptr, &lone_.ticket));
        lone_.kind = kLoneStaged;
    } else {
        host_query_hint_ = in_kernarg ? query : nullptr;
        const SearchError se = search_top_k_device(static_cast<const float*>(ws_queries_.ptr), 1, dim_, k, nullptr, rows_pin, scores_pin,
                                                   counts_pin, stream_);
        host_query_hint_ = nullptr;
        FSGPU_TRY(se);
        lone_.kind = kLoneExact;
    }
    return ok();
}

SearchError VectorIndex::lone_exact_end(uint32_t* out_rows, float* out_scores, uint32_t* out_count) {
    const LoneState st = lone_;
    lone_ = LoneState{};
    const uint32_t k = st.k;
    if (st.kind == kLoneEmpty) {
        *out_count = 0;
        return ok();
    }
    if (st.kind == kLoneNone) return make_error(FSGPU_ERR_INVALID_CONFIG, "no lone query was begun on this index");
    FSGPU_HIP(hipSetDevice(device_));
    const size_t qbytes = (size_t)dim_ * 4;
    if (st.kind == kLoneUnpinned) {   // no pinned staging block: pageable copies either side of the exact kernels
        FSGPU_TRY(ws_rows_.reserve((size_t)k * 4));
        FSGPU_TRY(ws_scores_.reserve((size_t)k * 4));
        FSGPU_TRY(ws_counts_.reserve(4));
        FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, st.query, qbytes, hipMemcpyHostToDevice, stream_));
        FSGPU_TRY(search_top_k_device(static_cast<const float*>(ws_queries_.ptr), 1, dim_, k, nullptr, static_cast<uint32_t*>(ws_rows_.ptr),
                                      static_cast<float*>(ws_scores_.ptr), static_cast<uint32_t*>(ws_counts_.ptr), stream_));
        FSGPU_HIP(hipMemcpyAsync(out_rows, ws_rows_.ptr, (size_t)k * 4, hipMemcpyDeviceToHost, stream_));
        FSGPU_HIP(hipMemcpyAsync(out_scores, ws_scores_.ptr, (size_t)k * 4, hipMemcpyDeviceToHost, stream_));
        FSGPU_HIP(hipMemcpyAsync(out_count, ws_counts_.ptr, 4, hipMemcpyDeviceToHost, stream_));
        FSGPU_HIP(hipStreamSynchronize(stream_));
        return ok();
    }
    unsigned char* io = static_cast<unsigned char*>(io_host_);
    float* q_pin = reinterpret_cast<float*>(io);
    uint32_t* rows_pin = reinterpret_cast<uint32_t*>(io + ((qbytes + 63) & ~(size_t)63));
    float* scores_pin = reinterpret_cast<float*>(rows_pin + k);
    uint32_t* counts_pin = reinterpret_cast<uint32_t*>(scores_pin + k);
    bool staged_blocking = st.kind == kLoneStagedBlocking;
    if (st.kind == kLoneCertified) {
        bool certified = false;
        FSGPU_TRY(certified_i8_check(out_rows, out_scores, out_count, &certified));
        if (certified) {
            cert_backoff_ = 0;
            return ok();
        }
        cert_backoff_ = cert_backoff_ ? std::min<uint32_t>(cert_backoff_ * 2, 64) : 1;
        cert_skip_ = cert_backoff_;
        if (st.staged_behind) {
            staged_blocking = true;
        } else {   // the pass was the default, not an opt-in: the exact kernels answer
            const uint32_t k_lat = (uint64_t)k < nrows_ ? k : (uint32_t)nrows_;
            const bool in_kernarg = !f32_ && dim_ % 8 == 0 && k_lat <= 256 && variant == 0 && scan_kernarg_query_supported((int)dim_, k_lat <= 64 ? 64 : 256);
            if (!in_kernarg) {
                std::memcpy(q_pin, st.query, qbytes);
                FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, q_pin, qbytes, hipMemcpyHostToDevice, stream_));
            }
            host_query_hint_ = in_kernarg ? st.query : nullptr;
            const SearchError se = search_top_k_device(static_cast<const float*>(ws_queries_.ptr), 1, dim_, k, nullptr, rows_pin, scores_pin,
                                                       counts_pin, stream_);
            host_query_hint_ = nullptr;
            FSGPU_TRY(se);
        }
    }
    if (staged_blocking) {   // the staged filter path, in one piece
        std::memcpy(q_pin, st.query, qbytes);
        FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, q_pin, qbytes, hipMemcpyHostToDevice, stream_));
        uint32_t fb = 0;
        FSGPU_TRY(search_top_k_batched_device(static_cast<const float*>(ws_queries_.ptr), 1, dim_, k, nullptr, rows_pin, scores_pin, counts_pin,
                                              stream_, &fb));
    } else if (st.kind == kLoneStaged) {
        uint32_t fb = 0;
        FSGPU_TRY(search_top_k_batched_device_end(st.ticket, &fb));
    }
    FSGPU_HIP(hipStreamSynchronize(stream_));
    std::memcpy(out_rows, rows_pin, (size_t)k * 4);
    std::memcpy(out_scores, scores_pin, (size_t)k * 4);
    *out_count = *counts_pin;
    return ok();
}

SearchError VectorIndex::gather_dot(const float* query, uint32_t query_len, const uint32_t* rows, uint32_t n,
                                    float* out) {
    FSGPU_TRY(ensure_query_dimension(query_len));
    if (n == 0) return ok();
    for (uint32_t i = 0; i < n; ++i)
        if (rows[i] < row_base_ || rows[i] - row_base_ >= nrows_)
            return make_error(FSGPU_ERR_INVALID_CONFIG, "row index out of range for dot_query_at");
    FSGPU_HIP(hipSetDevice(device_));
    FSGPU_TRY(ws_queries_.reserve((size_t)dim_ * 4));
    // Latency path (quality_scores_for_hits re-scores a few dozen rows per query): query through the pinned staging block, the
    // row ids read and the dots written by the kernel straight in pinned host memory — one H2D copy, one launch, one synchronisation.
    const size_t qbytes = (size_t)dim_ * 4, qpad = (qbytes + 63) & ~(size_t)63;
    if (qpad + (size_t)n * 8 + 128 <= kPinnedIoBytes && pinned_io() != nullptr) {
        unsigned char* io = static_cast<unsigned char*>(io_host_);
        float* q_pin = reinterpret_cast<float*>(io);
        uint32_t* rows_pin = reinterpret_cast<uint32_t*>(io + qpad);
        float* out_pin = reinterpret_cast<float*>(rows_pin + n);
        std::memcpy(q_pin, query, qbytes);
        std::memcpy(rows_pin, rows, (size_t)n * 4);
        FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, q_pin, qbytes, hipMemcpyHostToDevice, stream_));
        ScanArgs g = base_args(static_cast<const float*>(ws_queries_.ptr), nullptr);
        FSGPU_HIP(gather_dot_any(g, rows_pin, n, out_pin, stream_));
        FSGPU_HIP(hipStreamSynchronize(stream_));
        std::memcpy(out, out_pin, (size_t)n * 4);
        return ok();
    }
    FSGPU_TRY(ws_gather_rows_.reserve((size_t)n * 4));
    FSGPU_TRY(ws_gather_out_.reserve((size_t)n * 4));
    FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, query, (size_t)dim_ * 4, hipMemcpyHostToDevice, stream_));
    FSGPU_HIP(hipMemcpyAsync(ws_gather_rows_.ptr, rows, (size_t)n * 4, hipMemcpyHostToDevice, stream_));
    ScanArgs a = base_args(static_cast<const float*>(ws_queries_.ptr), nullptr);
    FSGPU_HIP(gather_dot_any(a, static_cast<const uint32_t*>(ws_gather_rows_.ptr), n,
                             static_cast<float*>(ws_gather_out_.ptr), stream_));
    FSGPU_HIP(hipMemcpyAsync(out, ws_gather_out_.ptr, (size_t)n * 4, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipStreamSynchronize(stream_));
    return ok();
}

// Batched search on the matrix cores; see mfma_scan.hip for the error bound that makes the result exact.
SearchError VectorIndex::search_top_k_batched_device(const float* queries_dev, uint32_t nq, uint32_t query_len,
                                                     uint32_t k, const uint64_t* allow_dev, uint32_t* out_rows_dev,
                                                     float* out_scores_dev, uint32_t* out_counts_dev,
                                                     hipStream_t stream, uint32_t* fallbacks, uint64_t* out_packed_dev) {
    // Which approximate scores filter the slab: the int8 slab on the integer matrix cores (half the bytes, half the MFMA
    // instructions of the f16 filter; a wider proven margin) unless this index has shown that its margin lets too many rows
    // through (outlier dimensions stretch the corpus-wide int8 scale), the caller forced one, or the shape is not covered.
    const bool strided = row_stride_ && row_stride_ != dim_ * 2;
    bool i8f = batched_filter != 1 && (batched_filter == 2 || !i8f_disabled_) && !f32_ && !strided && variant == 0 && knobs().filter != 1 &&
               scan_mfma_supported((int)dim_) && k >= 1 && k <= 64 && nrows_ >= 4 * 8192ull;
    // a few queries are not worth BUILDING the int8 copy for; once it exists (or the host asked for the int8 latency path) they
    // are answered from it too: one query 0.88 ms against 1.29 ms on the exact kernel at 10M x 384
    if (batched_filter == 0 && knobs().filter == 0 && nq < 16 && !filter_ready() && !int8_latency) i8f = false;
    if (i8f && !filter_ready()) {
        // the int8 copy of the slab (half its size again; rotated when the slab has outlier channels) is built on first use; no room
        // for it: the f16 filter needs none
        FSGPU_TRY(ensure_filter_copy(stream));
        if (!filter_ready()) i8f = false;
    }
    if (i8f) {
        uint32_t refiltered = 0;
        SearchError e = batched_impl(queries_dev, nq, query_len, k, allow_dev, out_rows_dev, out_scores_dev, out_counts_dev, stream,
                                     fallbacks, out_packed_dev, 0, 0, true, &refiltered);
        if (e.ok() && async_want_ >= 0 && async_state_[async_want_] == 1) {
            async_i8f_[async_want_] = true;   // (the bookkeeping below happens in _end, once the verdicts are in)
            return e;
        }
        if (e.ok()) i8f_account(nq, refiltered);
        return e;
    }
    return batched_impl(queries_dev, nq, query_len, k, allow_dev, out_rows_dev, out_scores_dev, out_counts_dev, stream,
                        fallbacks, out_packed_dev, 0, 0, false, nullptr);
}

// The int8 copy of the slab and its statistics (what the certified lone-query pass and the int8 filter read), built NOW instead of by
// the first batched search: a row-sharded handle switches its shards to the int8 latency path in one go.
SearchError VectorIndex::prepare_int8_latency() {
    const bool strided = row_stride_ && row_stride_ != dim_ * 2;
    if (f32_ || strided || nrows_ < 4 * 8192ull || !scan_mfma_supported((int)dim_) || i8f_disabled_) return ok();
    FSGPU_HIP(hipSetDevice(device_));
    FSGPU_TRY(ensure_filter_copy(stream_));
    FSGPU_HIP(hipStreamSynchronize(stream_));
    return ok();
}

// What a batch's verdicts teach the index about its int8 filter.
void VectorIndex::i8f_account(uint32_t nq, uint32_t refiltered) {
    {
        {
            i8f_queries += nq;
            i8f_refiltered += refiltered;
            // More than 1/8 of a batch uncertified.  What overflows on a corpus with a dense score tail (outlier dimensions, big
            // clusters) is the MAIN pass's lists: the rows within the margin of the k-th best number a few hundred whatever the
            // corpus size, but the main pass runs on the threshold of a 1/25 sample and lets N / RB times as many through
            // (scripts/r04/i8_bound_study.py).  So the first answer is a larger second sample for this index — 2 x, then 4 x: half /
            // a quarter as many survivors for +0.1 / +0.3 ms of sampling per 512-query pass at 10M rows — and only an index that
            // still hands an eighth of its batches on twice in a row at 4 x goes to the f16 filter.
            // (a handful of leftovers already costs a pass of their own over the f16 slab — as much as the 512 queries they came
            // with —, so the sample grows as soon as more than 1 in 64 of a wide batch is handed on; the filter is given up only
            // when an eighth still is, twice in a row, at the largest sample)
            // (measured on the anisotropic / Zipf corpus at 10M rows, scripts/r04/outlier_census.py: 150 of 1,024 queries handed on at
            // the base sample, 69 at 2 x, 580 at 4 x — the larger sample's own selection then overflows its candidate pool —, so 2 x is
            // as far as it goes)
            if (nq >= 256 && i8f_sample_boost_ < 2 && (uint64_t)refiltered * 64 > nq) {
                i8f_sample_boost_ *= 2;
                i8f_strikes_ = 0;
            } else if (nq >= 16 && (uint64_t)refiltered * 8 > nq) {
                if (++i8f_strikes_ >= 2 && batched_filter == 0 && knobs().filter == 0) i8f_disabled_ = true;
            } else {
                i8f_strikes_ = 0;
            }
        }
    }
}

SearchError VectorIndex::int8_filter_bound(const float* queries, uint32_t nq, uint32_t query_len, float* out_delta,
                                           float* out_query_scale, float* out_slab_scale, int8_t* out_queries_i8, int8_t* out_slab_i8) {
    FSGPU_TRY(ensure_query_dimension(query_len));
    if (f32_ || (row_stride_ && row_stride_ != dim_ * 2) || nrows_ == 0 || nrows_ > 0xffffffffull)
        return make_error(FSGPU_ERR_INVALID_CONFIG, "the int8 filter serves f16 slabs only");
    FSGPU_HIP(hipSetDevice(device_));
    FSGPU_TRY(ensure_filter_copy(stream_, true));
    if (!filter_ready()) return make_error(FSGPU_ERR_DEVICE, "no room for the int8 copy of the slab");
    float slab_max = 0.f;
    FSGPU_HIP(hipMemcpyAsync(&slab_max, filter_max(), 4, hipMemcpyDeviceToHost, stream_));
    std::vector<float> unit(nq, 0.f);
    if (nq) {
        FSGPU_TRY(ws_queries_.reserve((size_t)nq * dim_ * 4));
        FSGPU_TRY(mf_qh_.reserve((size_t)nq * dim_ * 2));
        FSGPU_TRY(mf_delta_.reserve((size_t)nq * 8));
        float* delta_dev = static_cast<float*>(mf_delta_.ptr);
        FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, queries, (size_t)nq * dim_ * 4, hipMemcpyHostToDevice, stream_));
        FSGPU_TRY(prepare_filter_queries(static_cast<const float*>(ws_queries_.ptr), nq, nq, dim_, mf_qh_.ptr, delta_dev, delta_dev + nq, stream_));
        if (out_delta) FSGPU_HIP(hipMemcpyAsync(out_delta, delta_dev, (size_t)nq * 4, hipMemcpyDeviceToHost, stream_));
        FSGPU_HIP(hipMemcpyAsync(unit.data(), delta_dev + nq, (size_t)nq * 4, hipMemcpyDeviceToHost, stream_));
        if (out_queries_i8) FSGPU_HIP(hipMemcpyAsync(out_queries_i8, mf_qh_.ptr, (size_t)nq * dim_, hipMemcpyDeviceToHost, stream_));
    }
    if (out_slab_i8) FSGPU_HIP(hipMemcpyAsync(out_slab_i8, filter_slab(), (size_t)nrows_ * dim_, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipStreamSynchronize(stream_));
    const float slab_scale = slab_max > 0.f ? 127.0f / slab_max : 0.f;
    if (out_slab_scale) *out_slab_scale = slab_scale;
    if (out_query_scale)
        for (uint32_t i = 0; i < nq; ++i) {
            if (i8f_rot_) {   // the scale of the ROTATED query: integer-score units per exact-score unit / the slab's scale
                out_query_scale[i] = slab_scale > 0.f ? unit[i] / slab_scale : 0.f;
                continue;
            }
            float m = 0.f;   // quantize_i8_query's scale, as the kernel computes it
            for (uint32_t d = 0; d < dim_; ++d) m = std::fmax(m, std::fabs(queries[(size_t)i * dim_ + d]));
            out_query_scale[i] = m > 0.f ? 127.0f / m : 0.f;
        }
    return ok();
}

// ---- the int8 filter's copy of the slab --------------------------------------------------------------------------------------
//
// Unrotated (the default for slabs without outlier channels): the reference's own int8 slab (quantize_f16_le_bytes_to_i8_generic),
// shared with the int8 two-pass search, + its statistics.  Rotated (round 5): a copy of its own — rows R x quantised with THEIR
// max-abs — for slabs whose largest element is far above what an even spread of a row's norm over its dimensions gives: the
// corpus-wide scale then wastes the int8 range on a few channels, and the filter's margin (fixed in integer units) is several times
// wider in cosine units than it has to be (int8_kernels.hip; scripts/r05/rotation_bound_study.py: 0.063 -> 0.021 on the bench's
// outlier corpus, 1,023 -> 83 rows within the margin of the k-th best).  Decided once per index, on first use.
namespace {
// a fixed random orthogonal matrix (seeded; modified Gram-Schmidt twice, in double) as its TRANSPOSE [j][d], and |R^T R - I|_F
void make_rotation(uint32_t dim, std::vector<double>& rt, double* ortho_err) {
    std::vector<double> r((size_t)dim * dim);
    uint64_t st = 0x9E3779B97F4A7C15ull ^ ((uint64_t)dim << 32);
    auto next = [&]() {   // splitmix64 -> two uniforms -> a normal (Box-Muller)
        auto u64 = [&]() {
            st += 0x9E3779B97F4A7C15ull;
            uint64_t z = st;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            return z ^ (z >> 31);
        };
        const double u1 = ((double)(u64() >> 11) + 1.0) / 9007199254740993.0, u2 = (double)(u64() >> 11) / 9007199254740992.0;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    };
    for (double& v : r) v = next();
    for (int pass = 0; pass < 2; ++pass)
        for (uint32_t i = 0; i < dim; ++i) {
            double* ri = r.data() + (size_t)i * dim;
            for (uint32_t j = 0; j < i; ++j) {
                const double* rj = r.data() + (size_t)j * dim;
                double d = 0.0;
                for (uint32_t x = 0; x < dim; ++x) d += ri[x] * rj[x];
                for (uint32_t x = 0; x < dim; ++x) ri[x] -= d * rj[x];
            }
            double n = 0.0;
            for (uint32_t x = 0; x < dim; ++x) n += ri[x] * ri[x];
            n = 1.0 / std::sqrt(n);
            for (uint32_t x = 0; x < dim; ++x) ri[x] *= n;
        }
    // rows orthonormal <=> R R^T = I <=> R^T R = I; measured as |R R^T - I|_F (the two Frobenius norms agree for a square matrix
    // up to the conditioning, which is 1 + O(err) here)
    double err2 = 0.0;
    for (uint32_t i = 0; i < dim; ++i)
        for (uint32_t j = 0; j <= i; ++j) {
            double d = 0.0;
            for (uint32_t x = 0; x < dim; ++x) d += r[(size_t)i * dim + x] * r[(size_t)j * dim + x];
            d -= i == j ? 1.0 : 0.0;
            err2 += (i == j ? 1.0 : 2.0) * d * d;
        }
    *ortho_err = std::sqrt(err2);
    rt.resize((size_t)dim * dim);
    for (uint32_t d = 0; d < dim; ++d)
        for (uint32_t j = 0; j < dim; ++j) rt[(size_t)j * dim + d] = r[(size_t)d * dim + j];
}
}  // namespace

SearchError VectorIndex::ensure_filter_copy(hipStream_t stream, bool must) {
    if (filter_ready()) return ok();
    const bool strided = row_stride_ && row_stride_ != dim_ * 2;
    if (f32_ || strided || nrows_ == 0) return ok();
    FSGPU_HIP(hipSetDevice(device_));
    if (!i8f_decided_) {
        // rotate? the slab's largest |element| against the largest row norm spread evenly over the dimensions: a Gaussian-like row
        // has max ~ 6 / sqrt(dim) of its norm (and so has every rotated row), the bench's outlier corpus 17.6 / sqrt(dim)
        bool rot = filter_rotation == 2;
        if (filter_rotation == 0 && dim_ >= 64 && dim_ <= 1024) {
            FSGPU_TRY(i8f_max_.reserve(8));
            unsigned int* w = static_cast<unsigned int*>(i8f_max_.ptr);
            FSGPU_HIP(launch_slab_maxabs(slab_dev_, (size_t)nrows_ * dim_, w, stream));
            FSGPU_HIP(launch_max_row_norm(slab_dev_, (uint32_t)nrows_, dim_, 0, w + 1, stream));
            float host[2] = {0.f, 0.f};
            FSGPU_HIP(hipMemcpyAsync(host, w, 8, hipMemcpyDeviceToHost, stream));
            FSGPU_HIP(hipStreamSynchronize(stream));
            rot = host[1] > 0.f && std::isfinite(host[0]) && std::isfinite(host[1]) &&
                  (double)host[0] * std::sqrt((double)dim_) > kRotateRatio * (double)host[1];
        }
        i8f_rot_ = rot;
        i8f_decided_ = true;
    }
    if (!i8f_rot_) {
        if (!i8_ready_) {
            if (!i8_slab_.reserve((size_t)nrows_ * dim_).ok()) {   // no room for the copy: the f16 paths need none
                (void)hipGetLastError();
                if (must) return make_error(FSGPU_ERR_DEVICE, "no room for the int8 copy of the slab");
                i8f_disabled_ = true;
                return ok();
            }
            FSGPU_TRY(i8_max_.reserve(4));
            FSGPU_HIP(launch_quantize_slab_i8(slab_dev_, (size_t)nrows_ * dim_, static_cast<unsigned int*>(i8_max_.ptr), i8_slab_.ptr, stream,
                                              quant_max_ready_));
            i8_ready_ = true;
        }
        if (!i8_stats_ready_) {
            FSGPU_TRY(i8_stats_.reserve(16));
            FSGPU_HIP(launch_i8_slab_stats(slab_dev_, i8_slab_.ptr, (uint32_t)nrows_, dim_, static_cast<const unsigned int*>(i8_max_.ptr),
                                           static_cast<unsigned int*>(i8_stats_.ptr), stream));
            i8_stats_ready_ = true;
        }
        return ok();
    }
    // the rotated copy: R (f64, transposed) -> two passes over the slab in chunks of rows — max-abs of the rotated values, then
    // quantise + statistics — through a chunk-sized f32 staging buffer
    if (!i8f_slab_.reserve((size_t)nrows_ * dim_).ok()) {
        (void)hipGetLastError();
        if (must) return make_error(FSGPU_ERR_DEVICE, "no room for the int8 copy of the slab");
        i8f_disabled_ = true;
        return ok();
    }
    std::vector<double> rt;
    double ortho_err = 0.0;
    make_rotation(dim_, rt, &ortho_err);
    rot_extra_coeff_ = (ortho_err + 2.01 * 5.9604644775390625e-8) * 1.001;   // |R^T R - I| + 2.01 x 2^-24 (two roundings to f32)
    FSGPU_TRY(rot_mat_.reserve(rt.size() * 8));
    FSGPU_HIP(hipMemcpyAsync(rot_mat_.ptr, rt.data(), rt.size() * 8, hipMemcpyHostToDevice, stream));
    FSGPU_HIP(hipStreamSynchronize(stream));   // rt is a local
    const uint32_t chunk = (uint32_t)std::min<uint64_t>(nrows_, 1u << 18);
    DeviceBuffer tmp;
    FSGPU_TRY(tmp.reserve((size_t)chunk * dim_ * 4));
    FSGPU_TRY(i8f_max_.reserve(8));
    FSGPU_TRY(i8f_stats_.reserve(16));
    unsigned int* maxw = static_cast<unsigned int*>(i8f_max_.ptr);
    unsigned int* stats = static_cast<unsigned int*>(i8f_stats_.ptr);
    const double* rmat = static_cast<const double*>(rot_mat_.ptr);
    float* t32 = static_cast<float*>(tmp.ptr);
    const unsigned char* slab8 = static_cast<const unsigned char*>(slab_dev_);
    SearchError err;
    auto pass = [&](bool second) -> SearchError {
        for (uint64_t r0 = 0; r0 < nrows_; r0 += chunk) {
            const uint32_t n = (uint32_t)std::min<uint64_t>(chunk, nrows_ - r0);
            FSGPU_HIP(launch_rotate_rows_f16(slab8 + (size_t)r0 * dim_ * 2, n, dim_, rmat, t32, stream));
            if (!second) {
                FSGPU_HIP(launch_maxabs_f32(t32, (size_t)n * dim_, maxw, stream));
            } else {
                signed char* dst = static_cast<signed char*>(i8f_slab_.ptr) + (size_t)r0 * dim_;
                FSGPU_HIP(launch_quantize_f32_i8(t32, (size_t)n * dim_, maxw, dst, stream));
                FSGPU_HIP(launch_i8_stats_f32(t32, dst, n, dim_, maxw, stats, stream));
            }
        }
        return ok();
    };
    FSGPU_HIP(hipMemsetAsync(maxw, 0, 8, stream));
    FSGPU_HIP(hipMemsetAsync(stats, 0, 16, stream));
    err = pass(false);
    if (err.ok()) err = pass(true);
    FSGPU_HIP(hipStreamSynchronize(stream));   // the staging buffer goes away with this scope
    tmp.release();
    FSGPU_TRY(err);
    i8f_ready_ = true;
    return ok();
}

// The filter's queries: quantised as quantize_i8_query does + the proven bound delta (prepare_queries_i8_filter_kernel) — of the
// ROTATED queries when the filter's copy is (the same map in f64, rounded once to f32; what that adds to the bound: rot_extra_coeff_).
SearchError VectorIndex::prepare_filter_queries(const float* q, uint32_t nq, uint32_t nq_pad, uint32_t q_stride, void* qi8, float* delta,
                                                float* unit, hipStream_t stream) {
    if (!i8f_rot_) {
        FSGPU_HIP(launch_prepare_queries_i8_filter(q, nq, nq_pad, dim_, q_stride, filter_max(), filter_stats(), qi8, delta, stream, unit));
        return ok();
    }
    FSGPU_TRY(rot_q_.reserve((size_t)std::max<uint32_t>(nq, 1) * dim_ * 4));
    float* rq = static_cast<float*>(rot_q_.ptr);
    FSGPU_HIP(launch_rotate_rows_f32(q, nq, q_stride, dim_, static_cast<const double*>(rot_mat_.ptr), rq, stream));
    FSGPU_HIP(launch_prepare_queries_i8_filter(rq, nq, nq_pad, dim_, dim_, filter_max(), filter_stats(), qi8, delta, stream, unit,
                                               rot_extra_coeff_));
    return ok();
}

// The shard's own max-abs into the quantisers' scale word (device), for a sharded index to reduce across shards
// (ncclAllReduce(max), SURVEY 8f-1: the reference quantises with ONE corpus-wide scale, simd.rs:1865-1886).
SearchError VectorIndex::compute_local_quant_max(unsigned int** max_bits_dev, hipStream_t stream) {
    FSGPU_HIP(hipSetDevice(device_));
    FSGPU_TRY(i8_max_.reserve(4));
    if (nrows_ == 0 || f32_) FSGPU_HIP(hipMemsetAsync(i8_max_.ptr, 0, 4, stream));
    else FSGPU_HIP(launch_slab_maxabs(slab_dev_, (size_t)nrows_ * dim_, static_cast<unsigned int*>(i8_max_.ptr), stream));
    *max_bits_dev = static_cast<unsigned int*>(i8_max_.ptr);
    return ok();
}

// The scale word now holds the CORPUS-wide max-abs: every quantised copy is (re)built from it, on first use.
void VectorIndex::adopt_global_quant_max() {
    quant_max_ready_ = true;
    i8_ready_ = n4_ready_ = n4u_ready_ = i8_stats_ready_ = false;
}

// int8 pass 1 on the matrix cores for a whole batch (exact integer scores), exact f16 rescore, top-k: the batched form of
// search_top_k_int8_two_pass (search.rs:514-661).  multiplier 0 counts as 1, as in the reference.
// bits = 4: the batched form of search_top_k_4bit_two_pass (search.rs:876-946) — the same pipeline over the 4-bit levels, kept one
// per byte so that the int8 matrix-core kernels serve them (that pass is bound by matrix instructions, not by bytes).
SearchError VectorIndex::search_top_k_int8_batched_device(const float* queries_dev, uint32_t nq, uint32_t query_len,
                                                          uint32_t k, uint32_t multiplier, uint32_t* out_rows_dev,
                                                          float* out_scores_dev, uint32_t* out_counts_dev,
                                                          hipStream_t stream, uint32_t* fallbacks, int bits) {
    return batched_impl(queries_dev, nq, query_len, k, nullptr, out_rows_dev, out_scores_dev, out_counts_dev, stream,
                        fallbacks, nullptr, multiplier ? multiplier : 1, 0, false, nullptr, bits == 4 ? 4 : 8);
}

SearchError VectorIndex::two_pass_candidates_device(const float* queries_dev, uint32_t nq, uint32_t query_len, uint32_t k,
                                                    uint32_t multiplier, int bits, uint64_t* approx_out_dev, uint64_t* exact_out_dev,
                                                    hipStream_t stream, uint32_t* fallbacks) {
    int32_t ticket = -1;
    FSGPU_TRY(two_pass_candidates_device_begin(queries_dev, nq, query_len, k, multiplier, bits, approx_out_dev, exact_out_dev, stream, &ticket));
    FSGPU_TRY(two_pass_candidates_device_end(ticket, fallbacks));
    if (nq) {
        FSGPU_HIP(hipSetDevice(device_));
        FSGPU_HIP(hipStreamSynchronize(stream));
    }
    return ok();
}

// ... in two halves, like search_top_k_batched_device_begin / _end (the same two tickets): begin enqueues pass 1, the candidate
// selection and the exact re-score; end waits for that search's event and answers what the batch could not (list overflow: a pile of
// tied integer scores at the threshold) per query.  ticket -1: nothing was enqueued that end would have to wait for.
SearchError VectorIndex::two_pass_candidates_device_begin(const float* queries_dev, uint32_t nq, uint32_t query_len, uint32_t k,
                                                          uint32_t multiplier, int bits, uint64_t* approx_out_dev, uint64_t* exact_out_dev,
                                                          hipStream_t stream, int32_t* ticket) {
    *ticket = -1;
    FSGPU_TRY(ensure_query_dimension(query_len));
    if (nq == 0) return ok();
    const uint64_t mult = multiplier ? multiplier : 1;
    const uint64_t cc = std::max<uint64_t>((uint64_t)k * mult, k);
    if (cc > 256 || k == 0) return make_error(FSGPU_ERR_INVALID_CONFIG, "sharded two-pass: 1 <= k, k * multiplier <= 256");
    FSGPU_HIP(hipSetDevice(device_));
    FSGPU_HIP(hipMemsetAsync(approx_out_dev, 0xff, (size_t)nq * cc * 8, stream));
    FSGPU_HIP(hipMemsetAsync(exact_out_dev, 0xff, (size_t)nq * cc * 8, stream));
    if (f32_) return make_error(FSGPU_ERR_INVALID_CONFIG, "two-pass searches need an F16 slab");
    if (nrows_ == 0) return ok();
    int t = -1;
    for (int i = 0; i < 2; ++i)
        if (async_state_[i] == 0) {
            t = i;
            break;
        }
    if (t < 0) return make_error(FSGPU_ERR_INVALID_CONFIG, "two begun batched searches are outstanding: end one first");
    // the shard-local top-k the pass also produces (not used by the root): one area per ticket
    DeviceBuffer& io = t == 0 ? mf_io_ : mf_io2_;
    FSGPU_TRY(io.reserve((size_t)nq * (k * 8 + 4)));
    uint32_t* rows = static_cast<uint32_t*>(io.ptr);
    float* scores = reinterpret_cast<float*>(rows + (size_t)nq * k);
    uint32_t* counts = reinterpret_cast<uint32_t*>(scores + (size_t)nq * k);
    async_state_[t] = 2;
    async_i8f_[t] = false;
    async_nq_[t] = nq;
    async_fb_[t] = 0;
    async_want_ = t;
    tp_approx_out_ = reinterpret_cast<u64*>(approx_out_dev);
    tp_exact_out_ = reinterpret_cast<u64*>(exact_out_dev);
    tp_stride_ = (uint32_t)cc;
    const SearchError e = batched_impl(queries_dev, nq, query_len, k, nullptr, rows, scores, counts, stream, &async_fb_[t], nullptr,
                                       (uint32_t)mult, 0, false, nullptr, bits == 4 ? 4 : 8);
    tp_approx_out_ = tp_exact_out_ = nullptr;
    tp_stride_ = 0;
    async_want_ = -1;
    if (!e.ok()) {
        async_state_[t] = 0;
        return e;
    }
    *ticket = t;
    return ok();
}

SearchError VectorIndex::two_pass_candidates_device_end(int32_t ticket, uint32_t* fallbacks) {
    if (fallbacks) *fallbacks = 0;
    if (ticket < 0) return ok();
    return search_top_k_batched_device_end(ticket, fallbacks);
}

// ---- the batched (matrix-core) search: prepare -> per round { sample -> main -> finish } -> fallback -------------------------
//
// int8_mult == 0: f16 slab, f16-rounded queries, approximate scores + proven margin (mfma_scan.hip header).
// int8_mult >= 1: int8 slab, int8 queries, exact integer scores; the k * int8_mult best rows are the candidates.
// i8_filter (int8_mult == 0): int8 slab and queries as the FILTER of the exact search — integer scores + the proven margin of
//                 prepare_queries_i8_filter_kernel; queries it cannot certify are re-filtered on the f16 path (*refiltered).

// What one call fixes for all its rounds: the arguments, the sample sizes, the workspaces.
struct VectorIndex::BatchedPlan {
    static constexpr uint32_t GMAX = 160;    // queries per pass: 128 (160 opt-in), or 64 for small batches / tails
    static constexpr uint32_t CAPQ = 8192;   // entries one selection pass covers: block lists + pool fit it at the wide shape
    static constexpr uint32_t SPILL = 4096;  // per-query overflow area for candidates that did not fit their block's list
    static constexpr uint32_t KC = kSelectPool;  // approximate candidates re-scored exactly (at most)
    static constexpr uint32_t RA_MAX = 8192;
    // arguments
    const float* queries_dev = nullptr;
    uint32_t nq = 0, query_len = 0, k = 0;
    const uint64_t* allow_dev = nullptr;
    uint32_t* out_rows_dev = nullptr;
    float* out_scores_dev = nullptr;
    uint32_t* out_counts_dev = nullptr;
    hipStream_t stream = nullptr;
    uint32_t* fallbacks = nullptr;
    uint64_t* out_packed_dev = nullptr;
    uint32_t int8_mult = 0, query_stride = 0;
    uint32_t* refiltered = nullptr;
    int bits = 8;
    // derived
    bool i8f = false, i8 = false, strided = false, skip_b = false, wide_ok = false;
    uint32_t qs = 0;                  // floats between queries
    uint32_t RA = 4096;               // stage A sample rows (dense; <= 8192)
    uint32_t RB = 131072;             // stage B sample rows (upper bound; shrinks with the slab)
    uint32_t ksel_est = 0, ksel = 0;  // the rank the selections anchor on (estimate incl. the int8 filter's growth; exact)
    uint32_t N = 0, QCAP = 0, wide_max = 0, k_eff = 0;
    int wide_pref = 3;
    // per-query verdicts, written by the kernels straight into pinned host memory and read after ONE stream synchronisation
    uint32_t *overflow_all = nullptr, *counts_all = nullptr;
    float *delta = nullptr, *tau = nullptr, *unit = nullptr, *tau_floor = nullptr;
    uint32_t* pool_flag = nullptr;
    u64 *spill = nullptr, *pool = nullptr;
    uint32_t* spill_count = nullptr;
    bool big_pool_last = false;       // the last round's finish had the second-chance launch (debug print only)
    // two_pass_candidates_device: where this batch leaves its candidate pairs (a parked plan's fallback needs them in _end too)
    u64 *tp_approx = nullptr, *tp_exact = nullptr;
    uint32_t tp_stride = 0;
};

// One round: up to QCAP queries — the sample stages and every selection are single launches over all its query groups, only the
// main pass is one launch per group.
struct VectorIndex::BatchedRound {
    uint32_t g0 = 0;                  // first query of the round
    int wide_qt = 0, shape = 0, wpb = 0, full_grid = 0, wide_grid = 0;
    uint32_t G = 0, wide_mult = 1, ngroups = 0, QP = 0, ng = 0, tile_rows = 0;
    const float* qg = nullptr;
    uint32_t *overflow = nullptr, *cand_counts = nullptr, *cand_count = nullptr;
    u64* cand = nullptr;
    MfmaScanArgs a{};
    SelectArgs sb{};
    bool anchor = false, short_stages = false;
    int grid_for(uint32_t rows, uint32_t tile) const {
        int g = (int)(((rows + tile - 1) / tile + wpb - 1) / wpb);
        if (g > full_grid) g = full_grid;
        return g < 1 ? 1 : g;
    }
    // one candidate list of `slots` entries per (query, block); 16..32 slots, sized so that lists + pool fit one selection pass
    // when the grid allows (the wide shape's 256 blocks do)
    uint32_t slots_for(int grid) const {
        return std::min<uint32_t>((uint32_t)scan_mfma_max_slots(shape),
                                  std::max<uint32_t>(16, (BatchedPlan::CAPQ - BatchedPlan::KC) / (uint32_t)grid));
    }
};

SearchError VectorIndex::batched_impl(const float* queries_dev, uint32_t nq, uint32_t query_len, uint32_t k,
                                      const uint64_t* allow_dev, uint32_t* out_rows_dev, float* out_scores_dev,
                                      uint32_t* out_counts_dev, hipStream_t stream, uint32_t* fallbacks,
                                      uint64_t* out_packed_dev, uint32_t int8_mult, uint32_t query_stride, bool i8_filter,
                                      uint32_t* refiltered, int bits) {
    BatchedPlan p;
    p.queries_dev = queries_dev;
    p.nq = nq;
    p.query_len = query_len;
    p.k = k;
    p.allow_dev = allow_dev;
    p.out_rows_dev = out_rows_dev;
    p.out_scores_dev = out_scores_dev;
    p.out_counts_dev = out_counts_dev;
    p.stream = stream;
    p.fallbacks = fallbacks;
    p.out_packed_dev = out_packed_dev;
    p.int8_mult = int8_mult;
    p.query_stride = query_stride;   // floats between queries (0 = dim): an MRL prefix view searches the first dim_ dimensions of full-length queries
    p.refiltered = refiltered;
    p.bits = bits;
    p.i8f = i8_filter && int8_mult == 0;
    p.i8 = int8_mult != 0 || p.i8f;
    p.tp_approx = tp_approx_out_;
    p.tp_exact = tp_exact_out_;
    p.tp_stride = tp_stride_;
    if (refiltered) *refiltered = 0;
    if (fallbacks) *fallbacks = 0;
    FSGPU_TRY(ensure_query_dimension(query_len));
    if (nq == 0) return ok();
    // Every batched search of this index — begun or blocking — works in ONE set of device workspaces (thresholds, candidate lists,
    // spill areas, prepared queries): a search on another stream than an outstanding ticket's is ordered behind that ticket's last
    // kernel (searches on the same stream queue behind it by themselves).
    for (int t = 0; t < 2; ++t)
        if (async_state_[t] == 1 && async_stream_[t] != stream && async_ev_[t]) {
            FSGPU_HIP(hipSetDevice(device_));
            FSGPU_HIP(hipStreamWaitEvent(stream, async_ev_[t], 0));
        }
    bool done = false;
    FSGPU_TRY(batched_prepare(p, &done));
    if (done) return ok();
    for (uint32_t g0 = 0; g0 < nq;) {
        BatchedRound r;
        FSGPU_TRY(batched_round_setup(p, r, g0));
        FSGPU_TRY(batched_sample(p, r));
        FSGPU_TRY(batched_main(p, r));
        FSGPU_TRY(batched_finish(p, r));
        g0 += r.ng;
    }
    // everything of this search is enqueued: the caller's window for host work that should run under it (one shot, outer call only)
    if (after_enqueue_fn && !hard_batch_) {
        void (*fn)(void*) = after_enqueue_fn;
        after_enqueue_fn = nullptr;
        fn(after_enqueue_ctx);
    }
    if (async_want_ >= 0 && !hard_batch_) {
        // fsgpu_search_topk_batched_device_begin: everything is enqueued — the verdicts are read (and the rare uncertified query
        // answered) by _end, behind an event instead of a stream synchronisation, so that the caller can enqueue its next search first
        const int t = async_want_;
        static_assert(std::is_trivially_copyable<BatchedPlan>::value, "the parked plan is copied as bytes");
        async_plan_[t].resize(sizeof(BatchedPlan));
        std::memcpy(async_plan_[t].data(), &p, sizeof(BatchedPlan));
        // (a DEVICE-scope release: the default event makes the GPU write back and invalidate its caches where it is recorded — ~30 us
        // between this search's last kernel and the next search's first, the very gap the two halves exist to close.  What the host
        // reads behind the event are the verdicts, which the kernels write to coherent pinned memory; the outputs in device memory
        // are read by work that is ordered behind them on the GPU, or through copies that bring their own release.)
        if (!async_ev_[t]) FSGPU_HIP(hipEventCreateWithFlags(&async_ev_[t], hipEventDisableTiming | hipEventReleaseToDevice));
        FSGPU_HIP(hipEventRecord(async_ev_[t], p.stream));
        async_stream_[t] = p.stream;
        async_state_[t] = 1;
        return ok();
    }
    return batched_fallback(p);
}

// Stage "prepare": the sample sizes, the shapes the matrix-core path does not cover (answered here, *done = true), the lazily
// built quantised copies and statistics, the workspaces.
SearchError VectorIndex::batched_prepare(BatchedPlan& p, bool* done) {
    *done = false;
    const uint32_t nq = p.nq, k = p.k;
    p.qs = p.query_stride ? p.query_stride : dim_;
    uint32_t RA = 4096, RB = 131072;
    if (knobs().ra > 0) RA = (uint32_t)knobs().ra;  // tuning experiments only
    if (knobs().rb > 0) RB = (uint32_t)knobs().rb;
    constexpr uint32_t RA_MAX = BatchedPlan::RA_MAX;
    if (RA < 256 || RA > RA_MAX || (RA & 63)) RA = 4096;
    // Small slabs (a row shard of a multi-GPU index): a dense sample of 8192 rows already gives a threshold that lets
    // only ~k N / 8192 rows of the main pass through, so the second sampling stage (a launch plus a selection, ~55 us)
    // is skipped when that many candidates fit the block lists comfortably.
    bool skip_b = false;
    // (Not when the main pass is the register-resident-query kernel, i.e. for batches of 256 and more: a row that passes its
    // threshold costs that kernel's 160-instruction tile loop a divergent append, and the looser threshold of a skipped stage
    // B lets 4 x as many through — 1.25M-row shard, 1,024 queries: main pass 0.366 -> 0.329 ms, 2.5M: 0.741 -> 0.642 ms.)
    const bool wide_main = knobs().wide != 0 && nq >= 256 && scan_wide_supported((int)dim_, p.i8 ? 1 : 2) && variant != 5 && variant != 6;
    if (knobs().ra <= 0 && !knobs().no_skip_b && !wide_main && nrows_ <= 4'000'000 && nrows_ >= 4 * (uint64_t)RA_MAX) {
        const uint64_t expect = (uint64_t)std::max<uint32_t>(k, 1) * (p.i8 ? std::max<uint32_t>(p.int8_mult, 1) : 1) * (nrows_ / RA_MAX);
        if (expect <= 4096) {
            RA = RA_MAX;
            skip_b = true;
        }
    }
    // The main pass lets ~ksel N / RB rows through and stage B ~ksel RB / RA: both must stay in the low thousands (block
    // lists, spill area, the selection's capacity), so the samples grow with the rank the selections anchor on.
    // (the int8 filter's margin lets a few times as many rows through each stage as its rank alone would: sized like a larger rank)
    const uint32_t i8f_growth = knobs().i8f_growth > 0 ? (uint32_t)knobs().i8f_growth : 4;
    const uint32_t ksel_est = std::max<uint32_t>(k, 1) * (p.int8_mult ? p.int8_mult : 1) * (p.i8f ? i8f_growth : 1);
    const uint32_t grow = knobs().rb > 0 ? 1 : std::min<uint32_t>(4, (ksel_est + 15) / 16);
    if (knobs().ra <= 0 && ksel_est > 32) RA = RA_MAX;
    // B = about 1/64 of the slab (times the growth), between 8 RA and the cap, a multiple of RA, at most a quarter of it
    RB = std::min<uint32_t>(RB * grow, std::max<uint32_t>(8 * RA, (uint32_t)(nrows_ / 64) * grow));
    if (knobs().rb <= 0) {
        // ... and large enough that the main pass lets ~1,000 rows per query through (ksel N / RB): beyond that the per-block
        // lists and the spill area of the hottest queries overflow (50M rows, k = 10: 10 of 1,024 queries fell back to the
        // exact kernels with the 131,072-row cap, none with 488k — a sample pass of 0.4 ms per 1,024 queries next to 37 ms)
        // (bounded so that the sample stage's own survivors, ksel RB / RA, stay within its lists too: a rank of 90 — the int8
        // fast tier's 3 x 30 candidates — already runs a 524k-row sample)
        const uint64_t need = std::min<uint64_t>((uint64_t)ksel_est * nrows_ / 1024, (uint64_t)6000 * RA_MAX / ksel_est);
        if (need > RB) {
            RB = (uint32_t)std::min<uint64_t>(need, nrows_ / 8);
            if (knobs().ra <= 0) RA = RA_MAX;   // keeps the sample stage's own survivors (ksel RB / RA) in the hundreds
        }
    }
    // (an index whose int8 margin overflowed the main pass's lists samples more: its wide rounds gate the second sample by rank,
    // so that stage's own survivors stay in the hundreds — see search_top_k_batched_device)
    if (p.i8f && wide_main && i8f_sample_boost_ > 1 && knobs().rb <= 0) RB = (uint32_t)std::min<uint64_t>((uint64_t)RB * i8f_sample_boost_, nrows_ / 6);
    if (knobs().rb_pct > 0) RB = (uint32_t)std::min<uint64_t>((uint64_t)RB * (uint32_t)knobs().rb_pct / 100, nrows_ / 4);
    RB = std::min<uint32_t>(RB, (uint32_t)(nrows_ / 4));
    RB = std::max<uint32_t>(RA, RB / RA * RA);
    p.RA = RA;
    p.RB = RB;
    p.skip_b = skip_b;
    p.ksel_est = ksel_est;
    // int8 mode: candidate_count of the reference (search.rs:603-607)
    uint64_t cc64 = std::min<uint64_t>((uint64_t)k * (p.int8_mult ? p.int8_mult : 1), nrows_);
    cc64 = std::max<uint64_t>(cc64, std::min<uint64_t>(k, nrows_));
    p.ksel = p.int8_mult ? (uint32_t)std::min<uint64_t>(cc64, 0xffffffffull) : k;  // rank that anchors the selections
    p.strided = row_stride_ && row_stride_ != dim_ * 2;   // an MRL prefix view
    const bool usable = scan_mfma_supported((int)dim_) && k >= 1 && k <= 64 && p.ksel <= kSelectMaxK && nrows_ >= 4 * (uint64_t)RA && variant != 4 &&
                        !f32_ && (!p.strided || (!p.i8 && p.qs >= dim_)) && (p.query_stride == 0 || !p.i8);
    if (!usable) {
        *done = true;
        return batched_unusable(p);
    }
    FSGPU_HIP(hipSetDevice(device_));
    p.N = (uint32_t)nrows_;
    hipStream_t stream = p.stream;
    if (p.i8 && p.bits == 4 && !n4u_ready_) {  // the 4-bit levels of VectorIndex::nibbles_slab(), one per byte: built lazily, once
        FSGPU_TRY(n4u_slab_.reserve((size_t)nrows_ * dim_));
        FSGPU_TRY(i8_max_.reserve(4));
        FSGPU_HIP(launch_quantize_slab_4bit_levels(slab_dev_, (size_t)nrows_ * dim_, static_cast<unsigned int*>(i8_max_.ptr),
                                                   n4u_slab_.ptr, stream, quant_max_ready_));
        n4u_ready_ = true;
    }
    if (p.i8f) {   // the filter's copy (rotated for slabs with outlier channels) + its statistics: built lazily, once
        FSGPU_TRY(ensure_filter_copy(stream, true));
    } else if (p.i8 && p.bits != 4 && !i8_ready_) {  // VectorIndex::int8_slab(): built lazily, once
        FSGPU_TRY(i8_slab_.reserve((size_t)nrows_ * dim_));
        FSGPU_TRY(i8_max_.reserve(4));
        FSGPU_HIP(launch_quantize_slab_i8(slab_dev_, (size_t)nrows_ * dim_, static_cast<unsigned int*>(i8_max_.ptr),
                                          i8_slab_.ptr, stream, quant_max_ready_));
        i8_ready_ = true;
    }
    if (!p.i8 && !mf_norm_ready_) {
        FSGPU_TRY(mf_max_norm_.reserve(4));
        FSGPU_HIP(launch_max_row_norm(slab_dev_, p.N, dim_, p.strided ? row_stride_ : 0, static_cast<unsigned int*>(mf_max_norm_.ptr), stream));
        mf_norm_ready_ = true;
    }
    // A large batch is answered a "round" of up to QCAP queries at a time: the sample stages and every selection of
    // a round are single launches over all its query groups (one block per query: 1024 blocks fill the chip where a
    // group's 128 leave half the CUs idle), only the main pass is one launch per group.
    constexpr uint32_t GMAX = BatchedPlan::GMAX, SPILL = BatchedPlan::SPILL, KC = BatchedPlan::KC;
    const uint32_t round_cap = knobs().round >= (int)GMAX ? (uint32_t)knobs().round : 1024;  // tuning experiments only
    const uint32_t QCAP = std::min<uint32_t>(round_cap, std::max<uint32_t>(GMAX, (nq + 127) / 128 * 128));
    p.QCAP = QCAP;
    FSGPU_TRY(mf_qh_.reserve((size_t)QCAP * dim_ * 2));
    FSGPU_TRY(mf_delta_.reserve(QCAP * 4));
    FSGPU_TRY(mf_tau_.reserve(QCAP * 16));
    FSGPU_TRY(mf_spill_.reserve((size_t)QCAP * SPILL * 8 + (size_t)QCAP * kMfmaSpillCountStride * 4));
    FSGPU_TRY(mf_dense_.reserve((size_t)QCAP * RA_MAX * 8));
    FSGPU_TRY(mf_sel_.reserve((size_t)QCAP * KC * 8));
    if (mf_shape_ < 0) {
        mf_shape_ = 2;                // 128-query kernel shape (mfma_scan.hip)
        if (knobs().mfma_shape) mf_shape_ = knobs().mfma_shape;  // tuning experiments only
        if (mf_shape_ < 1 || mf_shape_ > 3) mf_shape_ = 2;
        MfmaScanArgs probe{};
        probe.dim = dim_;
        probe.stage = 2;  // the main-pass instantiation
        FSGPU_HIP(launch_scan_mfma(probe, 0, 1, stream, &mf_per_cu_narrow_));
        FSGPU_HIP(launch_scan_mfma(probe, mf_shape_, 1, stream, &mf_per_cu_wide_));
        probe.elem_bytes = 1;
        mf_shape_i8_ = 4;             // int8 rows are half as long: 64-row tiles keep 24 KB in flight per wave
        if (knobs().mfma_shape_i8) mf_shape_i8_ = knobs().mfma_shape_i8;  // tuning experiments only
        if (mf_shape_i8_ < 1 || mf_shape_i8_ > 4) mf_shape_i8_ = 4;
        FSGPU_HIP(launch_scan_mfma(probe, 0, 1, stream, &mf_per_cu_narrow_i8_));
        FSGPU_HIP(launch_scan_mfma(probe, mf_shape_i8_, 1, stream, &mf_per_cu_wide_i8_));
        // 160-query shape: measured 1.49 ms per pass at 10M x 384 (0.64 of HBM peak) against 1.26 ms at 128 queries
        // (0.75) — 7 % more queries per second, but the pass is no longer HBM-bound; opt-in (FSGPU_USE_160=1)
        mf_use_160_ = knobs().use_160;
        FSGPU_HIP(launch_scan_mfma(probe, 5, 1, stream, &mf_per_cu_160_i8_));
        probe.elem_bytes = 2;
        FSGPU_HIP(launch_scan_mfma(probe, 5, 1, stream, &mf_per_cu_160_));
    }
    // the register-resident-query main pass (mfma_wide.hip): 256 queries per launch by default
    // (384 per launch when that many queries are left: the matrix pipe is the bound there and fewer passes leave it more of
    // the power budget — measured 148 k against 130 k queries/s at 10M x 384)
    p.wide_pref = knobs().wide >= 0 ? knobs().wide : 3;
    p.wide_max = (uint32_t)std::max(2, std::min(knobs().wide_max > 0 ? knobs().wide_max : 5, scan_wide_max_query_tiles((int)dim_, p.i8 ? 1 : 2)));
    p.wide_ok = (p.wide_pref == 2 || p.wide_pref == 3) && scan_wide_supported((int)dim_, p.i8 ? 1 : 2) && variant != 5 && variant != 6;
    // per-query verdicts, written by the kernels straight into pinned host memory and read after ONE stream
    // synchronisation for the whole batch: [0, cap) = overflow flags, [cap, 2 cap) = candidate counts
    const uint32_t flag_cap = (nq + GMAX - 1) / GMAX * GMAX + GMAX;
    if (flag_cap > mf_flags_cap_) {
        // (three areas: blocking calls, and one per begun search — a begun search's verdicts must survive the next call's reset)
        if (async_state_[0] == 1 || async_state_[1] == 1)
            return make_error(FSGPU_ERR_INVALID_CONFIG, "a begun batched search is outstanding: end it before searching with a larger batch");
        if (mf_flags_host_) (void)hipHostFree(mf_flags_host_);
        mf_flags_host_ = nullptr;
        mf_flags_cap_ = 0;
        FSGPU_HIP(hipHostMalloc(reinterpret_cast<void**>(&mf_flags_host_), (size_t)flag_cap * 8 * 3, hipHostMallocMapped));
        mf_flags_cap_ = flag_cap;
    }
    uint32_t* flags_area = mf_flags_host_ + (size_t)((async_want_ >= 0 && !hard_batch_) ? 1 + async_want_ : 0) * mf_flags_cap_ * 2;
    p.overflow_all = flags_area;
    p.counts_all = flags_area + mf_flags_cap_;
    std::memset(flags_area, 0, (size_t)mf_flags_cap_ * 8);
    p.delta = static_cast<float*>(mf_delta_.ptr);
    p.tau = static_cast<float*>(mf_tau_.ptr);
    p.unit = p.tau + QCAP;   // int8 filter: integer-score units per exact-score unit, per query
    p.pool_flag = reinterpret_cast<uint32_t*>(p.unit + QCAP);   // finish: candidates did not fit the pool (per query of the round)
    p.tau_floor = reinterpret_cast<float*>(p.pool_flag + QCAP);   // the first sample's proven threshold, kept next to a heuristic one
    p.spill = static_cast<u64*>(mf_spill_.ptr);
    p.spill_count = reinterpret_cast<uint32_t*>(p.spill + (size_t)QCAP * SPILL);
    p.pool = static_cast<u64*>(mf_sel_.ptr);
    p.k_eff = std::min<uint32_t>(k, p.N);
    return ok();
}

// Shapes the matrix-core path does not cover: answered by the per-query kernels (or handed to the f16 branch).
SearchError VectorIndex::batched_unusable(BatchedPlan& p) {
    const uint32_t nq = p.nq, k = p.k;
    hipStream_t stream = p.stream;
    if (p.i8f)   // the f16 branch sorts them out
        return batched_impl(p.queries_dev, nq, p.query_len, k, p.allow_dev, p.out_rows_dev, p.out_scores_dev, p.out_counts_dev, stream,
                            p.fallbacks, p.out_packed_dev, 0, p.query_stride, false, nullptr);
    if (p.i8) {
        // per-query int8 two-pass through host staging (rare shapes: huge candidate counts, tiny or odd-dimension slabs)
        std::vector<float> q((size_t)nq * dim_), sc((size_t)nq * k);
        std::vector<uint32_t> rw((size_t)nq * k, 0xffffffffu), cnt(nq);
        FSGPU_HIP(hipMemcpyAsync(q.data(), p.queries_dev, q.size() * 4, hipMemcpyDeviceToHost, stream));
        FSGPU_HIP(hipStreamSynchronize(stream));
        for (uint32_t i = 0; i < nq; ++i)
            FSGPU_TRY(quantized_two_pass(q.data() + (size_t)i * dim_, dim_, k, p.int8_mult, p.bits, rw.data() + (size_t)i * k,
                                         sc.data() + (size_t)i * k, &cnt[i], p.tp_approx ? p.tp_approx + (size_t)i * p.tp_stride : nullptr,
                                         p.tp_exact ? p.tp_exact + (size_t)i * p.tp_stride : nullptr));
        if (p.out_rows_dev) FSGPU_HIP(hipMemcpyAsync(p.out_rows_dev, rw.data(), rw.size() * 4, hipMemcpyHostToDevice, stream));
        if (p.out_scores_dev) FSGPU_HIP(hipMemcpyAsync(p.out_scores_dev, sc.data(), sc.size() * 4, hipMemcpyHostToDevice, stream));
        if (p.out_counts_dev) FSGPU_HIP(hipMemcpyAsync(p.out_counts_dev, cnt.data(), cnt.size() * 4, hipMemcpyHostToDevice, stream));
        FSGPU_HIP(hipStreamSynchronize(stream));
        if (p.fallbacks) *p.fallbacks = nq;
        return ok();
    }
    if (p.query_stride)
        return make_error(FSGPU_ERR_INVALID_CONFIG, "strided queries need the matrix-core path (caller falls back per query)");
    if (p.fallbacks) *p.fallbacks = nq;
    if (p.out_packed_dev) {
        FSGPU_TRY(search_top_k_packed_device(p.queries_dev, nq, p.query_len, k, p.allow_dev, p.out_packed_dev, stream));
        if (!p.out_rows_dev) return ok();
    }
    return search_top_k_device(p.queries_dev, nq, p.query_len, k, p.allow_dev, p.out_rows_dev, p.out_scores_dev, p.out_counts_dev, stream);
}

// The geometry of the round that starts at query g0, the prepared (rounded / quantised) queries, the scan arguments every stage shares.
SearchError VectorIndex::batched_round_setup(const BatchedPlan& p, BatchedRound& r, uint32_t g0) {
    const bool i8 = p.i8;
    hipStream_t stream = p.stream;
    r.g0 = g0;
    const uint32_t left = p.nq - g0;
    // Main pass at 384 / 256 queries per launch (mfma_wide.hip: queries in registers, row tiles through an LDS-DMA
    // ring) when that many are left; the sample stages then run as sub-groups of 128 on the LDS-query kernel.
    r.wide_qt = 0;
    if (p.wide_ok && left >= 256) {   // 128-query groups per launch
        // as many as the registers hold (f16 rows of 384 dimensions: 3, their int8 form: 5), the round's groups spread evenly
        // over its passes (8 groups: 3 + 3 + 2 on f16 rows, 4 + 4 on int8 rows)
        const uint32_t groups_left = std::min<uint32_t>(left / 128, p.QCAP / 128);
        const uint32_t passes = (groups_left + p.wide_max - 1) / p.wide_max;
        r.wide_qt = p.wide_pref == 2 ? 2 : (int)((groups_left + passes - 1) / passes);
    }
    // 160, 128 or 64 queries per pass
    r.shape = r.wide_qt ? (i8 ? mf_shape_i8_ : mf_shape_)
                        : (left > 64 && variant != 5) ? ((left > 128 && mf_use_160_) ? 5 : (i8 ? mf_shape_i8_ : mf_shape_)) : 0;
    r.G = (uint32_t)scan_mfma_query_tiles(r.shape) * 16;
    r.wide_mult = r.wide_qt ? (uint32_t)r.wide_qt : 1;   // sample groups per main-pass launch
    // this round: `ngroups` groups of G queries (the last one may be partly padding), QP query slots, ng real queries
    r.ngroups = left >= r.G ? std::min<uint32_t>(left / r.G, p.QCAP / r.G) : 1;
    if (r.wide_qt) r.ngroups = r.ngroups / r.wide_mult * r.wide_mult;
    r.QP = r.ngroups * r.G;
    r.ng = std::min(r.QP, left);
    r.wpb = scan_mfma_waves_per_block(r.shape);
    const int per_cu = r.shape == 5 ? (i8 ? mf_per_cu_160_i8_ : mf_per_cu_160_)
                                    : (i8 ? (r.shape ? mf_per_cu_wide_i8_ : mf_per_cu_narrow_i8_)
                                          : (r.shape ? mf_per_cu_wide_ : mf_per_cu_narrow_));
    r.full_grid = num_cus_ * per_cu;
    r.tile_rows = (uint32_t)scan_mfma_rows_per_tile(r.shape);
    r.qg = p.queries_dev + (size_t)g0 * p.qs;
    r.overflow = p.overflow_all + g0;
    r.cand_counts = p.counts_all + g0;
    if (p.i8f)
        FSGPU_TRY(prepare_filter_queries(r.qg, r.ng, r.QP, p.qs, mf_qh_.ptr, p.delta, p.unit, stream));
    else if (i8) FSGPU_HIP(launch_prepare_queries_i8(r.qg, r.ng, r.QP, dim_, mf_qh_.ptr, p.delta, stream, p.bits));
    else
        FSGPU_HIP(launch_prepare_queries(r.qg, r.ng, r.QP, dim_, p.qs, static_cast<const unsigned int*>(mf_max_norm_.ptr),
                                         mf_qh_.ptr, p.delta, stream));
    r.wide_grid = num_cus_ * mf_per_cu_wide_main_;
    FSGPU_TRY(mf_cand_.reserve((size_t)r.QP * std::max(r.full_grid, r.wide_grid) * kMfmaMaxSlots * 8));
    r.cand = static_cast<u64*>(mf_cand_.ptr);
    FSGPU_TRY(mf_cand_count_.reserve((size_t)r.QP * r.wide_grid * 4));   // the wide kernels' list lengths (no padding)
    r.cand_count = static_cast<uint32_t*>(mf_cand_count_.ptr);
    MfmaScanArgs& a = r.a;
    a = MfmaScanArgs{};
    a.slab = p.i8f ? filter_slab() : i8 ? (p.bits == 4 ? n4u_slab_.ptr : i8_slab_.ptr) : slab_dev_;
    a.elem_bytes = i8 ? 1 : 2;
    a.live = reinterpret_cast<const u64*>(live_dev_);
    a.allow = reinterpret_cast<const u64*>(p.allow_dev);
    a.queries = mf_qh_.ptr;
    a.tau = p.tau;
    a.cand = r.cand;
    a.spill = p.spill;
    a.spill_count = p.spill_count;
    a.spill_cap = BatchedPlan::SPILL;
    a.overflow = r.overflow;
    a.dim = dim_;
    a.row_stride = p.strided ? row_stride_ : 0;
    a.row_base = (uint32_t)row_base_;
    a.nrows = p.N;
    // (a row shard's stages are short: the lists' padding and the selection's reads are a visible part of them — 8 / 16 slots
    // there, overflow goes to the spill area; 1.25M rows: 0.809 -> 0.789 ms per 1,024 queries, nothing at 10M)
    r.short_stages = nrows_ < 4'000'000;
    // int8 filter: the sample stages' thresholds are anchored on EXACT scores (their candidates are re-scored from the f16
    // slab right in the selection): one delta below the k-th best instead of two — the margin's multiplier on the rows each
    // stage lets through is exponential in it
    r.anchor = p.i8f && !knobs().no_anchor;
    return ok();
}

// Stage "sample": A = dense approximate scores of a small sample -> a first threshold; B = the rows of a larger sample at or above
// it, one short list per (query, block) -> the threshold the main pass runs with.  Samples are 64-row groups spread evenly over
// the slab: B = every stride_b-th group, A = a subset of B.
SearchError VectorIndex::batched_sample(const BatchedPlan& p, BatchedRound& r) {
    constexpr uint32_t SPILL = BatchedPlan::SPILL;
    hipStream_t stream = p.stream;
    const bool i8 = p.i8, i8f = p.i8f, skip_b = p.skip_b;
    const uint32_t N = p.N, RA = p.RA, RB = p.RB, QP = r.QP, ksel = p.ksel;
    MfmaScanArgs& a = r.a;
    const uint32_t groups_a = RA / 64, groups_b = RB / 64;
    const uint32_t stride_b = (N / 64) / groups_b;  // >= 4
    // The int8 filter's wide rounds (ranks up to kGroupsTaken): ONE sample pass that appends nothing — every block reports, per query,
    // its four best GROUPS of 8 rows (best approximate score | where), and the selection re-scores the rows of the best 24 groups from
    // the f16 slab: tau = max(a_k - 2 delta, S_k x unit - delta) exactly as the exact-anchor step below, with no first sample to gate
    // the second, no lists, no divergent append path in the sample's loop (stage A 0.05 ms + stage B 0.20 + its selection 0.07 per
    // 1,024 queries at 10M rows became 0.1 + 0.03).
    // (under a tombstone / allow bitmap the sample pass takes its maxima over live, allowed rows only: with the maxima over ALL rows a
    // group's best row was as likely filtered out as the bitmap is sparse, and the threshold anchored on what was left of 24 groups
    // let 1.7 x the rows through the main pass at 50 % allowed — 199 k against 308 k queries/s for the thresholded stages)
    // (the batched int8 / 4-bit two-pass takes the same pass: its pass-1 scores are the reference's own, so the k x multiplier-th best
    // group maximum IS a valid threshold — no re-score, ranks up to 64)
    const bool rank_groups = i8 && !i8f && ksel <= 64;
    const bool group_sample = (r.anchor ? ksel <= kGroupsTaken : rank_groups) && r.wide_qt != 0 && !skip_b && !knobs().no_wide_b &&
                              !knobs().no_group_sample && (dim_ & 7) == 0 && dim_ <= 1024 && scan_wide_group_maxima_supported((int)dim_, r.wide_qt);
    if (group_sample) {
        const int grid_g = std::min(r.wide_grid, (int)std::max<uint32_t>(1, RB / 64 / 4));   // at least 4 sample groups per block
        // (enough groups for the picks: the rank form needs ksel of them — and not all from a wave or two of the selection)
        if (grid_g * 4 <= 1024 && (uint32_t)grid_g * 4 >= (r.anchor ? 96u : 4u * ksel)) {
            MfmaScanArgs c = a;
            c.dense = nullptr;
            c.stage = 3;
            c.group_stride = stride_b;
            c.group_count = groups_b;
            c.slots = 4;
            c.groups = r.ngroups / r.wide_mult;
            c.queries = mf_qh_.ptr;
            c.tau = p.tau;
            c.cand = r.cand;
            c.cand_count = nullptr;
            c.spill = p.spill;
            c.spill_count = p.spill_count;
            c.overflow = r.overflow;
            FSGPU_HIP(launch_scan_wide(c, r.wide_qt, grid_g, stream, nullptr));
            GroupSelectArgs g{};
            g.groups = r.cand;
            g.nentries = (uint32_t)grid_g * 4;
            g.k = ksel;
            g.delta = p.delta;
            g.anchor_unit = p.unit;
            g.tau_out = p.tau;
            g.overflow = r.overflow;
            g.spill_reset = p.spill_count;   // the main pass appends from zero
            g.slab = slab_dev_;
            g.live = a.live;
            g.allow = a.allow;
            g.queries = r.qg;
            g.dim = dim_;
            g.nrows = N;
            g.row_base = (uint32_t)row_base_;
            g.query_stride = p.qs;
            g.hreduce = hreduce;
            g.valid_queries = r.ng;
            g.rank_only = r.anchor ? 0u : 1u;
            FSGPU_HIP(launch_select_groups(g, (int)QP, stream));
            a.dense = nullptr;
            a.stage = 1;
            a.group_stride = stride_b;
            a.group_count = groups_b;
            a.groups = r.ngroups;
            SelectArgs& sb = r.sb;   // what the main pass and the finish expect from this stage
            sb = SelectArgs{};
            sb.lists = r.cand;
            sb.k = ksel;
            sb.take_topk = (i8 && !i8f) ? 1 : 0;
            sb.delta = p.delta;
            sb.overflow = r.overflow;
            sb.spill = p.spill;
            sb.spill_count = p.spill_count;
            sb.spill_cap = SPILL;
            return ok();
        }
    }
    // stage A: dense approximate scores of the A sample -> tau = (k-th best) - 2 delta
    a.dense = static_cast<u64*>(mf_dense_.ptr);
    a.stage = 0;
    a.group_stride = stride_b * (groups_b / groups_a);
    a.group_count = groups_a;
    a.slots = 0;
    a.groups = r.ngroups;
    FSGPU_HIP(launch_scan_mfma(a, r.shape, r.grid_for(RA, 16), stream, nullptr));
    SelectArgs sa{};
    sa.lists = a.dense;
    sa.q_stride = RA;
    sa.l_stride = RA;
    sa.nlists = 1;
    sa.list_len = RA;
    sa.k = ksel;
    sa.delta = p.delta;
    sa.tau_out = p.tau;
    auto set_rescore = [&](SelectArgs& x) {
        x.slab = slab_dev_;
        x.queries = r.qg;
        x.dim = dim_;
        x.row_stride = 0;
        x.query_stride = p.qs;
        x.nrows = N;
        x.row_base = (uint32_t)row_base_;
        x.hreduce = hreduce;
        x.k_out = p.k_eff;
    };
    // (a wide round's second sample only anchors the main pass's threshold — that pass visits every row — so it may take ANY
    // subset of the sample: the rows above the first sample's ~9th best score WITHOUT a margin, half as many as the proven
    // threshold lets through, and the first selection needs no exact re-score)
    // (the int8 two-pass takes the same shortcut: its threshold is the ksel-th best integer score of whatever subset came through)
    const bool heur_b = (r.anchor || (i8 && !i8f)) && r.wide_qt != 0 && !skip_b && !knobs().no_wide_b && !knobs().no_heur_b && ksel >= 8;
    if (heur_b) {
        // rank r of the first sample: the second sample holds RB / RA x as many rows above that score as the first (r, up to
        // an order statistic's spread ~ Gamma(r)), and k of them are needed — r = 5 + k / 8 puts "fewer than k came through"
        // (which only costs that query a looser threshold) near 1e-6 per query for k <= 64 and RB / RA >= 47; every rank less is
        // ~47 fewer appends per query in the append-bound sample pass (r = 9 -> 6 at k = 10: 0.7 % of a step, scripts/r03/sweep_rb.sh)
        sa.heur_rank = knobs().heur_rank > 0 ? (uint32_t)std::min<int>(knobs().heur_rank, (int)ksel) : std::min<uint32_t>(ksel, 5 + ksel / 8);
        sa.tau_floor_out = p.tau_floor;
    } else if (r.anchor) {
        set_rescore(sa);
        sa.anchor_unit = p.unit;
    }
    sa.valid_queries = r.ng;   // (the launch covers the round's padded query slots)
    sa.spill_reset = p.spill_count;   // the round's spill counters start at zero for the stage that follows (B, or the main pass)
    FSGPU_HIP(launch_select(sa, (int)QP, stream));
    // stage B: the B sample's rows at or above tau, one short list per (query, block) -> tighter tau; the rows
    // still at or above it form the pool carried into the last selection
    a.dense = nullptr;
    a.stage = 1;
    a.group_stride = stride_b;
    a.group_count = groups_b;
    const int grid_b = r.grid_for(RB, r.tile_rows);
    a.slots = r.slots_for(grid_b);
    // (a wide round samples on the register-resident-query kernel too: one launch per main-pass group, the same lists)
    const bool wide_b = r.wide_qt && !skip_b && !knobs().no_wide_b;
    const int wide_grid_b = std::min(r.wide_grid, (int)std::max<uint32_t>(1, RB / 64 / 4));   // at least 4 sample groups per block
    if (wide_b) a.slots = knobs().slots_b > 0 ? (uint32_t)knobs().slots_b : r.short_stages ? 8 : kWideSlots;
    if (!skip_b) {
        if (wide_b) {
            // ONE launch for all the round's main-pass groups (gridDim.y: group g's arrays follow group g - 1's, mfma_wide.hip)
            MfmaScanArgs c = a;
            c.groups = r.ngroups / r.wide_mult;
            c.queries = mf_qh_.ptr;
            c.tau = p.tau;
            c.cand = r.cand;
            c.cand_count = r.cand_count;
            c.spill = p.spill;
            c.spill_count = p.spill_count;
            c.overflow = r.overflow;
            FSGPU_HIP(launch_scan_wide(c, r.wide_qt, wide_grid_b, stream, nullptr));
        } else {
            FSGPU_HIP(launch_scan_mfma(a, r.shape, grid_b, stream, nullptr));
        }
    }
    const int lists_b = wide_b ? wide_grid_b : grid_b;
    SelectArgs& sb = r.sb;
    sb = SelectArgs{};
    sb.lists = r.cand;
    sb.q_stride = (uint64_t)lists_b * a.slots;
    sb.l_stride = a.slots;
    sb.nlists = (uint32_t)lists_b;
    sb.list_len = a.slots;
    sb.list_counts = wide_b ? r.cand_count : nullptr;
    sb.k = ksel;
    sb.take_topk = (i8 && !i8f) ? 1 : 0;
    sb.delta = p.delta;
    sb.overflow = r.overflow;
    sb.spill = p.spill;
    sb.spill_count = p.spill_count;
    sb.spill_cap = SPILL;
    if (!skip_b) {
        sb.tau_out = p.tau;
        sb.pool_out = p.pool;
        if (r.anchor) {
            set_rescore(sb);
            sb.anchor_unit = p.unit;
        }
        if (heur_b) sb.tau_floor_in = p.tau_floor;
        sb.valid_queries = r.ng;   // (the launch covers the round's padded query slots)
        sb.spill_reset = p.spill_count;   // ... and at zero again for the main pass
        FSGPU_HIP(launch_select(sb, (int)QP, stream));
        sb.spill_reset = nullptr;
        sb.valid_queries = 0;
        sb.anchor_unit = nullptr;
        sb.tau_floor_in = nullptr;
    }
    return ok();
}

// Stage "main": every row (the wide pass) or every group the B sample did not cover against the round's thresholds; one launch per
// query group, candidates into per-(query, block) lists + the spill area.
SearchError VectorIndex::batched_main(const BatchedPlan& p, BatchedRound& r) {
    constexpr uint32_t SPILL = BatchedPlan::SPILL, CAPQ = BatchedPlan::CAPQ, KC = BatchedPlan::KC;
    hipStream_t stream = p.stream;
    MfmaScanArgs& a = r.a;
    SelectArgs& sb = r.sb;
    // (an event pair idles the stream ~6 us on either side of a launch: with a period, the main launches of every n-th call are timed)
    const bool profiling = this->profiling && (profile_period <= 1 || profile_tick_++ % (uint32_t)profile_period == 0);
    if (p.skip_b) {
        a.group_stride = 1;  // nothing was sampled by a stage B: the main pass visits every group
        a.group_count = 0;
    }
    // stage C: every group the B sample did not cover
    a.stage = 2;
    const int main_grid = r.wide_qt ? r.wide_grid : r.full_grid;
    if (r.wide_qt) {  // the wide main pass visits every row (no skip test in its loop): stage B only tightened tau
        a.group_stride = 1;
        a.group_count = 0;
    }
    // (the wide pass appends to global lists: 16 slots per (query, block) keep lists + pool inside one selection pass;
    // ranks above 32 — the int8 fast tier anchors on 90 — let ~1,700 rows per query through and get 32)
    a.slots = r.wide_qt ? (knobs().slots_main > 0 ? (uint32_t)knobs().slots_main
                           : p.ksel_est > 32 ? (r.short_stages && p.i8f ? 16 : kWideSlots)
                                             : std::min<uint32_t>(16, std::max<uint32_t>(8, (CAPQ - KC) / (uint32_t)main_grid)))
                        : r.slots_for(r.full_grid);
    // (the spill counters were zeroed by the selection in front of this stage: SelectArgs::spill_reset)
    a.groups = 1;
    // one pass over the slab per query group, one launch each (all groups in one launch — a group's blocks taking
    // over the CUs the previous group's leave — measured 1.5 % slower at 10M rows: two groups' streams interleave)
    const uint32_t GM = r.G * r.wide_mult;  // queries per main-pass launch
    // The register-resident-query kernel takes ALL the round's groups in one launch (gridDim.y = passes over the slab): a group's
    // blocks start on a CU as the previous group's block leaves it, so a step pays one launch ramp and one chip-wide tail instead of
    // one per 512 queries — what a 1.25M-row shard, whose pass is 0.2 ms, feels most.
#ifdef FSGPU_LAB_SPLIT_LAUNCHES   // lab: one launch per 512-query group, as before round 4 (same-box A/B of the merged launch)
    const uint32_t wide_groups = 0;
#else
    const uint32_t wide_groups = r.wide_qt ? r.ngroups / r.wide_mult : 0;
#endif
    if (wide_groups) {
        MfmaScanArgs c = a;
        c.groups = wide_groups;
        c.queries = mf_qh_.ptr;
        c.tau = p.tau;
        c.cand = r.cand;
        c.cand_count = r.cand_count;
        c.spill = p.spill;
        c.spill_count = p.spill_count;
        c.overflow = r.overflow;
        c.reverse = knobs().no_reverse ? 0 : (mf_pass_parity_ & 1);   // group g walks in direction (parity + g) & 1
        mf_pass_parity_ += wide_groups;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (profiling) {
            FSGPU_HIP(hipEventCreateWithFlags(&e0, hipEventReleaseToDevice));   // (timing only: no cache write-back around the launch)
            FSGPU_HIP(hipEventCreateWithFlags(&e1, hipEventReleaseToDevice));
            FSGPU_HIP(hipEventRecord(e0, stream));
        }
        FSGPU_HIP(launch_scan_wide(c, r.wide_qt, main_grid, stream, nullptr));
        if (profiling) {
            FSGPU_HIP(hipEventRecord(e1, stream));
            events_.emplace_back(e0, e1);
            profiled_rows_ += (uint64_t)p.N * wide_groups;   // every group streams the whole slab
            profiled_elem_bytes_ = p.i8 ? 1 : 2;
        }
    }
    for (uint32_t j = 0; !wide_groups && j < r.ngroups / r.wide_mult; ++j) {
        MfmaScanArgs c = a;
        c.queries = static_cast<const unsigned char*>(mf_qh_.ptr) + (size_t)j * GM * dim_ * (p.i8 ? 1 : 2);
        c.tau = p.tau + (size_t)j * GM;
        c.cand = r.cand + (size_t)j * GM * main_grid * a.slots;
        c.cand_count = r.wide_qt ? r.cand_count + (size_t)j * GM * main_grid : nullptr;
        c.spill = p.spill + (size_t)j * GM * SPILL;
        c.spill_count = p.spill_count + (size_t)j * GM * kMfmaSpillCountStride;
        c.overflow = r.overflow + (size_t)j * GM;
        c.reverse = knobs().no_reverse ? 0 : (mf_pass_parity_++ & 1);  // consecutive passes alternate direction
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (profiling) {
            FSGPU_HIP(hipEventCreate(&e0));
            FSGPU_HIP(hipEventCreate(&e1));
            FSGPU_HIP(hipEventRecord(e0, stream));
        }
        if (r.wide_qt) {
            c.groups = 1;
            FSGPU_HIP(launch_scan_wide(c, r.wide_qt, main_grid, stream, nullptr));
        } else {
            FSGPU_HIP(launch_scan_mfma(c, r.shape, r.full_grid, stream, nullptr));
        }
        if (profiling) {
            FSGPU_HIP(hipEventRecord(e1, stream));
            events_.emplace_back(e0, e1);
            profiled_rows_ += (p.skip_b || r.wide_qt) ? p.N : p.N - p.RB;
            profiled_elem_bytes_ = p.i8 ? 1 : 2;
        }
    }
    sb.q_stride = (uint64_t)main_grid * a.slots;
    sb.l_stride = a.slots;
    sb.nlists = (uint32_t)main_grid;
    sb.list_len = a.slots;
    sb.list_counts = r.wide_qt ? r.cand_count : nullptr;
    sb.extra = (p.skip_b || r.wide_qt) ? nullptr : p.pool;
    sb.extra_len = (p.skip_b || r.wide_qt) ? 0 : KC;
    sb.tau_out = nullptr;
    sb.pool_out = nullptr;
    return ok();
}

// Stage "finish": every row whose approximate score is within 2 delta of the k-th best (more than KC of them: the query goes to
// the exact path) is re-scored in the reference's order; the best k exact entries are the answer.
SearchError VectorIndex::batched_finish(BatchedPlan& p, BatchedRound& r) {
    hipStream_t stream = p.stream;
    SelectArgs& sb = r.sb;
    const uint32_t k = p.k, g0 = r.g0;
    sb.cand_counts = r.cand_counts;
    // the int8 filter's margin (and whatever it hands on to the f16 filter) can put thousands of rows within reach of the k-th
    // score: the finish re-scores up to 8,192 of them per query instead of 1,024
    const bool second_chance = (p.i8f || hard_batch_) && !knobs().no_big_pool;
    p.big_pool_last = second_chance;
    sb.pool_flag = second_chance ? p.pool_flag : nullptr;
    sb.slab = slab_dev_;
    sb.queries = r.qg;
    sb.dim = dim_;
    sb.row_stride = p.strided ? row_stride_ : 0;
    sb.query_stride = p.qs;
    sb.nrows = p.N;
    sb.row_base = (uint32_t)row_base_;
    sb.hreduce = hreduce;
    sb.k_out = p.k_eff;
    sb.out_stride = k;
    sb.out_rows = p.out_rows_dev ? p.out_rows_dev + (size_t)g0 * k : nullptr;
    sb.out_scores = p.out_scores_dev ? p.out_scores_dev + (size_t)g0 * k : nullptr;
    sb.out_counts = p.out_counts_dev ? p.out_counts_dev + g0 : nullptr;
    sb.out_packed = p.out_packed_dev ? reinterpret_cast<u64*>(p.out_packed_dev) + (size_t)g0 * k : nullptr;
    if (p.int8_mult && p.tp_approx) {   // a sharded index's shard: the candidate pairs themselves (two_pass_candidates_device)
        sb.cand_approx_out = p.tp_approx + (size_t)g0 * p.tp_stride;
        sb.cand_exact_out = p.tp_exact + (size_t)g0 * p.tp_stride;
        sb.cand_out_stride = p.tp_stride;
    }
#ifdef FSGPU_EXPERIMENTS
    static unsigned long long* sel_stamps = nullptr;   // FSGPU_SELECT_STAMPS=1: shader clocks of the phases of blocks 0, 256, 512, 768 of the finish
    if (fsgpu::lab_env("FSGPU_SELECT_STAMPS") && !sel_stamps) (void)hipHostMalloc(reinterpret_cast<void**>(&sel_stamps), 64 * 8, hipHostMallocMapped);
    if (sel_stamps) {
        if (sel_stamps[8]) {
            for (int b = 0; b < 4; ++b) {
                std::fprintf(stderr, "[select stamps] block %4d start %+8lld:", b * 256, (long long)(sel_stamps[b * 16] - sel_stamps[0]));
                for (int i = 1; i <= 8; ++i) std::fprintf(stderr, " %lld", (long long)(sel_stamps[b * 16 + i] - sel_stamps[b * 16]));
                std::fprintf(stderr, " nc=%lld\n", (long long)sel_stamps[b * 16 + 15]);
            }
        }
        std::memset(sel_stamps, 0, 64 * 8);
        sb.stamps = sel_stamps;
    }
#endif
    FSGPU_HIP(launch_select(sb, (int)r.ng, stream));
    sb.stamps = nullptr;
    if (second_chance) {   // queries whose candidates did not fit the pool: the sorted finish over the same lists (others return at once)
        sb.big_pool = 1;
        FSGPU_HIP(launch_select(sb, (int)r.ng, stream));
    }
    return ok();
}

// Stage "fallback": ONE stream synchronisation for the whole batch, then the host reads the per-query verdicts — margin / capacity
// overflow, or fewer than k candidates — and the uncertified queries are answered by the exact kernels (the int8 filter hands a
// larger set to the f16 filter first; the int8 two-pass to its per-query form).
SearchError VectorIndex::batched_fallback(BatchedPlan& p, bool already_waited) {
    constexpr uint32_t KC = BatchedPlan::KC;
    hipStream_t stream = p.stream;
    const uint32_t nq = p.nq, k = p.k, k_eff = p.k_eff;
    // (polling the stream with hipStreamQuery before blocking was measured: no change at 10M rows or on a 1.25M-row shard,
    // profiles/r04/step_overheads.txt — the runtime's wait is already an active one for waits this short)
    if (!already_waited) FSGPU_HIP(hipStreamSynchronize(stream));
    std::vector<uint32_t> fb;
    for (uint32_t i = 0; i < nq; ++i)
        if (p.overflow_all[i] || p.counts_all[i] < k_eff) fb.push_back(i);
    if (knobs().debug_batched) {
        uint32_t big = 0, slot = 0, few = 0, mx = 0;
        for (uint32_t i = 0; i < nq; ++i) {
            if (p.counts_all[i] > (p.big_pool_last ? 8192u : KC)) ++big;
            else if (p.overflow_all[i]) ++slot;
            if (p.counts_all[i] < k_eff) ++few;
            mx = std::max(mx, p.counts_all[i]);
        }
        std::fprintf(stderr, "[fsgpu batched] nq=%u k=%u fallbacks=%zu  pool_overflow=%u  slot_or_skip=%u  few=%u  max_cand=%u\n", nq, k,
                     fb.size(), big, slot, few, mx);
        for (size_t j = 0; j < fb.size() && j < 4; ++j)
            std::fprintf(stderr, "    query %u: overflow=%u candidates=%u\n", fb[j], p.overflow_all[fb[j]], p.counts_all[fb[j]]);
    }
    const uint32_t total_fallbacks = (uint32_t)fb.size();
    if (total_fallbacks && p.i8 && !p.i8f) {
        // list/spill overflow (a pile of tied scores at the threshold): the per-query int8 two-pass answers those
        std::vector<float> qh(dim_), sc(k);
        std::vector<uint32_t> rw(k);
        for (uint32_t i : fb) {
            uint32_t cnt = 0;
            FSGPU_HIP(hipMemcpyAsync(qh.data(), p.queries_dev + (size_t)i * dim_, (size_t)dim_ * 4, hipMemcpyDeviceToHost, stream));
            FSGPU_HIP(hipStreamSynchronize(stream));
            std::fill(rw.begin(), rw.end(), 0xffffffffu);
            FSGPU_TRY(quantized_two_pass(qh.data(), dim_, k, p.int8_mult, p.bits, rw.data(), sc.data(), &cnt,
                                         p.tp_approx ? p.tp_approx + (size_t)i * p.tp_stride : nullptr,
                                         p.tp_exact ? p.tp_exact + (size_t)i * p.tp_stride : nullptr));
            if (p.out_rows_dev) FSGPU_HIP(hipMemcpyAsync(p.out_rows_dev + (size_t)i * k, rw.data(), (size_t)k * 4, hipMemcpyHostToDevice, stream));
            if (p.out_scores_dev) FSGPU_HIP(hipMemcpyAsync(p.out_scores_dev + (size_t)i * k, sc.data(), (size_t)k * 4, hipMemcpyHostToDevice, stream));
            if (p.out_counts_dev) FSGPU_HIP(hipMemcpyAsync(p.out_counts_dev + i, &cnt, 4, hipMemcpyHostToDevice, stream));
            FSGPU_HIP(hipStreamSynchronize(stream));
        }
    } else if (total_fallbacks) {
        // compact the uncertified queries, answer them with the exact kernels (8 per pass), scatter the hits back
        const size_t nf = fb.size();
        auto align_up = [](size_t v, size_t a) { return (v + a - 1) / a * a; };
        const size_t o_idx = 0, o_q = align_up(o_idx + nf * 4, 256), o_rows = align_up(o_q + nf * dim_ * 4, 256),
                     o_scores = align_up(o_rows + nf * k * 4, 256), o_counts = align_up(o_scores + nf * k * 4, 256),
                     total = align_up(o_counts + nf * 4, 256);
        // (the int8 filter hands its leftovers to a nested f16-filter call, which may itself use mf_fallback_)
        DeviceBuffer& fbuf = p.i8f ? mf_fallback2_ : mf_fallback_;
        FSGPU_TRY(fbuf.reserve(total));
        unsigned char* base = static_cast<unsigned char*>(fbuf.ptr);
        uint32_t* idx_dev = reinterpret_cast<uint32_t*>(base + o_idx);
        float* q_dev = reinterpret_cast<float*>(base + o_q);
        uint32_t* rows_dev = reinterpret_cast<uint32_t*>(base + o_rows);
        float* scores_dev = reinterpret_cast<float*>(base + o_scores);
        uint32_t* counts_dev = reinterpret_cast<uint32_t*>(base + o_counts);
        FSGPU_HIP(hipMemcpyAsync(idx_dev, fb.data(), nf * 4, hipMemcpyHostToDevice, stream));
        FSGPU_HIP(hipStreamSynchronize(stream));  // fb is a stack-owned pageable buffer
        FSGPU_HIP(launch_gather_queries(p.queries_dev, idx_dev, (uint32_t)nf, dim_, p.qs, q_dev, stream));
        if (p.i8f && nf > 8) {
            // rows within the int8 margin of the k-th best did not fit the lists (or the query cannot be certified on the int8
            // slab at all): the f16 filter, whose margin is ~20 x narrower, answers these as a batch of its own
            uint32_t inner_fb = 0;
            hard_batch_ = true;
            const SearchError inner = batched_impl(q_dev, (uint32_t)nf, p.query_len, k, p.allow_dev, rows_dev, scores_dev, counts_dev, stream,
                                                   &inner_fb, nullptr, 0, 0, false, nullptr);
            hard_batch_ = false;
            FSGPU_TRY(inner);
            if (p.refiltered) *p.refiltered = (uint32_t)nf;
            if (p.fallbacks) *p.fallbacks = inner_fb;
            FSGPU_HIP(launch_scatter_hits(idx_dev, (uint32_t)nf, k, rows_dev, scores_dev, counts_dev, p.out_rows_dev,
                                          p.out_scores_dev, p.out_counts_dev, reinterpret_cast<u64*>(p.out_packed_dev), stream));
            return ok();
        }
        if (p.i8f && p.refiltered) *p.refiltered = (uint32_t)nf;
        FSGPU_TRY(fused_search(q_dev, (uint32_t)nf, k, k_eff, p.allow_dev, rows_dev, scores_dev, counts_dev, nullptr, stream));
        FSGPU_HIP(launch_scatter_hits(idx_dev, (uint32_t)nf, k, rows_dev, scores_dev, counts_dev, p.out_rows_dev,
                                      p.out_scores_dev, p.out_counts_dev, reinterpret_cast<u64*>(p.out_packed_dev), stream));
    }
    if (p.fallbacks) *p.fallbacks = total_fallbacks;
    return ok();
}

// The batched search in two halves (fsgpu_search_topk_batched_device_begin / _end): begin enqueues everything and returns a ticket;
// end waits for THAT search's last kernel (an event — not the stream, which may already hold the caller's next search), reads the
// verdicts and answers the rare uncertified query.  Two tickets at most; queries and outputs stay the caller's until end.
SearchError VectorIndex::search_top_k_batched_device_begin(const float* queries_dev, uint32_t nq, uint32_t query_len, uint32_t k,
                                                           const uint64_t* allow_dev, uint32_t* out_rows_dev, float* out_scores_dev,
                                                           uint32_t* out_counts_dev, hipStream_t stream, uint64_t* out_packed_dev,
                                                           int32_t* ticket) {
    int t = -1;
    for (int i = 0; i < 2; ++i)
        if (async_state_[i] == 0) {
            t = i;
            break;
        }
    if (t < 0) return make_error(FSGPU_ERR_INVALID_CONFIG, "two begun batched searches are outstanding: end one first");
    async_state_[t] = 2;   // (complete unless batched_impl parks its plan: shapes answered by the per-query kernels finish inside)
    async_i8f_[t] = false;
    async_nq_[t] = nq;
    async_fb_[t] = 0;
    async_want_ = t;
    const SearchError e = search_top_k_batched_device(queries_dev, nq, query_len, k, allow_dev, out_rows_dev, out_scores_dev, out_counts_dev,
                                                      stream, &async_fb_[t], out_packed_dev);
    async_want_ = -1;
    if (!e.ok()) {
        async_state_[t] = 0;
        return e;
    }
    *ticket = t;
    return ok();
}

SearchError VectorIndex::search_top_k_batched_device_end(int32_t ticket, uint32_t* fallbacks) {
    if (ticket < 0 || ticket > 1 || async_state_[ticket] == 0) return make_error(FSGPU_ERR_INVALID_CONFIG, "no such begun batched search");
    const int t = ticket;
    if (async_state_[t] == 1) {
        FSGPU_HIP(hipSetDevice(device_));
        FSGPU_HIP(hipEventSynchronize(async_ev_[t]));
        BatchedPlan p;
        std::memcpy(&p, async_plan_[t].data(), sizeof(BatchedPlan));
        uint32_t refiltered = 0;
        p.refiltered = async_i8f_[t] ? &refiltered : nullptr;   // (begin's were the addresses of its own locals)
        p.fallbacks = &async_fb_[t];
        async_state_[t] = 0;   // (before the fallback: it may search again, blocking, on this index)
        FSGPU_TRY(batched_fallback(p, true));
        if (async_i8f_[t]) i8f_account(async_nq_[t], refiltered);
    }
    async_state_[t] = 0;
    if (fallbacks) *fallbacks = async_fb_[t];
    return ok();
}

SearchError VectorIndex::search_top_k_batched(const float* queries, uint32_t nq, uint32_t query_len, uint32_t k,
                                              const uint64_t* allow, uint32_t* out_rows, float* out_scores,
                                              uint32_t* out_counts, uint32_t* fallbacks, const uint64_t* allow_resident_dev,
                                              bool queries_on_device) {
    if (fallbacks) *fallbacks = 0;
    FSGPU_TRY(ensure_query_dimension(query_len));
    if (nq == 0) return ok();
    if (k == 0 || nrows_ == 0) {
        for (uint32_t q = 0; q < nq; ++q) out_counts[q] = 0;
        return ok();
    }
    FSGPU_HIP(hipSetDevice(device_));
    const size_t qbytes = (size_t)nq * dim_ * 4;
    FSGPU_TRY(ws_rows_.reserve((size_t)nq * k * 4));
    FSGPU_TRY(ws_scores_.reserve((size_t)nq * k * 4));
    FSGPU_TRY(ws_counts_.reserve((size_t)nq * 4));
    const float* q_dev = queries;
    if (!queries_on_device) {
        FSGPU_TRY(ws_queries_.reserve(qbytes));
        FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, queries, qbytes, hipMemcpyHostToDevice, stream_));
        q_dev = static_cast<const float*>(ws_queries_.ptr);
    }
    const uint64_t* allow_dev = allow ? allow_resident_dev : nullptr;
    if (allow && !allow_dev) {
        const size_t words = (size_t)((nrows_ + 63) / 64);
        FSGPU_TRY(ws_allow_.reserve(words * 8));
        FSGPU_HIP(hipMemcpyAsync(ws_allow_.ptr, allow, words * 8, hipMemcpyHostToDevice, stream_));
        allow_dev = static_cast<const uint64_t*>(ws_allow_.ptr);
    }
    FSGPU_TRY(search_top_k_batched_device(q_dev, nq, query_len, k, allow_dev,
                                          static_cast<uint32_t*>(ws_rows_.ptr), static_cast<float*>(ws_scores_.ptr),
                                          static_cast<uint32_t*>(ws_counts_.ptr), stream_, fallbacks));
    FSGPU_HIP(hipMemcpyAsync(out_rows, ws_rows_.ptr, (size_t)nq * k * 4, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipMemcpyAsync(out_scores, ws_scores_.ptr, (size_t)nq * k * 4, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipMemcpyAsync(out_counts, ws_counts_.ptr, (size_t)nq * 4, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipStreamSynchronize(stream_));
    return ok();
}

SearchError VectorIndex::search_top_k_int8_batched(const float* queries, uint32_t nq, uint32_t query_len, uint32_t k,
                                                   uint32_t multiplier, uint32_t* out_rows, float* out_scores,
                                                   uint32_t* out_counts, uint32_t* fallbacks, int bits) {
    if (fallbacks) *fallbacks = 0;
    FSGPU_TRY(ensure_query_dimension(query_len));
    if (nq == 0) return ok();
    // what the fast path does not cover goes through the per-query search, like the reference (search.rs:579-585);
    // an index with a doc-id table also does (resolve_hits dedups by doc id there)
    if (k == 0 || nrows_ == 0 || !wal_.empty() || has_doc_ids()) {
        for (uint32_t i = 0; i < nq; ++i)
            FSGPU_TRY(bits == 4 ? search_top_k_4bit_two_pass(queries + (size_t)i * dim_, query_len, k, multiplier,
                                                             out_rows + (size_t)i * k, out_scores + (size_t)i * k, &out_counts[i])
                                : search_top_k_int8_two_pass(queries + (size_t)i * dim_, query_len, k, multiplier,
                                                             out_rows + (size_t)i * k, out_scores + (size_t)i * k, &out_counts[i]));
        if (fallbacks) *fallbacks = nq;
        return ok();
    }
    FSGPU_HIP(hipSetDevice(device_));
    // dedicated staging: the per-query fallback inside reuses the ws_* workspaces
    auto align_up = [](size_t v, size_t a) { return (v + a - 1) / a * a; };
    const size_t o_q = 0, o_rows = align_up((size_t)nq * dim_ * 4, 256), o_scores = align_up(o_rows + (size_t)nq * k * 4, 256),
                 o_counts = align_up(o_scores + (size_t)nq * k * 4, 256), total = align_up(o_counts + (size_t)nq * 4, 256);
    FSGPU_TRY(mf_io_.reserve(total));
    unsigned char* base = static_cast<unsigned char*>(mf_io_.ptr);
    float* q_dev = reinterpret_cast<float*>(base + o_q);
    uint32_t* rows_dev = reinterpret_cast<uint32_t*>(base + o_rows);
    float* scores_dev = reinterpret_cast<float*>(base + o_scores);
    uint32_t* counts_dev = reinterpret_cast<uint32_t*>(base + o_counts);
    FSGPU_HIP(hipMemcpyAsync(q_dev, queries, (size_t)nq * dim_ * 4, hipMemcpyHostToDevice, stream_));
    FSGPU_TRY(search_top_k_int8_batched_device(q_dev, nq, query_len, k, multiplier, rows_dev, scores_dev, counts_dev,
                                               stream_, fallbacks, bits));
    FSGPU_HIP(hipMemcpyAsync(out_rows, rows_dev, (size_t)nq * k * 4, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipMemcpyAsync(out_scores, scores_dev, (size_t)nq * k * 4, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipMemcpyAsync(out_counts, counts_dev, (size_t)nq * 4, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipStreamSynchronize(stream_));
    return ok();
}

// A strided view of this index: same slab, same live bitmap, rows read over their first `dims` dimensions only.
VectorIndex* VectorIndex::mrl_view(uint32_t dims) {
    auto it = views_.find(dims);
    if (it == views_.end()) {
        auto v = std::make_unique<VectorIndex>();
        if (!v->init_device(device_, dims, nrows_, slab_dev_, live_dev_, row_base_).ok()) return nullptr;
        v->row_stride_ = dim_ * (f32_ ? 4 : 2);
        v->f32_ = f32_;
        it = views_.emplace(dims, std::move(v)).first;
    }
    VectorIndex* v = it->second.get();
    v->slab_dev_ = slab_dev_;   // re-bind: the live bitmap may have been re-uploaded since the view was made
    v->live_dev_ = live_dev_;
    v->hreduce = hreduce;
    v->profiling = profiling;
    return v;
}

SearchError VectorIndex::ensure_replicas() {
    while (replicas_.size() + 1 < kLanes) {
        auto v = std::make_unique<VectorIndex>();
        FSGPU_TRY(v->init_device(device_, dim_, nrows_, slab_dev_, live_dev_, row_base_));
        v->row_stride_ = row_stride_;
        v->f32_ = f32_;
        replicas_.push_back(std::move(v));
    }
    sync_replicas();
    return ok();
}

void VectorIndex::sync_replicas() {
    for (auto& v : replicas_) {
        v->slab_dev_ = slab_dev_;
        v->live_dev_ = live_dev_;
        v->hreduce = hreduce;
        v->variant = variant;
        v->int8_latency = int8_latency;       // (every lane answers a call the same way, whichever one it lands on)
        v->batched_filter = batched_filter;
    }
}

// VectorIndex::mrl_search_with_stats (crates/frankensearch-index/src/mrl.rs:241-395).
SearchError VectorIndex::mrl_search(const float* query, uint32_t query_len, uint32_t k, uint32_t search_dims,
                                    uint32_t rescore_dims, uint32_t rescore_top_k, uint32_t* out_rows, float* out_scores,
                                    uint32_t* out_count, MrlStats* stats) {
    *out_count = 0;
    MrlStats st;
    FSGPU_TRY(ensure_query_dimension(query_len));
    if (search_dims == 0) return make_error(FSGPU_ERR_INVALID_CONFIG, "search_dims must be at least 1");
    if (search_dims >= dim_) {  // no truncation benefit: the standard search (mrl.rs:283-296)
        st.scan_dims = st.rescore_dims = dim_;
        st.records_scanned = nrows_ + wal_.size();
        st.fell_back_to_full = true;
        if (stats) *stats = st;
        if (has_doc_ids()) return search_hits(query, query_len, k, out_rows, out_scores, out_count);
        if (k == 0 || nrows_ == 0) return ok();
        return search_top_k(query, 1, query_len, k, nullptr, out_rows, out_scores, out_count);
    }
    if (k == 0 || (nrows_ == 0 && wal_.empty())) {
        if (stats) *stats = st;
        return ok();
    }
    uint32_t rdims = (rescore_dims == 0 || rescore_dims > dim_) ? dim_ : rescore_dims;  // mrl.rs:92-105
    if (rdims < search_dims) rdims = search_dims;
    const uint64_t rtop64 = rescore_top_k ? rescore_top_k : (uint64_t)k * 3;             // mrl.rs:108-114
    const uint32_t rtop = (uint32_t)std::min<uint64_t>(rtop64, 0x7fffffffull);
    struct Cand {
        uint64_t index;  // main row, or WAL-tagged (top bit, wal.rs:557-569)
        float score;
    };
    const uint64_t wal_tag = 1ull << 63;
    std::vector<Cand> cand;
    // phase 1: truncated scan of the main rows on the GPU (prefix view), top rtop
    if (nrows_ > 0) {
        VectorIndex* view = mrl_view(search_dims);
        if (!view) return make_error(FSGPU_ERR_DEVICE, "cannot create the truncated view");
        std::vector<uint32_t> rows(rtop);  // search_top_k pads every query's output to k entries
        std::vector<float> scores(rtop);
        uint32_t count = 0;
        FSGPU_TRY(view->search_top_k(query, 1, search_dims, rtop, nullptr, rows.data(), scores.data(), &count));
        for (uint32_t i = 0; i < count; ++i) cand.push_back(Cand{rows[i], scores[i]});
        // the view's timed launches count as this index's (fsgpu_index_scan_stats)
        for (auto& ev : view->events_) events_.push_back(ev);
        view->events_.clear();
        profiled_rows_ += view->profiled_rows_;
        view->profiled_rows_ = 0;
    }
    // resident WAL entries: truncated f32 dot, non-finite scores skipped (mrl.rs:539-583)
    for (size_t w = 0; w < wal_.size(); ++w) {
        const float s = dot_f32_f32(wal_[w].embedding.data(), query, search_dims, hreduce);
        if (!std::isfinite(s)) continue;
        cand.push_back(Cand{wal_tag | w, s});
    }
    auto best_first = [](const Cand& a, const Cand& b) {
        const uint32_t ka = host_score_ord(a.score), kb = host_score_ord(b.score);
        if (ka != kb) return ka > kb;
        return a.index < b.index;
    };
    std::sort(cand.begin(), cand.end(), best_first);
    if (cand.size() > rtop) cand.resize(rtop);
    st.scan_dims = search_dims;
    st.rescore_dims = rdims;
    st.candidates_rescored = (uint32_t)cand.size();
    st.records_scanned = nrows_ + wal_.size();
    // phase 2: rescore over rdims (mrl.rs:587-618)
    std::vector<uint32_t> main_rows;
    for (const Cand& c : cand)
        if (!(c.index & wal_tag)) main_rows.push_back((uint32_t)c.index);
    std::vector<float> main_scores(main_rows.size());
    if (!main_rows.empty()) {
        if (rdims == dim_) {
            FSGPU_TRY(gather_dot(query, dim_, main_rows.data(), (uint32_t)main_rows.size(), main_scores.data()));
        } else {
            VectorIndex* rv = mrl_view(rdims);
            if (!rv) return make_error(FSGPU_ERR_DEVICE, "cannot create the rescore view");
            FSGPU_TRY(rv->gather_dot(query, rdims, main_rows.data(), (uint32_t)main_rows.size(), main_scores.data()));
        }
    }
    size_t mi = 0;
    for (Cand& c : cand) {
        if (c.index & wal_tag) c.score = dot_f32_f32(wal_[(size_t)(c.index & ~wal_tag)].embedding.data(), query, rdims, hreduce);
        else c.score = main_scores[mi++];
    }
    std::sort(cand.begin(), cand.end(), best_first);
    if (cand.size() > k) cand.resize(k);
    // resolve_mrl_hits (mrl.rs:642-683): WAL hits at the virtual index, deleted rows dropped; no dedup, no shadowing
    uint32_t n = 0;
    for (const Cand& c : cand) {
        if (c.index & wal_tag) {
            out_rows[n] = (uint32_t)(nrows_ + (c.index & ~wal_tag));
        } else {
            const size_t r = (size_t)(c.index - row_base_);
            if (!live_host_.empty() && !((live_host_[r >> 6] >> (r & 63)) & 1ull)) continue;
            out_rows[n] = (uint32_t)c.index;
        }
        out_scores[n] = c.score;
        ++n;
    }
    *out_count = n;
    if (stats) *stats = st;
    return ok();
}

// mrl_search for a whole batch (mrl.rs:241-395 per query): phase 1 is the batched matrix-core scan of the prefix view (the
// first search_dims dimensions of every row, rows dim_ * 2 bytes apart, full-length queries read through a stride) with
// k = rescore_top_k — exact top-rtop of the truncated scores, as the per-query scan gives; phase 2 re-scores each query's
// candidates over rescore_dims in one select_kernel launch (exact-order dot, best k emitted).  Indexes with resident WAL
// entries, F32 slabs and shapes the matrix-core path does not cover are answered query by query.
SearchError VectorIndex::mrl_search_batched(const float* queries, uint32_t nq, uint32_t query_len, uint32_t k, uint32_t search_dims,
                                            uint32_t rescore_dims, uint32_t rescore_top_k, uint32_t* out_rows, float* out_scores,
                                            uint32_t* out_counts, uint32_t* fallbacks) {
    if (fallbacks) *fallbacks = 0;
    FSGPU_TRY(ensure_query_dimension(query_len));
    if (search_dims == 0) return make_error(FSGPU_ERR_INVALID_CONFIG, "search_dims must be at least 1");
    if (nq == 0) return ok();
    uint32_t rdims = (rescore_dims == 0 || rescore_dims > dim_) ? dim_ : rescore_dims;  // mrl.rs:92-105
    if (rdims < search_dims) rdims = search_dims;
    const uint64_t rtop64 = rescore_top_k ? rescore_top_k : (uint64_t)k * 3;             // mrl.rs:108-114
    const bool fast = search_dims < dim_ && k >= 1 && k <= 64 && rtop64 >= 1 && rtop64 <= 64 && wal_.empty() && !f32_ && nrows_ > 0 &&
                      scan_mfma_supported((int)search_dims) && (rdims % 8 == 0) && nrows_ >= 4 * 8192ull && variant != 4 &&
                      row_stride_ == 0;
    if (!fast) {
        for (uint32_t i = 0; i < nq; ++i)
            FSGPU_TRY(mrl_search(queries + (size_t)i * dim_, query_len, k, search_dims, rescore_dims, rescore_top_k,
                                 out_rows + (size_t)i * k, out_scores + (size_t)i * k, &out_counts[i], nullptr));
        if (fallbacks) *fallbacks = nq;
        return ok();
    }
    const uint32_t rtop = (uint32_t)rtop64;
    VectorIndex* view = mrl_view(search_dims);
    if (!view) return make_error(FSGPU_ERR_DEVICE, "cannot create the truncated view");
    FSGPU_HIP(hipSetDevice(device_));
    auto align_up = [](size_t v, size_t a) { return (v + a - 1) / a * a; };
    const size_t o_q = 0, o_packed = align_up((size_t)nq * dim_ * 4, 256), o_zero = align_up(o_packed + (size_t)nq * rtop * 8, 256),
                 o_rows = align_up(o_zero + (size_t)nq * 4, 256), o_scores = align_up(o_rows + (size_t)nq * k * 4, 256),
                 o_counts = align_up(o_scores + (size_t)nq * k * 4, 256), total = align_up(o_counts + (size_t)nq * 4, 256);
    FSGPU_TRY(mf_io_.reserve(total));
    unsigned char* base = static_cast<unsigned char*>(mf_io_.ptr);
    float* q_dev = reinterpret_cast<float*>(base + o_q);
    u64* packed = reinterpret_cast<u64*>(base + o_packed);
    float* zero = reinterpret_cast<float*>(base + o_zero);
    uint32_t* rows_dev = reinterpret_cast<uint32_t*>(base + o_rows);
    float* scores_dev = reinterpret_cast<float*>(base + o_scores);
    uint32_t* counts_dev = reinterpret_cast<uint32_t*>(base + o_counts);
    FSGPU_HIP(hipMemcpyAsync(q_dev, queries, (size_t)nq * dim_ * 4, hipMemcpyHostToDevice, stream_));
    FSGPU_HIP(hipMemsetAsync(zero, 0, (size_t)nq * 4, stream_));
    view->hreduce = hreduce;
    uint32_t fb = 0;
    FSGPU_TRY(view->batched_impl(q_dev, nq, search_dims, rtop, nullptr, nullptr, nullptr, nullptr, stream_, &fb,
                                 reinterpret_cast<uint64_t*>(packed), 0, dim_, false, nullptr));
    for (auto& ev : view->events_) events_.push_back(ev);   // the view's timed launches count as this index's
    view->events_.clear();
    profiled_rows_ += view->profiled_rows_;
    view->profiled_rows_ = 0;
    // phase 2 (mrl.rs:587-618): every candidate re-scored over rdims in the reference's order, best k per query
    SelectArgs s{};
    s.lists = packed;
    s.q_stride = rtop;
    s.l_stride = rtop;
    s.nlists = 1;
    s.list_len = rtop;
    s.k = rtop;
    s.take_topk = 1;          // the list IS the candidate set
    s.delta = zero;
    s.slab = slab_dev_;
    s.queries = q_dev;
    s.dim = rdims;
    s.row_stride = rdims == dim_ ? 0 : dim_ * 2;
    s.query_stride = dim_;
    s.nrows = (uint32_t)nrows_;
    s.row_base = (uint32_t)row_base_;
    s.hreduce = hreduce;
    s.k_out = (uint32_t)std::min<uint64_t>(k, nrows_);
    s.out_stride = k;
    s.out_rows = rows_dev;
    s.out_scores = scores_dev;
    s.out_counts = counts_dev;
    FSGPU_HIP(launch_select(s, (int)nq, stream_));
    FSGPU_HIP(hipMemcpyAsync(out_rows, rows_dev, (size_t)nq * k * 4, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipMemcpyAsync(out_scores, scores_dev, (size_t)nq * k * 4, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipMemcpyAsync(out_counts, counts_dev, (size_t)nq * 4, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipStreamSynchronize(stream_));
    if (fallbacks) *fallbacks = fb;
    return ok();
}

// search_top_k_int8_two_pass_impl (crates/frankensearch-index/src/search.rs:589-661)
SearchError VectorIndex::search_top_k_int8_two_pass(const float* query, uint32_t query_len, uint32_t k,
                                                    uint32_t multiplier, uint32_t* out_rows, float* out_scores,
                                                    uint32_t* out_count) {
    return quantized_two_pass(query, query_len, k, multiplier, 8, out_rows, out_scores, out_count);
}

// search_top_k_4bit_two_pass (crates/frankensearch-index/src/search.rs:876-946)
SearchError VectorIndex::search_top_k_4bit_two_pass(const float* query, uint32_t query_len, uint32_t k,
                                                    uint32_t multiplier, uint32_t* out_rows, float* out_scores,
                                                    uint32_t* out_count) {
    return quantized_two_pass(query, query_len, k, multiplier, 4, out_rows, out_scores, out_count);
}

// The two-pass searches' lane for ONE caller (see quantized_two_pass).  query / qi: the f32 query and its quantised form (host);
// rows / scores: [k] on the host.  *answered = false: nothing was written.
SearchError VectorIndex::two_pass_lone_certified(const float* query, const unsigned char* qi, uint32_t qbytes, uint32_t k, uint32_t k_eff,
                                                 uint32_t cc, int bits, const void* qslab, uint32_t* rows, float* scores, uint32_t* count,
                                                 bool* answered) {
    *answered = false;
    bool enqueued = false;
    FSGPU_TRY(two_pass_lone_enqueue(query, qi, qbytes, k, k_eff, cc, bits, qslab, false, &enqueued));
    if (!enqueued) return ok();
    return two_pass_lone_check(rows, scores, count, nullptr, nullptr, answered);
}

// Enqueue only: pass 1 keeping 32 entries per block, the cut, the cc best pass-1 entries (best first, into pinned memory), their exact
// scores, the k best of those.  want_pairs: the candidates' exact entries go to pinned memory as well, aligned with the pass-1 entries
// (what a row-sharded handle's root merges).
SearchError VectorIndex::two_pass_lone_enqueue(const float* query, const unsigned char* qi, uint32_t qbytes, uint32_t k, uint32_t k_eff,
                                               uint32_t cc, int bits, const void* qslab, bool want_pairs, bool* enqueued) {
    *enqueued = false;
    constexpr uint32_t LK = 32;
    const bool fused = bits == 8 ? scan_i8_fused_supported((int)dim_, 64) : scan_4bit_fused_supported((int)dim_, 64);
    if (!fused || pinned_io() == nullptr) return ok();
    const size_t fbytes = (size_t)dim_ * 4;
    const size_t o_qi = (fbytes + 255) & ~(size_t)255, o_out = (o_qi + qbytes + 255) & ~(size_t)255,
                 o_flags = (o_out + (size_t)k * 8 + 4 + 255) & ~(size_t)255;
    const size_t o_approx = (o_flags + 64 + 255) & ~(size_t)255, o_exact = (o_approx + (size_t)cc * 8 + 255) & ~(size_t)255;
    if (o_exact + (size_t)cc * 8 > kPinnedIoBytes) return ok();
    unsigned char* io = static_cast<unsigned char*>(io_host_);
    std::memcpy(io, query, fbytes);
    std::memcpy(io + o_qi, qi, qbytes);
    const float* q_pin = reinterpret_cast<const float*>(io);
    float* delta_pin = reinterpret_cast<float*>(io + o_flags);   // the pass-1 scores are the reference's own: no margin
    float* cut_pin = delta_pin + 2;
    *delta_pin = 0.f;
    int grid = num_cus_;   // 256 lists x 32 entries: what the sorted selection holds in one piece
    const int max_useful = (int)(((nrows_ + 15) / 16 + 3) / 4);
    grid = std::max(1, std::min(grid, max_useful));
    if ((size_t)grid * LK > 8192) return ok();
    // (every wave of the scan reads the whole quantised query: from device memory, not over the bus; the finish's one block reads
    // the f32 query where it lies)
    FSGPU_TRY(ws_i8_query_.reserve(qbytes));
    FSGPU_TRY(ws_partial_.reserve((size_t)grid * LK * 8));
    FSGPU_TRY(ws_cand_packed_.reserve((size_t)cc * 8));
    FSGPU_TRY(ws_cand_rows_.reserve((size_t)cc * 4));
    FSGPU_TRY(ws_cand_scores_.reserve((size_t)cc * 4));
    FSGPU_HIP(hipMemcpyAsync(ws_i8_query_.ptr, io + o_qi, qbytes, hipMemcpyHostToDevice, stream_));
    ScanArgs a = base_args(q_pin, nullptr);
    a.partial = static_cast<u64*>(ws_partial_.ptr);
    a.k = LK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (profiling) {
        FSGPU_HIP(hipEventCreate(&e0));
        FSGPU_HIP(hipEventCreate(&e1));
        FSGPU_HIP(hipEventRecord(e0, stream_));
    }
    if (bits == 8) FSGPU_HIP(launch_scan_i8(a, qslab, ws_i8_query_.ptr, 64, grid, stream_, nullptr));
    else FSGPU_HIP(launch_scan_4bit(a, qslab, ws_i8_query_.ptr, 64, grid, stream_, nullptr));
    if (profiling) {
        FSGPU_HIP(hipEventRecord(e1, stream_));
        events_.emplace_back(e0, e1);
        profiled_rows_ += nrows_;
    }
    FSGPU_HIP(launch_list_cut(a.partial, (uint32_t)grid, LK, cut_pin, stream_));
    // the cc best pass-1 entries of the 8,192 kept (ONE pass of the merge; the selection's sorted finish took 0.10 ms here), their exact
    // scores, the k best of those — the general sequence's kernels over lists a third as long
    u64* approx_pin = reinterpret_cast<u64*>(io + o_approx);
    u64* exact_pin = reinterpret_cast<u64*>(io + o_exact);
    uint32_t* cand_rows = static_cast<uint32_t*>(ws_cand_rows_.ptr);
    float* cand_scores = static_cast<float*>(ws_cand_scores_.ptr);
    u64* cand_packed = static_cast<u64*>(ws_cand_packed_.ptr);
    MergeArgs m;
    m.lists = a.partial;
    m.q_stride = (uint64_t)grid * LK;
    m.l_stride = LK;
    m.nlists = (uint32_t)grid;
    m.list_len = LK;
    m.k = cc;
    m.out_stride = cc;
    m.out_rows = cand_rows;
    m.out_scores = nullptr;
    m.out_counts = nullptr;
    m.out_packed = approx_pin;   // best first: the certificate reads the last one
    FSGPU_HIP(launch_merge_topk(m, 1, stream_));
    FSGPU_HIP(hipMemsetAsync(cand_scores, 0, (size_t)cc * 4, stream_));
    FSGPU_HIP(launch_gather_dot(a, cand_rows, cc, cand_scores, stream_));
    FSGPU_HIP(launch_pack_hits(cand_rows, cand_scores, cc, cand_packed, stream_));
    if (want_pairs) FSGPU_HIP(launch_pack_hits(cand_rows, cand_scores, cc, exact_pin, stream_));
    MergeArgs m2;
    m2.lists = cand_packed;
    m2.q_stride = cc;
    m2.l_stride = cc;
    m2.nlists = 1;
    m2.list_len = cc;
    m2.k = k_eff;
    m2.out_stride = k;
    m2.out_rows = reinterpret_cast<uint32_t*>(io + o_out);
    m2.out_scores = reinterpret_cast<float*>(io + o_out + (size_t)k * 4);
    m2.out_counts = reinterpret_cast<uint32_t*>(io + o_out + (size_t)k * 8);
    m2.out_packed = nullptr;
    m2.lists_sorted = 0;  // candidates arrive in pass-1 order
    FSGPU_HIP(launch_merge_topk(m2, 1, stream_));
    tp_lane_ = TwoPassLane{k, cc, o_out, o_flags, o_approx, o_exact};
    *enqueued = true;
    return ok();
}

// The other half: one synchronisation, then the certificate — complete when no list was full (nothing dropped) or the cc-th best entry
// outranks everything dropped, STRICTLY: a dropped row with the same integer score may have the smaller row id.
// approx_out / exact_out (may be null): the cc candidate pairs.
SearchError VectorIndex::two_pass_lone_check(uint32_t* rows, float* scores, uint32_t* count, u64* approx_out, u64* exact_out, bool* answered) {
    *answered = false;
    const TwoPassLane L = tp_lane_;
    unsigned char* io = static_cast<unsigned char*>(io_host_);
    FSGPU_HIP(hipSetDevice(device_));
    FSGPU_HIP(hipStreamSynchronize(stream_));
    const float cut = *(reinterpret_cast<const float*>(io + L.o_flags) + 2);
    const u64* approx_pin = reinterpret_cast<const u64*>(io + L.o_approx);
    bool complete = cut == -INFINITY;
    if (!complete && approx_pin[L.cc - 1] != ~0ull) {
        float tau;
        const uint32_t tb = (uint32_t)(approx_pin[L.cc - 1] >> 32);
        std::memcpy(&tau, &tb, 4);
        complete = cut < tau;
    }
    if (!complete) return ok();
    if (rows) std::memcpy(rows, io + L.o_out, (size_t)L.k * 4);
    if (scores) std::memcpy(scores, io + L.o_out + (size_t)L.k * 4, (size_t)L.k * 4);
    if (count) *count = *reinterpret_cast<const uint32_t*>(io + L.o_out + (size_t)L.k * 8);
    if (approx_out) std::memcpy(approx_out, approx_pin, (size_t)L.cc * 8);
    if (exact_out) std::memcpy(exact_out, io + L.o_exact, (size_t)L.cc * 8);
    *answered = true;
    return ok();
}

// quantize_i8_query (search.rs:1616-1626) / pack_4bit_query (:1640-1653): the query's own max-abs scale, round half away from zero,
// clamp; NaN -> 0
static void quantize_query_host(const float* query, uint32_t dim, int bits, std::vector<unsigned char>& qi) {
    const uint32_t qbytes = bits == 8 ? dim : (dim + 1) / 2;
    qi.assign(qbytes, 0);
    float max_abs = 0.f;
    for (uint32_t i = 0; i < dim; ++i) {
        const float v = std::fabs(query[i]);
        if (v > max_abs) max_abs = v;
    }
    const float lim = bits == 8 ? 127.0f : 7.0f;
    const bool usable = bits == 8 ? max_abs > 0.f : max_abs > 1e-9f;
    const float scale = usable ? lim / max_abs : 0.f;
    if (bits == 4 || usable) {
        for (uint32_t i = 0; i < dim; ++i) {
            float v = std::round(query[i] * scale);
            if (v != v) v = 0.f;
            v = std::min(std::max(v, -lim), lim);
            const int qv = (int)v;
            if (bits == 8) qi[i] = (unsigned char)(signed char)qv;
            else qi[i / 2] |= (unsigned char)((qv & 0xF) << ((i & 1) ? 4 : 0));
        }
    }
}

// The quantised copy a two-pass search scans, built lazily, once (VectorIndex::int8_slab() / nibbles_slab(), search.rs:988-1000).
SearchError VectorIndex::ensure_two_pass_slab(int bits, const void** qslab) {
    const size_t n = (size_t)nrows_;
    const uint32_t qbytes = bits == 8 ? dim_ : (dim_ + 1) / 2;
    if (bits == 8 && !i8_ready_) {
        FSGPU_TRY(i8_slab_.reserve(n * dim_));
        FSGPU_TRY(i8_max_.reserve(4));
        FSGPU_HIP(launch_quantize_slab_i8(slab_dev_, n * dim_, static_cast<unsigned int*>(i8_max_.ptr), i8_slab_.ptr,
                                          stream_, quant_max_ready_));
        i8_ready_ = true;
    }
    if (bits == 4 && !n4_ready_) {
        FSGPU_TRY(n4_slab_.reserve(n * qbytes));
        FSGPU_TRY(i8_max_.reserve(4));
        FSGPU_HIP(launch_pack_slab_4bit(slab_dev_, nrows_, dim_, static_cast<unsigned int*>(i8_max_.ptr), n4_slab_.ptr,
                                        stream_, quant_max_ready_));
        n4_ready_ = true;
    }
    *qslab = bits == 8 ? i8_slab_.ptr : n4_slab_.ptr;
    return ok();
}

// One query of a row-sharded two-pass search, this shard's half, in two halves: begin enqueues (the lone caller's lane when the
// shape allows, else the batched sequence with one query), end yields the shard's cc_out = max(k * multiplier, k) candidate pairs
// (pass-1 entry, exact entry; kEmpty beyond the candidates) — what two_pass_candidates_device yields for one query.
SearchError VectorIndex::lone_two_pass_begin(const float* query, uint32_t k, uint32_t multiplier, int bits) {
    lone_ = LoneState{};
    lone_.query = query;
    lone_.k = k;
    lone_.mult = multiplier ? multiplier : 1;
    lone_.bits = bits == 4 ? 4 : 8;
    const uint64_t cc_out64 = std::max<uint64_t>((uint64_t)k * lone_.mult, k);
    if (cc_out64 > 256 || k == 0) return make_error(FSGPU_ERR_INVALID_CONFIG, "sharded two-pass: 1 <= k, k * multiplier <= 256");
    lone_.cc_out = (uint32_t)cc_out64;
    if (f32_) return make_error(FSGPU_ERR_INVALID_CONFIG, "two-pass searches need an F16 slab");
    if (nrows_ == 0) {
        lone_.kind = kLoneEmpty;
        return ok();
    }
    FSGPU_HIP(hipSetDevice(device_));
    const void* qslab = nullptr;
    FSGPU_TRY(ensure_two_pass_slab(lone_.bits, &qslab));
    uint64_t cc64 = std::min<uint64_t>((uint64_t)k * lone_.mult, nrows_);
    cc64 = std::max<uint64_t>(cc64, std::min<uint64_t>(k, nrows_));
    const uint32_t cc = (uint32_t)cc64, k_eff = (uint32_t)std::min<uint64_t>(k, nrows_);
    lone_.cc = cc;
    if (cc <= kSelectMaxK && k_eff <= 64 && k <= 64 && (dim_ & 7) == 0 && nrows_ >= 4096 && !(row_stride_ && row_stride_ != dim_ * 2) && variant == 0) {
        if (tp_skip_ > 0) {
            --tp_skip_;
        } else {
            std::vector<unsigned char> qi;
            quantize_query_host(query, dim_, lone_.bits, qi);
            bool enqueued = false;
            FSGPU_TRY(two_pass_lone_enqueue(query, qi.data(), (uint32_t)qi.size(), k, k_eff, cc, lone_.bits, qslab, true, &enqueued));
            if (enqueued) {
                lone_.kind = kLoneTwoPassLane;
                return ok();
            }
        }
    }
    FSGPU_TRY(ws_pairs_.reserve((size_t)lone_.cc_out * 16));
    if (async_state_[0] != 0 && async_state_[1] != 0) {
        lone_.kind = kLoneTwoPassBlocking;
        return ok();
    }
    FSGPU_TRY(ws_queries_.reserve((size_t)dim_ * 4));
    FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, query, (size_t)dim_ * 4, hipMemcpyHostToDevice, stream_));
    u64* pairs = static_cast<u64*>(ws_pairs_.ptr);
    FSGPU_TRY(two_pass_candidates_device_begin(static_cast<const float*>(ws_queries_.ptr), 1, dim_, k, lone_.mult, lone_.bits,
                                               reinterpret_cast<uint64_t*>(pairs), reinterpret_cast<uint64_t*>(pairs + lone_.cc_out), stream_,
                                               &lone_.ticket));
    lone_.kind = kLoneTwoPassBatched;
    return ok();
}

SearchError VectorIndex::lone_two_pass_end(uint64_t* out_approx, uint64_t* out_exact) {
    const LoneState st = lone_;
    lone_ = LoneState{};
    if (st.kind == kLoneNone) return make_error(FSGPU_ERR_INVALID_CONFIG, "no lone query was begun on this index");
    for (uint32_t i = 0; i < st.cc_out; ++i) out_approx[i] = out_exact[i] = ~0ull;
    if (st.kind == kLoneEmpty) return ok();
    FSGPU_HIP(hipSetDevice(device_));
    bool blocking = st.kind == kLoneTwoPassBlocking;
    if (st.kind == kLoneTwoPassLane) {
        bool answered = false;
        FSGPU_TRY(two_pass_lone_check(nullptr, nullptr, nullptr, reinterpret_cast<u64*>(out_approx), reinterpret_cast<u64*>(out_exact), &answered));
        if (answered) {
            tp_backoff_ = 0;
            return ok();
        }
        tp_backoff_ = tp_backoff_ ? std::min<uint32_t>(tp_backoff_ * 2, 64) : 1;
        tp_skip_ = tp_backoff_;
        FSGPU_TRY(ws_pairs_.reserve((size_t)st.cc_out * 16));
        blocking = true;
    }
    u64* pairs = static_cast<u64*>(ws_pairs_.ptr);
    if (blocking) {   // the general sequence, in one piece (quantized_two_pass hands the pairs on when asked to)
        FSGPU_HIP(hipMemsetAsync(pairs, 0xff, (size_t)st.cc_out * 16, stream_));
        std::vector<uint32_t> rows(st.k);
        std::vector<float> scores(st.k);
        uint32_t cnt = 0;
        FSGPU_TRY(quantized_two_pass(st.query, dim_, st.k, st.mult, st.bits, rows.data(), scores.data(), &cnt, pairs, pairs + st.cc_out));
    } else {
        uint32_t fb = 0;
        FSGPU_TRY(two_pass_candidates_device_end(st.ticket, &fb));
    }
    FSGPU_HIP(hipMemcpyAsync(out_approx, pairs, (size_t)st.cc_out * 8, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipMemcpyAsync(out_exact, pairs + st.cc_out, (size_t)st.cc_out * 8, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipStreamSynchronize(stream_));
    return ok();
}

// Shared body of the int8 (bits = 8) and 4-bit (bits = 4) two-pass searches: quantised pass 1 over the lazily built
// slab, exact f16 rescore of the candidates, best-first selection of k.
SearchError VectorIndex::quantized_two_pass(const float* query, uint32_t query_len, uint32_t k, uint32_t multiplier,
                                            int bits, uint32_t* out_rows, float* out_scores, uint32_t* out_count,
                                            u64* approx_out_dev, u64* exact_out_dev) {
    *out_count = 0;
    // anything the fast path does not cover goes through the exact search (search.rs:579-585)
    if (!approx_out_dev && (k == 0 || nrows_ == 0 || !wal_.empty() || f32_)) {  // ... || quantization != F16
        if (has_doc_ids()) return search_hits(query, query_len, k, out_rows, out_scores, out_count);
        FSGPU_TRY(ensure_query_dimension(query_len));
        if (k == 0 || nrows_ == 0) return ok();
        return search_top_k(query, 1, query_len, k, nullptr, out_rows, out_scores, out_count);
    }
    FSGPU_TRY(ensure_query_dimension(query_len));
    FSGPU_HIP(hipSetDevice(device_));
    const size_t n = (size_t)nrows_;
    const uint32_t qbytes = bits == 8 ? dim_ : (dim_ + 1) / 2;  // quantised bytes per vector
    const void* qslab = nullptr;
    FSGPU_TRY(ensure_two_pass_slab(bits, &qslab));
    const uint64_t mult = multiplier ? multiplier : 1;
    uint64_t cc64 = std::min<uint64_t>((uint64_t)k * mult, nrows_);
    cc64 = std::max<uint64_t>(cc64, std::min<uint64_t>(k, nrows_));
    const uint32_t cc = (uint32_t)cc64;
    const uint32_t k_eff = (uint32_t)std::min<uint64_t>(k, nrows_);
    std::vector<unsigned char> qi;
    quantize_query_host(query, dim_, bits, qi);
    std::vector<uint32_t> rows(k);
    std::vector<float> scores(k);
    uint32_t count = 0;
    bool answered = false;
    // The lone caller's lane: pass 1 keeping 32 entries per block, the selection's finish (candidates = the cc best pass-1 entries,
    // exact re-score, k best) — three launches, nothing copied — certified on the host: the cc-th best pass-1 entry lies strictly
    // above everything a block can have dropped.  Otherwise (and for a row-sharded index's shards, which hand the candidate pairs on)
    // the general sequence below answers; a failed certificate backs off like the exact search's (certified_i8_lone_query).
    // (worth it from 65 candidates on, where the general sequence's block lists no longer fit the one-pass merge — the two-tier
    // host's fast tier fetches 30 x 3: 10M x 256 p50 0.59 -> 0.50 ms; below that both sequences measured the same)
    if (!approx_out_dev && !exact_out_dev && cc > 64 && cc <= kSelectMaxK && k_eff <= 64 && k <= 64 && (dim_ & 7) == 0 && nrows_ >= 4096 &&
        !(row_stride_ && row_stride_ != dim_ * 2) && variant == 0) {
        if (tp_skip_ > 0) {
            --tp_skip_;
        } else {
            FSGPU_TRY(two_pass_lone_certified(query, qi.data(), qbytes, k, k_eff, cc, bits, qslab, rows.data(), scores.data(), &count, &answered));
            if (answered) {
                tp_backoff_ = 0;
            } else {
                tp_backoff_ = tp_backoff_ ? std::min<uint32_t>(tp_backoff_ * 2, 64) : 1;
                tp_skip_ = tp_backoff_;
            }
        }
    }
    if (!answered) {
    FSGPU_TRY(ws_i8_query_.reserve(qbytes));
    FSGPU_TRY(ws_queries_.reserve((size_t)dim_ * 4));
    FSGPU_TRY(ws_cand_packed_.reserve((size_t)cc * 8));
    FSGPU_TRY(ws_cand_rows_.reserve((size_t)cc * 4));
    FSGPU_TRY(ws_cand_scores_.reserve((size_t)cc * 4));
    FSGPU_TRY(ws_rows_.reserve((size_t)k * 4));
    FSGPU_TRY(ws_scores_.reserve((size_t)k * 4));
    FSGPU_TRY(ws_counts_.reserve(4));
    // both query forms go through the pinned staging block when it exists (DMA instead of pageable staging)
    const size_t qin_bytes = (((size_t)dim_ * 4 + qbytes) + 255) & ~(size_t)255;
    if (qin_bytes <= kPinnedIoBytes / 2 && pinned_io() != nullptr) {
        unsigned char* io = static_cast<unsigned char*>(io_host_);
        std::memcpy(io, query, (size_t)dim_ * 4);
        std::memcpy(io + (size_t)dim_ * 4, qi.data(), qbytes);
        FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, io, (size_t)dim_ * 4, hipMemcpyHostToDevice, stream_));
        FSGPU_HIP(hipMemcpyAsync(ws_i8_query_.ptr, io + (size_t)dim_ * 4, qbytes, hipMemcpyHostToDevice, stream_));
    } else {
        FSGPU_HIP(hipMemcpyAsync(ws_i8_query_.ptr, qi.data(), qbytes, hipMemcpyHostToDevice, stream_));
        FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, query, (size_t)dim_ * 4, hipMemcpyHostToDevice, stream_));
    }
    ScanArgs a = base_args(static_cast<const float*>(ws_queries_.ptr), nullptr);
    u64* cand_packed = static_cast<u64*>(ws_cand_packed_.ptr);
    uint32_t* cand_rows = static_cast<uint32_t*>(ws_cand_rows_.ptr);
    float* cand_scores = static_cast<float*>(ws_cand_scores_.ptr);
    // ---- pass 1: top-cc rows by the int8 dot ----
    const int kcap = cc <= 64 ? 64 : 256;
    auto launch_pass1 = [&](int grid, int* occ) {
        return bits == 8 ? launch_scan_i8(a, qslab, ws_i8_query_.ptr, kcap, grid, stream_, occ)
                         : launch_scan_4bit(a, qslab, ws_i8_query_.ptr, kcap, grid, stream_, occ);
    };
    const bool fused = bits == 8 ? scan_i8_fused_supported((int)dim_, kcap) : scan_4bit_fused_supported((int)dim_, kcap);
    if (cc <= 256 && fused) {
        int per_cu = 1;
        FSGPU_HIP(launch_pass1(1, &per_cu));
        // one block per CU for int8: the quantised rows are short, so four double-buffered waves already keep the HBM pipe full,
        // and every extra block is another candidate list for the merge and another top-k to maintain (10M x 256, 90
        // candidates: p50 0.65 -> 0.59 ms; 10M x 384, 30 candidates: 0.73 -> 0.69 ms).  FSGPU_I8_PER_CU overrides.
        // 4-bit rows are half as long again: two blocks per CU (10M x 384: 0.44 -> 0.41 ms against one, 0.44 against four).
        per_cu = std::min(per_cu, knobs().i8_per_cu > 0 ? knobs().i8_per_cu : (bits == 8 ? 1 : 2));
        int grid = num_cus_ * per_cu;
        const int max_useful = (int)(((nrows_ + 15) / 16 + 3) / 4);
        if (grid > max_useful) grid = max_useful;
        if (grid < 1) grid = 1;
        FSGPU_TRY(ws_partial_.reserve((size_t)grid * cc * 8));
        a.partial = static_cast<u64*>(ws_partial_.ptr);
        a.k = cc;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (profiling) {
            FSGPU_HIP(hipEventCreate(&e0));
            FSGPU_HIP(hipEventCreate(&e1));
            FSGPU_HIP(hipEventRecord(e0, stream_));
        }
        FSGPU_HIP(launch_pass1(grid, nullptr));
        if (profiling) {
            FSGPU_HIP(hipEventRecord(e1, stream_));
            events_.emplace_back(e0, e1);
            profiled_rows_ += nrows_;
        }
        MergeArgs m;
        m.lists = a.partial;
        m.q_stride = (uint64_t)grid * cc;
        m.l_stride = cc;
        m.nlists = (uint32_t)grid;
        m.list_len = cc;
        m.k = cc;
        m.out_stride = cc;
        m.out_rows = cand_rows;
        m.out_scores = nullptr;
        m.out_counts = nullptr;
        m.out_packed = approx_out_dev;   // (a sharded index's root wants the pass-1 entries themselves)
        FSGPU_HIP(launch_merge_topk(m, 1, stream_));
    } else {
        FSGPU_TRY(ws_keys_a_.reserve(n * 8));
        FSGPU_TRY(ws_keys_b_.reserve(n * 8));
        size_t tmp_bytes = 0;
        FSGPU_HIP(sort_keys_desc_temp_bytes(n, &tmp_bytes));
        FSGPU_TRY(ws_sort_tmp_.reserve(tmp_bytes));
        u64* keys_a = static_cast<u64*>(ws_keys_a_.ptr);
        u64* keys_b = static_cast<u64*>(ws_keys_b_.ptr);
        if (bits == 8) FSGPU_HIP(launch_score_rows_i8(a, qslab, ws_i8_query_.ptr, keys_a, stream_));
        else FSGPU_HIP(launch_score_rows_4bit(a, qslab, ws_i8_query_.ptr, keys_a, stream_));
        FSGPU_HIP(launch_packed_to_sortkey(keys_a, n, stream_));
        FSGPU_HIP(sort_keys_desc(ws_sort_tmp_.ptr, ws_sort_tmp_.bytes, keys_a, keys_b, n, stream_));
        FSGPU_HIP(launch_sorted_keys_to_rows(keys_b, cc, cand_rows, static_cast<uint32_t*>(ws_counts_.ptr), stream_, approx_out_dev));
    }
    // ---- pass 2: exact f16 rescore of the candidates, then the usual best-first selection of k ----
    FSGPU_HIP(hipMemsetAsync(cand_scores, 0, (size_t)cc * 4, stream_));
    FSGPU_HIP(launch_gather_dot(a, cand_rows, cc, cand_scores, stream_));
    FSGPU_HIP(launch_pack_hits(cand_rows, cand_scores, cc, cand_packed, stream_));
    if (exact_out_dev) FSGPU_HIP(hipMemcpyAsync(exact_out_dev, cand_packed, (size_t)cc * 8, hipMemcpyDeviceToDevice, stream_));
    MergeArgs m2;
    m2.lists = cand_packed;
    m2.q_stride = cc;
    m2.l_stride = cc;
    m2.nlists = 1;
    m2.list_len = cc;
    m2.k = k_eff;
    m2.out_stride = k;
    m2.out_rows = static_cast<uint32_t*>(ws_rows_.ptr);
    m2.out_scores = static_cast<float*>(ws_scores_.ptr);
    m2.out_counts = static_cast<uint32_t*>(ws_counts_.ptr);
    const bool pin_out = (size_t)k * 8 + 4 + qin_bytes <= kPinnedIoBytes && pinned_io() != nullptr;
    if (pin_out) {
        unsigned char* io = static_cast<unsigned char*>(io_host_) + qin_bytes;
        m2.out_rows = reinterpret_cast<uint32_t*>(io);
        m2.out_scores = reinterpret_cast<float*>(io + (size_t)k * 4);
        m2.out_counts = reinterpret_cast<uint32_t*>(io + (size_t)k * 8);
    }
    m2.out_packed = nullptr;
    m2.lists_sorted = 0;  // candidates arrive in pass-1 (int8) order
    FSGPU_HIP(launch_merge_topk(m2, 1, stream_));
    if (pin_out) {  // the last merge wrote into pinned host memory
        FSGPU_HIP(hipStreamSynchronize(stream_));
        std::memcpy(rows.data(), m2.out_rows, (size_t)k * 4);
        std::memcpy(scores.data(), m2.out_scores, (size_t)k * 4);
        count = *m2.out_counts;
    } else {
        FSGPU_HIP(hipMemcpyAsync(rows.data(), ws_rows_.ptr, (size_t)k * 4, hipMemcpyDeviceToHost, stream_));
        FSGPU_HIP(hipMemcpyAsync(scores.data(), ws_scores_.ptr, (size_t)k * 4, hipMemcpyDeviceToHost, stream_));
        FSGPU_HIP(hipMemcpyAsync(&count, ws_counts_.ptr, 4, hipMemcpyDeviceToHost, stream_));
        FSGPU_HIP(hipStreamSynchronize(stream_));
    }
    }   // (!answered)
    // resolve_hits (search.rs:1503-1558): first (best) hit per doc id when the index knows doc ids
    uint32_t outn = 0;
    for (uint32_t i = 0; i < count; ++i) {
        bool dup = false;
        if (has_doc_ids()) {
            const size_t r = (size_t)(rows[i] - row_base_);
            const char* di = doc_blob_.data() + doc_offsets_[r];
            const size_t dl = (size_t)(doc_offsets_[r + 1] - doc_offsets_[r]);
            for (uint32_t j = 0; j < outn && !dup; ++j) {
                const size_t rj = (size_t)(out_rows[j] - row_base_);
                const size_t lj = (size_t)(doc_offsets_[rj + 1] - doc_offsets_[rj]);
                dup = lj == dl && std::memcmp(doc_blob_.data() + doc_offsets_[rj], di, dl) == 0;
            }
        }
        if (dup) continue;
        out_rows[outn] = rows[i];
        out_scores[outn] = scores[i];
        ++outn;
    }
    *out_count = outn;
    return ok();
}

SearchError VectorIndex::scan_time(double* total_ms, uint64_t* launches, uint64_t* rows, bool reset) {
    FSGPU_HIP(hipSetDevice(device_));
    double sum = 0.0;
    for (auto& ev : events_) {
        FSGPU_HIP(hipEventSynchronize(ev.second));
        float ms = 0.f;
        FSGPU_HIP(hipEventElapsedTime(&ms, ev.first, ev.second));
        sum += ms;
    }
    *total_ms = sum;
    *launches = events_.size();
    if (rows) *rows = profiled_rows_;
    if (reset) {
        profiled_rows_ = 0;
        for (auto& ev : events_) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
        events_.clear();
    }
    return ok();
}

// ------------------------------------------------------------------------------------------------
// Model2VecEmbedder
// ------------------------------------------------------------------------------------------------

Model2VecEmbedder::~Model2VecEmbedder() {
    if (device_ >= 0) (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamDestroy(stream_);
    for (DeviceBuffer* b : {&table_, &ids_, &offsets_, &out_}) b->release();
}

SearchError Model2VecEmbedder::init(int device, const float* table, uint32_t vocab, uint32_t dim) {
    if (!table) return make_error(FSGPU_ERR_NULL_ARGUMENT, "table is null");
    if (dim == 0 || vocab == 0) return make_error(FSGPU_ERR_INVALID_CONFIG, "vocab and dim must be non-zero");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return make_error(FSGPU_ERR_NO_DEVICE, "no HIP device visible (libfsgpu has no CPU fallback)");
    if (device < 0 || device >= count) return make_error(FSGPU_ERR_INVALID_CONFIG, "device ordinal out of range");
    FSGPU_HIP(hipSetDevice(device));
    device_ = device;
    vocab_ = vocab;
    dim_ = dim;
    {
        // short gather kernel vs the scans' chip-filling launches on other streams: highest priority (see bert_embedder.cpp)
        int least = 0, greatest = 0;
        FSGPU_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        FSGPU_HIP(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, greatest));
    }
    FSGPU_TRY(table_.reserve((size_t)vocab * dim * 4));
    FSGPU_HIP(hipMemcpy(table_.ptr, table, (size_t)vocab * dim * 4, hipMemcpyHostToDevice));
    return ok();
}

SearchError Model2VecEmbedder::embed_batch(const uint32_t* ids, const uint32_t* offsets, uint32_t n, float* out, float* out_dev) {
    if (n == 0) return ok();
    if (!offsets || (!out && !out_dev)) return make_error(FSGPU_ERR_NULL_ARGUMENT, "offsets/out is null");
    for (uint32_t i = 0; i < n; ++i)
        if (offsets[i + 1] < offsets[i]) return make_error(FSGPU_ERR_INVALID_CONFIG, "offsets must be non-decreasing");
    const uint32_t total = offsets[n];
    // the reference embedder is immutable and lock-free (model2vec_embedder.rs:55-58); here the staging buffers and the
    // stream are per handle, so concurrent callers take turns (fsgpu.h: calls on one handle serialise internally)
    std::lock_guard<std::mutex> lock(mu_);
    FSGPU_HIP(hipSetDevice(device_));
    FSGPU_TRY(ids_.reserve((size_t)(total ? total : 1) * 4));
    FSGPU_TRY(offsets_.reserve((size_t)(n + 1) * 4));
    FSGPU_TRY(out_.reserve((size_t)n * dim_ * 4));
    if (total) FSGPU_HIP(hipMemcpyAsync(ids_.ptr, ids, (size_t)total * 4, hipMemcpyHostToDevice, stream_));
    FSGPU_HIP(hipMemcpyAsync(offsets_.ptr, offsets, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, stream_));
    FSGPU_HIP(launch_m2v_embed(static_cast<const float*>(table_.ptr), vocab_, dim_,
                               static_cast<const uint32_t*>(ids_.ptr), static_cast<const uint32_t*>(offsets_.ptr), n,
                               out_dev ? out_dev : static_cast<float*>(out_.ptr), stream_));   // (device output: the vectors stay in HBM)
    if (out) FSGPU_HIP(hipMemcpyAsync(out, out_dev ? out_dev : out_.ptr, (size_t)n * dim_ * 4, hipMemcpyDeviceToHost, stream_));
    FSGPU_HIP(hipStreamSynchronize(stream_));
    return ok();
}

}  // namespace fsgpu

namespace fsgpu {

SearchError VectorIndex::search_top_k_packed_device(const float* queries_dev, uint32_t nq, uint32_t query_len,
                                                    uint32_t k, const uint64_t* allow_dev, uint64_t* out_packed_dev,
                                                    hipStream_t stream) {
    if (query_len != dim_) {
        SearchError e;
        e.code = FSGPU_ERR_DIMENSION_MISMATCH;
        e.detail = "expected " + std::to_string(dim_) + ", found " + std::to_string(query_len);
        return e;
    }
    if (nq == 0 || k == 0) return SearchError{};
    if (dim_ % 8 != 0 || k > 256 || f32_) {
        SearchError e;
        e.code = FSGPU_ERR_INVALID_CONFIG;
        e.detail = "packed shard search supports F16 slabs, k <= 256 and dim % 8 == 0";
        return e;
    }
    if (hipSetDevice(device_) != hipSuccess) {
        SearchError e;
        e.code = FSGPU_ERR_DEVICE;
        e.detail = "hipSetDevice failed";
        return e;
    }
    if (nrows_ == 0) {
        if (hipMemsetAsync(out_packed_dev, 0xff, (size_t)nq * k * 8, stream) != hipSuccess) {
            SearchError e;
            e.code = FSGPU_ERR_DEVICE;
            e.detail = "hipMemsetAsync failed";
            return e;
        }
        return SearchError{};
    }
    const uint32_t k_eff = (uint64_t)k < nrows_ ? k : (uint32_t)nrows_;
    return fused_search(queries_dev, nq, k, k_eff, allow_dev, nullptr, nullptr, nullptr,
                        reinterpret_cast<u64*>(out_packed_dev), stream);
}

// Cross-shard merge of packed best-first lists (the step after the RCCL all-gather, SURVEY §8e;
// same selection rule as merge_partial_heaps, search.rs:1704-1720).
SearchError merge_packed_lists_device(int device, const uint64_t* lists_dev, uint32_t nq, uint32_t nlists,
                                      uint32_t list_len, uint64_t q_stride, uint64_t l_stride, uint32_t k,
                                      uint32_t* out_rows_dev, float* out_scores_dev, uint32_t* out_counts_dev,
                                      hipStream_t stream) {
    SearchError e;
    if (nq == 0) return e;
    if (hipSetDevice(device) != hipSuccess) {
        e.code = FSGPU_ERR_DEVICE;
        e.detail = "hipSetDevice failed";
        return e;
    }
    MergeArgs m;
    m.lists = reinterpret_cast<const u64*>(lists_dev);
    m.q_stride = q_stride;
    m.l_stride = l_stride;
    m.nlists = nlists;
    m.list_len = list_len;
    m.k = k;
    m.out_stride = k;
    m.out_rows = out_rows_dev;
    m.out_scores = out_scores_dev;
    m.out_counts = out_counts_dev;
    m.out_packed = nullptr;
    hipError_t he = launch_merge_topk(m, (int)nq, stream);
    if (he != hipSuccess) {
        e.code = FSGPU_ERR_DEVICE;
        e.detail = hipGetErrorString(he);
    }
    return e;
}

}  // namespace fsgpu
