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
#include "vector_index_internal.hpp"

namespace fsgpu {

using namespace detail;

namespace {

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
                            &mf_fallback_, &mf_fallback2_, &mf_spill_, &mf_io_, &mf_io2_, &ws_out_, &i8_stats_, &n4u_slab_, &mf_cand_count_, &ws_pairs_,
                            &i8f_slab_, &i8f_max_, &i8f_stats_, &rot_mat_, &rot_q_})
        b->release();
    if (mf_flags_host_) (void)hipHostFree(mf_flags_host_);
    if (io_host_) (void)hipHostFree(io_host_);
    if (batch_io_host_) (void)hipHostFree(batch_io_host_);
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

// A pinned block for the results of batched searches at the host-pointer ABI (grows to what a call needs, up to 32 MB).
void* VectorIndex::pinned_batch_io(size_t bytes) {
    constexpr size_t kMax = 32u << 20;
    if (bytes > kMax || batch_io_failed_) return nullptr;
    if (bytes > batch_io_bytes_) {
        if (batch_io_host_) (void)hipHostFree(batch_io_host_);
        batch_io_host_ = nullptr;
        batch_io_bytes_ = 0;
        size_t want = 1u << 20;
        while (want < bytes) want <<= 1;
        if (hipHostMalloc(&batch_io_host_, want, hipHostMallocDefault) != hipSuccess) {
            batch_io_host_ = nullptr;
            batch_io_failed_ = true;
            (void)hipGetLastError();
            return nullptr;
        }
        batch_io_bytes_ = want;
    }
    return batch_io_host_;
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
        int pass = left >= 2 && !f32_ ? 2 : 1;
        if (f32_) pass = scan_f32_queries_per_pass((int)dim_, kcap, (int)std::min<uint32_t>(left, 4));   // F32 slabs: 4 / 2 / 1 queries per pass
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
            FSGPU_HIP(launch_scan_topk_f32(probe, kcap, pass, 1, stream, &per_cu));
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
        if (f32_) FSGPU_HIP(launch_scan_topk_f32(a, kcap, pass, grid, stream, nullptr));
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
        FSGPU_HIP(sort_keys_desc(ws_sort_tmp_.ptr, ws_sort_tmp_.bytes, keys_a, keys_b, n, stream, sortkey_varying_bits(live_dev_ || allow_dev)));
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
            FSGPU_HIP(sort_keys_desc(ws_sort_tmp_.ptr, ws_sort_tmp_.bytes, packed, keys_b, n, stream, sortkey_varying_bits(false)));   // (allowed, live rows only)
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

SearchError VectorIndex::gather_dot_batched(const float* queries, uint32_t nq, uint32_t query_len, const uint32_t* rows, const uint32_t* qidx,
                                            uint32_t n, float* out) {
    FSGPU_TRY(ensure_query_dimension(query_len));
    if (n == 0) return ok();
    for (uint32_t i = 0; i < n; ++i) {
        if (rows[i] < row_base_ || rows[i] - row_base_ >= nrows_)
            return make_error(FSGPU_ERR_INVALID_CONFIG, "row index out of range for dot_query_at");
        if (qidx[i] >= nq) return make_error(FSGPU_ERR_INVALID_CONFIG, "query index out of range");
    }
    if (f32_ || (dim_ & 7) != 0) {   // shapes the multi-query kernel does not cover: query by query (items of a query are consecutive or not)
        std::vector<uint32_t> r1;
        std::vector<float> o1;
        for (uint32_t q = 0; q < nq; ++q) {
            r1.clear();
            for (uint32_t i = 0; i < n; ++i)
                if (qidx[i] == q) r1.push_back(rows[i]);
            if (r1.empty()) continue;
            o1.resize(r1.size());
            FSGPU_TRY(gather_dot(queries + (size_t)q * dim_, query_len, r1.data(), (uint32_t)r1.size(), o1.data()));
            size_t j = 0;
            for (uint32_t i = 0; i < n; ++i)
                if (qidx[i] == q) out[i] = o1[j++];
        }
        return ok();
    }
    FSGPU_HIP(hipSetDevice(device_));
    const size_t qbytes = (size_t)nq * dim_ * 4;
    FSGPU_TRY(ws_queries_.reserve(qbytes));
    FSGPU_TRY(ws_gather_rows_.reserve((size_t)n * 8));
    FSGPU_TRY(ws_gather_out_.reserve((size_t)n * 4));
    uint32_t* rows_dev = static_cast<uint32_t*>(ws_gather_rows_.ptr);
    FSGPU_HIP(hipMemcpyAsync(ws_queries_.ptr, queries, qbytes, hipMemcpyHostToDevice, stream_));
    FSGPU_HIP(hipMemcpyAsync(rows_dev, rows, (size_t)n * 4, hipMemcpyHostToDevice, stream_));
    FSGPU_HIP(hipMemcpyAsync(rows_dev + n, qidx, (size_t)n * 4, hipMemcpyHostToDevice, stream_));
    ScanArgs a = base_args(static_cast<const float*>(ws_queries_.ptr), nullptr);
    FSGPU_HIP(launch_gather_dot_mq(a, rows_dev, rows_dev + n, n, static_cast<float*>(ws_gather_out_.ptr), stream_));
    FSGPU_HIP(hipMemcpyAsync(out, ws_gather_out_.ptr, (size_t)n * 4, hipMemcpyDeviceToHost, stream_));
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