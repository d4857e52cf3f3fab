// vector_index_internal.hpp — what the translation units of VectorIndex share: status helpers, the HIP / status early-return
// macros and the switches read from the environment.  Not installed, not part of the C ABI (include/fsgpu.h is).
//   vector_index.cpp          lifecycle, FSVI image, WAL, tombstones, the exact scan paths, MRL views, packed lists
//   vector_index_batched.cpp  the batched (matrix-core) search: plan, sample, main pass, selections, fallback, tickets, and the
//                             int8 filter's copy of the slab
//   vector_index_lone.cpp     one query at a time: the certified int8 pass, the exact halves, the quantised two-pass lanes
#pragma once

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>

#include "../../include/fsgpu.h"
#include "vector_index.hpp"

namespace fsgpu {
namespace detail {

inline SearchError ok() { return SearchError{}; }

inline SearchError hip_fail(hipError_t e, const char* what) {
    SearchError err;
    err.code = FSGPU_ERR_DEVICE;
    err.detail = std::string(what) + ": " + hipGetErrorString(e);
    return err;
}

inline SearchError make_error(int32_t code, std::string detail) {
    SearchError e;
    e.code = code;
    e.detail = std::move(detail);
    return e;
}

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
    bool narrow_i8f = false;   // FSGPU_NARROW_I8F: batches of up to 64 queries of the int8-filtered exact search on the 64-query shape (rounds 2-5)
    int rb_pct = 0;      // FSGPU_RB_PCT: the second sample's size in percent of what the plan chose (tuning experiments only)
    int heur_rank = 0;   // FSGPU_HEUR_RANK: rank of the first sample whose score gates the anchoring-only second sample (default 4)
    int wide_min = 0;    // FSGPU_WIDE_MIN: fewest queries left that take the register-resident-query main pass (default 129)
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
        wide_min = num("FSGPU_WIDE_MIN");
        slots_b = std::min(num("FSGPU_SLOTS_B"), (int)kWideSlots);
        slots_main = std::min(num("FSGPU_SLOTS_MAIN"), (int)kWideSlots);
        no_skip_b = env("FSGPU_NO_SKIP_B") != nullptr;
        no_wide_b = env("FSGPU_NO_WIDE_B") != nullptr;
        no_anchor = env("FSGPU_NO_ANCHOR") != nullptr;
        no_big_pool = env("FSGPU_NO_BIG_POOL") != nullptr;
        no_heur_b = env("FSGPU_NO_HEUR_B") != nullptr;
        no_group_sample = env("FSGPU_NO_GROUP_SAMPLE") != nullptr;
        narrow_i8f = env("FSGPU_NARROW_I8F") != nullptr;
        heur_rank = num("FSGPU_HEUR_RANK");
        no_reverse = env("FSGPU_NO_REVERSE") != nullptr;
        use_160 = env("FSGPU_USE_160") != nullptr;
#endif
    }
};
inline const Knobs& knobs() {
    static const Knobs k;
    return k;
}
// Fewest queries that ride the register-resident-query main pass: one more than the LDS-query kernel answers in ONE pass over the slab
// (the tail of a 256-slot launch is padding — batched_round_setup).
// The int8 TWO-PASS (its own pass-1 scores, no anchored threshold) crosses over earlier: 65..128 queries cost 0.86-0.89 ms on the 128-slot
// LDS-query shape and 0.82 ms as a padded 256-slot pass at 10M x 256 (profiles/r06/lds_query_shape_ab.txt).
inline uint32_t wide_min_queries(bool two_pass = false) {
    if (knobs().wide_min > 64) return (uint32_t)knobs().wide_min;
    return two_pass ? 65u : 129u;
}

}  // namespace detail
}  // namespace fsgpu

#define FSGPU_HIP(expr)                                                   \
    do {                                                                  \
        hipError_t _e = (expr);                                           \
        if (_e != hipSuccess) return ::fsgpu::detail::hip_fail(_e, #expr); \
    } while (0)

#define FSGPU_TRY(expr)            \
    do {                           \
        SearchError _s = (expr);   \
        if (!_s.ok()) return _s;   \
    } while (0)
