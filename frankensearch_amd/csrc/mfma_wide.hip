// mfma_wide.hip — the main pass of the batched scan at 256 / 384 queries per pass over the slab.
//
// scan_mfma_kernel (mfma_scan.hip) keeps the f16 queries of a group in LDS (800 bytes each): 128 queries fill 100 KB, so
// one pass over the slab serves 128 queries and the step is HBM-bound with the matrix pipe mostly idle.  Here the roles
// are swapped:
//   * the QUERIES live in registers: each of the block's 8 waves owns QT query tiles of 16 (QT = 2: 32 queries, 256 per
//     block) and keeps their B fragments for the whole dimension resident (QT x dim/32 x 4 VGPRs = 96 at dim 384);
//   * the ROWS go through LDS: a ring of row tiles is filled by LDS-DMA (global_load_lds_dwordx4: no staging registers,
//     no ds_write pass, no ds_bpermute transpose) in the coalesced quad layout of the exact kernels (four consecutive
//     lanes fetch 64 contiguous bytes), several tiles ahead of the tile being consumed; every wave reads each row tile's
//     A fragments with ds_read_b128 and multiplies them with its own queries.
// One s_barrier per tile; the DMA queue is never drained inside the loop (counted s_waitcnt vmcnt).  The DMA and its
// waits are inline asm: hipcc would otherwise wait for the newest LDS-DMA before every ds_read (it cannot tell the ring
// slots apart), which serialises load and compute.
//
// LDS image of one DMA instruction (1 KB = 16 rows x 64 bytes = one MFMA k-step of a 16-row sub-tile): the DMA writes
// lane-linear (lane L at L*16), lane L = 4p + c fetches the 16-byte chunk c ^ f(p) of row p, f(p) = (4 - (p >> 2)) & 3.
// A fragment lane (row i, k-group g) then reads position i*64 + ((g ^ f(i)) * 16): with this swizzle each of the four
// 16-lane groups ds_read_b128 is served in ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...; MI355X_MICROARCH.md, LDS)
// covers all 64 banks exactly once.
//
// Arithmetic, candidate filter and outputs are those of scan_mfma_kernel's main pass (approximate f16-query scores or
// exact int8 scores, rows at or above tau kept in per-(query, block) lists), so the selection / exact re-score that
// follows is unchanged and the final hits are still bit-identical to the reference order.  The live / allow bitmaps
// are consulted only for the few rows that pass the threshold, and those rows are appended straight to the block's
// global list (LDS keeps one counter per query), which leaves the LDS to the ring.
#include <atomic>
#include <cstdlib>
#include <string>
#include <vector>
#include <type_traits>

#include "scan_common.hpp"

namespace fsgpu {

using namespace scan_detail;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace {

// One LDS-DMA: every lane's 16 bytes at gsrc land at lds_dst + lane * 16 (lds_dst wave-uniform).  Invisible to hipcc's
// s_waitcnt bookkeeping: counted by hand below (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

// N LDS-DMAs of one wave behind ONE scalar base: lane offsets in VGPRs (constant for the whole kernel), destinations 1 KB
// apart from lds_dst on.  m0 is saved / restored once per group; no vector address arithmetic per instruction.
// a wave-uniform 64-bit value into scalar registers whatever instructions computed it (an "s" operand of inline asm is not
// moved there by the compiler: a value it kept in vector registers fails to assemble — seen when the tile loop was restructured)
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

template <int N>
__device__ __forceinline__ void glds16_group(uint64_t sbase_any, const uint32_t (&voff)[N], uint32_t lds_dst) {
    static_assert(N >= 1 && N <= 3, "a wave issues 1..3 DMA instructions per tile");
    const uint64_t sbase = uniform_u64(sbase_any);
    unsigned keep;
    // (one asm statement: hipcc must not get to place anything that reads m0 between the pieces)
    if constexpr (N == 1)
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %3, %2\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep) : "s"(lds_dst), "s"(sbase), "v"(voff[0]) : "memory");
    else if constexpr (N == 2)
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %3, %2\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %4, %2\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep) : "s"(lds_dst), "s"(sbase), "v"(voff[0]), "v"(voff[1]) : "memory", "scc");
    else
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %3, %2\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %4, %2\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %5, %2\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep) : "s"(lds_dst), "s"(sbase), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]) : "memory", "scc");
}

// The smallest integer c with (s >= c) == ((float)s >= t) for every integer |s| <= 2^24 (scores of the int8 rows this
// kernel takes: at most 768 x 127 x 127): ceil(t), clamped so that s - c cannot overflow; NaN / +inf: nothing passes.
__device__ __forceinline__ int ceil_threshold(float t) {
    if (!(t <= 33554432.0f)) return 1 << 25;
    if (t < -33554432.0f) return -(1 << 25);
    return (int)ceilf(t);
}

// a wave vote: true for every lane when the predicate holds in any lane
__device__ __forceinline__ bool wave_any(bool x) { return __ballot(x) != 0; }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// the arrive / wait counters of kOptFlags (LDS byte addresses; hipcc's s_waitcnt bookkeeping does not see these, so each waits itself)
[[maybe_unused]] __device__ __forceinline__ void flag_post(uint32_t lds_addr, int lane) {
    asm volatile("" ::: "memory");   // the wave's earlier LDS reads stay in front of the post (the LDS serves a wave's operations in order)
    if (lane == 0) {
        const uint32_t one = 1;
        asm volatile("ds_add_u32 %0, %1" ::"v"(lds_addr), "v"(one) : "memory");
    }
}
[[maybe_unused]] __device__ __forceinline__ void flag_wait(uint32_t lds_addr, uint32_t target) {
    for (;;) {
        uint32_t v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr) : "memory");
        if ((uint32_t)__builtin_amdgcn_readfirstlane(v) >= target) break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}

}  // namespace

// ROWB = bytes per slab row (dim * 2 for f16 rows, dim for int8 rows), EB as in scan_mfma_kernel, QT = query tiles of 16
// per wave (8 waves: 128 * QT queries per pass), NSLOT = ring slots.
//
// Schedule of one tile (NC chunks of CK k-steps x 2 sub-tiles; fragments double-buffered in registers):
//     [entry: chunk 0's fragment reads are already in flight — issued during the previous tile]
//     reads(chunk 1) | MFMA(chunk 0) | s_waitcnt vmcnt + s_barrier: tile n+1 has landed for everyone and tile n-1's slot is
//     free -> DMA(tile n+NSLOT-1) | reads(chunk 2) | MFMA(chunk 1) | ... | reads(chunk 0 of tile n+1) | MFMA(last chunk)
// so neither the LDS latency nor the barrier sits between two MFMA groups of a wave with nothing else to issue.
// DBG: 0 = the main pass; 3 = a sample stage (its own instantiation, so that profiles tell the two apart and the main pass
// carries no trace of the sample mapping); 1 = no MFMAs, 2 = no DMA (timing experiments only: -DFSGPU_EXPERIMENTS)
//
// OPT (bit set):
//   kOptNegTau int8 rows: the accumulators start at -ceil(tau) instead of 0, so "some score of the pair reaches its query's
//              threshold" is ONE sign test on the maximum of the lane's 8 x QT accumulators (v_max3_i32) instead of QT
//              maxima, conversions and compares.  (s >= ceil(tau)) == ((float)s >= tau) for these integer scores.
//   kOptSaddr  DMA addresses = one scalar tile base + per-lane offsets that are constant for the whole kernel
//              (global_load_lds_dwordx4 v, s[..]); only a tile that crosses the end of the slab computes clamped addresses.
//   kOptSplit  (needs kOptNegTau; row tiles of two sub-tile pairs or more) the loop is organised by QUERY-TILE halves instead of
//              k-step chunks: a pair's fragments for ALL its k-steps sit in registers (2 x KS x 4 VGPRs, what the chunk
//              double buffer took), phase 1 multiplies them with the first half of the wave's query tiles, phase 2 with the
//              second half — and as phase 2 retires a k-step's fragments, the next pair's fragments of that k-step are read
//              into the same registers (a rolling prefetch).  The threshold test of a finished half then lies in the shadow
//              of the OTHER half's MFMAs (phase-1 scores are tested during phase 2, phase-2 scores during the next pair's
//              phase 1), the LDS reads are spread one pair per four MFMAs, and nothing but the barrier interrupts the matrix
//              stream of a wave.
//   kOptBig    (with kOptSplit, main pass only) row tiles of 128 rows in a ring of THREE slots (the same 144 KB): half as many
//              barriers, tile cursors and DMA bookkeeping per row; tile n is consumed while n+1 is landing and n-1's slot takes
//              n+2 — one tile of lookahead is 4-5 us of matrix work, well past the HBM latency.
//   kOptFlags  (with kOptBig) no block-wide s_barrier in the tile loop: two LDS counters per ring slot — `landed` (waves whose share
//              of the tile's DMA is in LDS) and `freed` (waves past their last read of the tile) — are posted EARLY and waited for
//              LATE.  A wave posts landed(n+1) at the start of tile n (its share went out a whole tile earlier) and needs
//              everyone's only before it reads tile n+1's first fragments at the END of tile n; it posts freed(n) one sub-tile pair
//              before the end of tile n and the refill of that slot waits for everyone's a few MFMAs into tile n+1.  The eight
//              waves therefore drift up to about a tile apart instead of meeting at every tile: a wave held up in the append path
//              (a scalar bitmap load, an LDS atomic, a store — ~5 events per tile at 10M rows, ~40 on a 1.25M-row shard) no longer
//              stops the other seven, and the two waves of a SIMD no longer stop together.
constexpr int kOptNegTau = 2, kOptSaddr = 4, kOptSplit = 8, kOptBig = 16, kOptFlags = 32;

// k-steps per chunk of the chunk loop (fewer where the resident queries leave fewer registers for the fragment double buffer)
constexpr int wide_chunk_ksteps(int KS, int QT) {
    return KS == 12 ? (QT >= 3 ? 2 : 3) : KS == 6 ? (QT >= 5 ? 1 : 3) : KS == 4 ? (QT >= 7 ? 1 : 2) : (KS >= 2 ? KS / 2 : 1);
}
// shapes whose resident queries already fill the register file take no option that costs registers
constexpr bool wide_room(int ROWB, int QT) {
    const int KS = ROWB / 64;
    return QT * KS * 4 + 4 * wide_chunk_ksteps(KS, QT) * 4 + 8 * QT <= 190;
}
// the query-tile-split loop: int8 rows with neg-tau, tiles of two pairs or more, and registers for a whole pair's fragments
constexpr bool wide_split_ok(int ROWB, int EB, int QT, int OPT, int DBG) {
    const int KS = ROWB / 64;
    return (OPT & kOptSplit) != 0 && (OPT & kOptNegTau) != 0 && EB == 1 && wide_room(ROWB, QT) && ROWB < 512 && QT >= 2 && DBG != 1 &&
           QT * KS * 4 + 2 * KS * 4 + 9 * QT <= 200;
}
// 128-row tiles in a three-slot ring: the main pass of a shape that runs the split loop
constexpr bool wide_big_ok(int ROWB, int EB, int QT, int NSLOT, int OPT, int DBG) {
    return (OPT & kOptBig) != 0 && wide_split_ok(ROWB, EB, QT, OPT, DBG) && (DBG == 0 || DBG == 8 || DBG >= 16) && NSLOT == 3;
}   // (bit 1 was a barrier-phase shift between the SIMD twins: measured null, removed)
// one 64-bit word through the scalar cache (wave-uniform address): counted by lgkmcnt, not by the vmcnt the DMA ring lives on
__device__ __forceinline__ u64 sload_u64(const u64* pv) {
    const u64* p = reinterpret_cast<const u64*>((uintptr_t)uniform_u64((uint64_t)(uintptr_t)pv));
    u64 w;
#ifdef FSGPU_LAB_SLOAD_NOGLC
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(w) : "s"(p) : "memory");
#elif defined(FSGPU_LAB_SLOAD_SINGLE)   // (lab: one read, as rounds 3 and 4 shipped it)
    asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(w) : "s"(p) : "memory");   // (glc: past the scalar cache)
#else
    // Read TWICE (both past the scalar cache), behind eight wait states after the address was written, and taken only when the two
    // reads agree.  Round 4 traced one wrong answer in ~16,000 filtered batches on ONE box of three to a transient wrong return of this
    // hand-written load (profiles/r04/bitmap_soak_*.txt: never with vector loads, never with the per-wave s_dcache_inv, never on the
    // other boxes; the scalar cache itself is coherent: scache_repro.txt) without finding its cause.  The path is the rare excursion of
    // a passing score: a second 8-byte load and a compare cost nothing measurable, and a value that comes back wrong once does not come
    // back wrong twice the same way.
    u64 w2;
    int tries = 0;
    do {
        asm volatile("s_nop 7\n\ts_load_dwordx2 %0, %2, 0x0 glc\n\ts_load_dwordx2 %1, %2, 0x0 glc\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(w), "=&s"(w2)
                     : "s"(p)
                     : "memory");
    } while (w != w2 && ++tries < 8);
    if (w != w2) {
        // Eight disagreeing pairs: the scalar path is not to be believed for this word.  The word is fetched ONCE more through the
        // vector path (system-coherent, past every cache) — the path round 4's soak never saw return a wrong value.  vmcnt(0) also
        // drains this wave's DMA ring, which is always safe (the counted waits that follow find fewer loads outstanding, never more);
        // never taken in any soak, so its cost is a cold branch.
        u64 wv;
        asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(wv) : "v"(p) : "memory");
        w = uniform_u64(wv);
    }
#endif
    return w;
}
#ifdef FSGPU_LAB_BITMAP_CHECK
// lab: every bitmap word of the append path is fetched through the scalar path AND through the vector path; a disagreement is
// recorded (up to 64 records of 8 words: block, wave, word index, which bitmap, scalar value, vector value, scalar value read
// again, vector value read again) — scripts/r04/bitmap_soak.py prints them through fsgpu_lab_bitmap_debug
__device__ unsigned long long g_bm_dbg[64 * 8];
__device__ unsigned int g_bm_dbg_n;
__device__ __forceinline__ u64 vload_u64(const u64* p) {
    u64 w;
    asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(w) : "v"(p) : "memory");
    return w;
}
__device__ __noinline__ void bm_check(const u64* pv, u64 sv, uint32_t wi, int which, int wave) {
    const uint64_t pa = (uint64_t)(uintptr_t)pv;   // (a function argument arrives in vector registers: back to a scalar pointer)
    const u64* p = reinterpret_cast<const u64*>((uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pa >> 32)) << 32) |
                                                            (uint32_t)__builtin_amdgcn_readfirstlane((int)pa)));
    const u64 vv = vload_u64(p);
    if (vv == sv) return;
    if ((threadIdx.x & 63) != (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x & 63)) return;   // first active lane only
    const unsigned n = atomicAdd(&g_bm_dbg_n, 1u);
    if (n >= 64) return;
    const u64 sv2 = sload_u64(p), vv2 = vload_u64(p);
    unsigned long long* r = g_bm_dbg + n * 8;
    r[0] = blockIdx.x;
    r[1] = (unsigned)wave;
    r[2] = wi;
    r[3] = (unsigned long long)which | ((unsigned long long)(uintptr_t)p << 8);
    r[4] = sv;
    r[5] = vv;
    r[6] = sv2;
    r[7] = vv2;
}
#endif

// The kernel's body for ONE wave: QT query tiles starting at the block's query q0_wave; NQB = the block's queries (what its waves hold
// together).  Every wave of a block runs the same tile sequence, DMA share and barriers whatever its QT (scan_wide_asym_kernel).
template <int ROWB, int EB, int QT, int NSLOT, int OPT, int DBG, int WPBT, int NQB>
__device__ __forceinline__ void scan_wide_body(MfmaScanArgs args, const int q0_wave) {
    using acc_t = std::conditional_t<EB == 2, f32x4, i32x4>;
    constexpr int WPB = WPBT, NT = WPB * 64;   // (16 waves of two query tiles — four per SIMD, 128 registers each — spilled in the loop: 4.23 against 2.64 ms)
    constexpr int KS = ROWB / 64;                                   // MFMA k-steps (64 bytes of a row each)
    constexpr bool BIG = wide_big_ok(ROWB, EB, QT, NSLOT, OPT, DBG);
    constexpr bool FLAGS = BIG && (OPT & kOptFlags) != 0;
    constexpr int TR = BIG ? 128 : ROWB >= 512 ? 32 : 64;           // rows per tile
    constexpr int RS = TR / 16;                                     // 16-row sub-tiles per tile
    constexpr int NI = RS * KS;                                     // DMA instructions per tile (1 KB each)
    constexpr int PW = NI / WPB;                                    // ... per wave
    static_assert(NI % WPB == 0 && PW >= 1, "a tile's DMA instructions divide evenly over the 8 waves");
    static_assert(RS % 2 == 0, "sub-tiles are consumed in pairs");
    constexpr int CK = wide_chunk_ksteps(KS, QT);
    constexpr int NCH = KS / CK;                                    // chunks per sub-tile pair
    constexpr int NC = (RS / 2) * NCH;                              // chunks per tile
    static_assert(KS % CK == 0 && NC % 2 == 0, "an even number of chunks per tile: the register double buffer starts every tile in the same phase");
    static_assert(NSLOT >= 3, "tile n is consumed while n+1 .. n+NSLOT-2 are in flight and n-1's slot is being refilled");
    constexpr bool ROOM = wide_room(ROWB, QT);
    constexpr bool NEGTAU = (OPT & kOptNegTau) != 0 && EB == 1 && ROOM;
    constexpr bool SADDR = (OPT & kOptSaddr) != 0 && ROOM;
    constexpr bool SPLIT = wide_split_ok(ROWB, EB, QT, OPT, DBG);
    // Timing skeletons of the split loop (experiments builds, scripts/r04/skeletons.sh; their answers are not valid).  DBG in [16, 32):
    // a bit set on 16 — 1 no DMA, 2 no per-tile wait + barrier, 4 no fragment reads in the loop, 8 no threshold tests; the append path
    // is one LDS store.  DBG >= 32: everything real but the append path, a bit set on 32 — 1 no global store, 2 no LDS counter (slot 0),
    // 4 an empty append (the per-lane tests still run).
    // DBG 7: a SAMPLE stage that appends nothing — every lane keeps, per query tile, the best score it has seen and the sub-tile pair it
    // saw it in (the lane's 8 rows of that pair: "a group"); the block writes its 4 groups per query ([q][block][4] in args.cand, kEmpty
    // where a lane saw nothing) and select_groups_kernel (mfma_scan.hip) re-scores the best groups' rows exactly.  The groups of one
    // query are disjoint row sets, so any k re-scored rows bound the k-th best score from below: no thresholds in, no lists, no
    // divergent path in the loop.
    constexpr bool GMAX = DBG == 7 && SPLIT && EB == 1;   // (built on the split loop over int8 rows: scan_wide_group_maxima_supported)
    constexpr bool SK = DBG >= 16 && DBG < 32;
    constexpr bool SK_NO_DMA = DBG == 2 || (SK && (DBG & 1)), SK_NO_SYNC = SK && (DBG & 2), SK_NO_READS = SK && (DBG & 4), SK_NO_TEST = SK && (DBG & 8);
    constexpr bool AK = DBG >= 32;
    [[maybe_unused]] constexpr bool AK_NO_STORE = AK && (DBG & 1), AK_NO_ATOMIC = AK && (DBG & 2), AK_NO_BODY = AK && (DBG & 4);
    // DBG 8 (experiments builds, scripts/r04/wide_stamps.sh): the real kernel with s_memtime stamps — per wave the cycles of the whole
    // tile loop, the cycles between arriving at a tile's wait + barrier and leaving it, the cycles inside the append path and its
    // entries; written to args.dense as [block][wave][4] 64-bit words.  Answers stay valid; the stamps cost a few percent.
    constexpr bool STAMPS = DBG == 8;
    [[maybe_unused]] unsigned long long st_loop = 0, st_bar = 0;
    static_assert(!SPLIT || (NEGTAU && RS >= 4), "the split loop runs on neg-tau accumulators over tiles of two pairs or more");
    // -ceil(tau) kept in all four registers of an accumulator (the first MFMA of a pair takes it as C: no initialising moves) where
    // 4 x QT more registers fit; else one register per query tile and four moves per accumulator (in the MFMAs' shadow when SPLIT)
    constexpr bool NT4 = NEGTAU && !SPLIT && QT * KS * 4 + 4 * CK * 4 + 12 * QT <= 200 && QT * KS * 4 <= 96;
    constexpr int TILE_BYTES = TR * ROWB;
    constexpr int NQ = NQB;
    if (gridDim.y > 1) {
        // several query groups in ONE launch (args.groups = gridDim.y): group g's queries / thresholds / lists / spill area follow group
        // g - 1's, its blocks take over the CUs as the previous group's blocks leave them (one block per CU is resident) — no launch
        // ramp and no chip-wide tail between two passes over the slab; consecutive groups walk the slab in opposite directions
        const uint32_t grp = blockIdx.y;
        args.queries = static_cast<const unsigned char*>(args.queries) + (size_t)grp * NQ * ROWB;
        args.tau += (size_t)grp * NQ;
        args.cand += (size_t)grp * NQ * gridDim.x * args.slots;
        if (args.cand_count) args.cand_count += (size_t)grp * NQ * gridDim.x;
        args.spill += (size_t)grp * NQ * args.spill_cap;
        args.spill_count += (size_t)grp * NQ * kMfmaSpillCountStride;
        args.overflow += (size_t)grp * NQ;
        args.reverse ^= grp & 1u;
    }
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const uint32_t ring = (uint32_t)(uintptr_t)smem;                // LDS byte address of the ring (low half of the flat address)
    int* lcnt = reinterpret_cast<int*>(smem + (size_t)NSLOT * TILE_BYTES);   // entries appended per query by this block
    // kOptFlags: landed[slot] at flags + 4 slot, freed[slot] at flags + 16 + 4 slot (counts of waves, monotonic: 8 per use of the slot)
    const uint32_t flags = ring + (uint32_t)(NSLOT * TILE_BYTES + NQ * 4);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fk = lane >> 4;
    const int slots = (int)args.slots;
    for (int i = tid; i < NQ; i += NT) lcnt[i] = 0;
    if (FLAGS && tid < 8) lcnt[NQ + tid] = 0;
    if (STAMPS && tid < 32) lcnt[NQ + 16 + tid] = 0;

    // this wave's queries: B fragments for the whole dimension, resident in registers
    const int q0 = q0_wave;
    half8 bq[QT][KS];
    float tau[QT];
    int ctau[QT];       // NEGTAU: ceil(tau) as an integer ...
    i32x4 ntau4[QT];    // ... and its negative in all four accumulator registers of a (sub-tile, query tile)
    {
        const unsigned char* qbase = static_cast<const unsigned char*>(args.queries);
#pragma unroll
        for (int nt = 0; nt < QT; ++nt) {
            const unsigned char* qp = qbase + (size_t)(q0 + nt * 16 + frow) * ROWB + fk * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) bq[nt][ks] = *reinterpret_cast<const half8*>(qp + ks * 64);
            tau[nt] = GMAX ? 0.f : args.tau[q0 + nt * 16 + frow];   // (GMAX: the accumulators hold the scores themselves)
            ctau[nt] = ceil_threshold(tau[nt]);
            if constexpr (NT4) ntau4[nt] = i32x4{-ctau[nt], -ctau[nt], -ctau[nt], -ctau[nt]};
        }
        // the fragments must have ARRIVED before the first DMA is issued: hipcc places the wait for a load at its first use,
        // which is inside the tile loop — a vmcnt(0) there would drain the DMA ring on every iteration
#pragma unroll
        for (int nt = 0; nt < QT; ++nt) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(bq[nt][ks]));
            asm volatile("" : "+v"(tau[nt]));
            if constexpr (NT4) asm volatile("" : "+v"(ntau4[nt]));
            else if constexpr (NEGTAU) asm volatile("" : "+v"(ctau[nt]));
        }
    }

    // Tiles: this block's n-th tile is n * grid + ((block - n) mod grid) — neighbouring tiles go to different blocks (a run
    // of similar rows spreads over many blocks' lists) and the rotation keeps rows that recur with a power-of-two period
    // from landing in the same blocks; args.reverse walks the slab from its end.  The whole slab is visited (the rows of
    // the stage-1 sample included: the sample only tightened tau, its pool is not merged for this kernel), so the
    // sequence needs no skip test and is advanced with a handful of scalar adds.
    // A sample stage (args.group_count != 0) visits 64-row groups group_stride apart instead: the same tile sequence over
    // the sample's tiles, each mapped to its place in the slab.
    constexpr uint32_t TPG = TR <= 64 ? 64 / TR : 1;   // tiles per 64-row sample group (128-row tiles never sample)
    constexpr bool sampled = DBG == 3 || DBG == 7;
    const uint32_t ntiles = sampled ? args.group_count * TPG : (args.nrows + TR - 1) / TR;
    auto tile_row0 = [&](uint32_t t) -> uint32_t {
        return sampled ? (t / TPG) * args.group_stride * 64u + (t % TPG) * TR : t * TR;
    };
    const uint32_t grid = gridDim.x;
    const uint32_t rounds = (ntiles + grid - 1) / grid;
    const uint32_t last_row = args.nrows - 1;
    const unsigned char* slab = reinterpret_cast<const unsigned char*>(args.slab);
    const size_t row_pitch = args.row_stride ? (size_t)args.row_stride : (size_t)ROWB;   // MRL prefix views: rows further apart
    struct Cursor {
        uint32_t n, t, j;   // round, tile number n * grid + j in forward order, j = (block - n) mod grid
    };
    const uint32_t rev_top = rounds * grid - 1;
    auto cursor_tile = [&](const Cursor& c) -> uint32_t {   // >= ntiles: nothing to scan in this round
        return args.reverse ? rev_top - c.t : c.t;
    };
    auto cursor_next = [&](Cursor& c) {   // four scalar operations per tile
        const bool wrap = c.j == 0;
        c.t += wrap ? 2 * grid - 1 : grid - 1;
        c.j = wrap ? grid - 1 : c.j - 1;
        ++c.n;
    };
    // Every block walks all `rounds` rounds; in the first or the last round (the only ragged ones) a block's tile may not exist:
    // it then fetches the last tile again (the counted waits stay exact) and discards the scores — no search loop, a handful of
    // scalar operations per tile.

    // DMA of one tile into a ring slot: wave w issues instructions j = w * PW + x, instruction j = (sub-tile j / KS,
    // k-step j % KS).  Rows past the end are clamped to the last row (their scores are discarded below).
    const int dma_p = lane >> 2, dma_c = lane & 3;
    const int dma_chunk = dma_c ^ ((4 - (dma_p >> 2)) & 3);
    // a wave's DMA instructions go out in groups of at most three (groups of two / one measured 1.36 / 1.40 ms per 512 queries
    // against 1.33 with three: every group is an M0 write + its loads in the MFMA stream)
    constexpr int PARTS = (PW + 2) / 3, PP = PW / PARTS;
    static_assert(PW % PARTS == 0, "a tile's DMA instructions per wave split evenly into groups");
    uint32_t dma_off[PARTS][PP];   // SADDR: this lane's byte offset from the tile's first row, per instruction of the wave
#pragma unroll
    for (int x = 0; x < PW; ++x) {
        const int j = wave * PW + x;
        const int s = j / KS, ks = j - s * KS;
        dma_off[x / PP][x % PP] = (uint32_t)(s * 16 + dma_p) * (uint32_t)row_pitch + (uint32_t)(ks * 64 + dma_chunk * 16);
    }
    auto issue_part = [&](uint32_t t, uint32_t slot, int part) {
        const uint32_t row0 = __builtin_amdgcn_readfirstlane(tile_row0(t));
        if constexpr (SADDR && !SK_NO_DMA) {
            if (row0 + TR <= args.nrows) {   // (wave-uniform) every row of the tile exists: no clamp, no vector address arithmetic
                const uint64_t sbase = (uint64_t)(uintptr_t)slab + (uint64_t)row0 * row_pitch;
                const uint32_t dst = __builtin_amdgcn_readfirstlane(ring + slot * TILE_BYTES + (uint32_t)(wave * PW + part * PP) * 1024u);
                glds16_group<PP>(sbase, dma_off[part], dst);
                return;
            }
        }
#pragma unroll
        for (int x = part * PP; x < part * PP + PP; ++x) {
            const int j = wave * PW + x;
            const int s = j / KS, ks = j - s * KS;
            uint32_t row = row0 + s * 16 + dma_p;
            row = row < args.nrows ? row : last_row;
            const unsigned char* g = slab + (size_t)row * row_pitch + ks * 64 + dma_chunk * 16;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(ring + slot * TILE_BYTES + (uint32_t)j * 1024u);
            if constexpr (!SK_NO_DMA) glds16(g, dst);
            else asm volatile("" ::"v"(g), "s"(dst));
        }
    };

    // fragment reads of chunk c (sub-tile pair c / NCH, k-steps (c % NCH) * CK ...) of a ring slot
    const uint32_t a_off = (uint32_t)frow * 64u + (uint32_t)((fk ^ ((4 - (frow >> 2)) & 3)) << 4);
    [[maybe_unused]] bool frag_ready = false;   // (DBG 5)
    auto slot_base = [&](uint32_t slot) { return smem + (size_t)slot * TILE_BYTES + a_off; };   // one address per tile, not per chunk
    auto read_chunk = [&](const unsigned char* base, int c, half8 (&f)[2][CK]) {
        const int sp = (c / NCH) * 2, k0 = (c % NCH) * CK;
#ifdef FSGPU_EXPERIMENTS
        if constexpr (DBG == 5) {   // timing skeleton: the fragments stay what the prologue read
            if (frag_ready) {
#pragma unroll
                for (int kk = 0; kk < CK; ++kk)
#pragma unroll
                    for (int h = 0; h < 2; ++h) asm volatile("" : "+v"(f[h][kk]));
                return;
            }
        }
#endif
#pragma unroll
        for (int kk = 0; kk < CK; ++kk)
#pragma unroll
            for (int h = 0; h < 2; ++h) f[h][kk] = *reinterpret_cast<const half8*>(base + ((sp + h) * KS + k0 + kk) * 1024);
    };
    auto mfma_chunk = [&](int c, const half8 (&f)[2][CK], acc_t (&acc)[2][QT]) {
        const int k0 = (c % NCH) * CK;
#ifdef FSGPU_EXPERIMENTS
        if constexpr (DBG == 1) {
#pragma unroll
            for (int kk = 0; kk < CK; ++kk)
#pragma unroll
                for (int h = 0; h < 2; ++h) asm volatile("" ::"v"(f[h][kk]));
            return;
        }
#endif
#pragma unroll
        for (int kk = 0; kk < CK; ++kk)
#pragma unroll
            for (int nt = 0; nt < QT; ++nt)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    if constexpr (EB == 2)
                        acc[h][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[h][kk], bq[nt][k0 + kk], acc[h][nt], 0, 0, 0);
                    else
                        acc[h][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, f[h][kk]),
                                                                           __builtin_bit_cast(i32x4, bq[nt][k0 + kk]), acc[h][nt], 0, 0, 0);
    };
    // rows of a finished sub-tile pair at or above the threshold -> the block's list of their query (straight to global
    // memory: a few entries per query and block, so LDS holds only the counters)
    // byte offset of this lane's list for its wave's query tile 0: ((q * grid + block) * slots) * 8 — NQ x grid x slots x 8 <= 42 MB
    const uint32_t list0 = (((uint32_t)q0_wave + (uint32_t)(threadIdx.x & 15)) * gridDim.x + blockIdx.x) * args.slots * 8u;
    const uint32_t tile_step = 16u * gridDim.x * args.slots * 8u;   // ... per query tile (scalar)
    uint32_t lcount = 0;   // entries this wave has appended to the block's list of each of its queries: 6 bits per query tile, saturating
    static_assert(QT * 6 <= 32 && kWideSlots < 63, "one register holds the wave's list lengths");
    auto lane_max = [&](const acc_t (&acc)[2][QT], int nt) {
        auto m = acc[0][nt][0];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (EB == 2) m = __builtin_fmaxf(m, acc[h][nt][r]);   // (max ignores NaN, and NaN >= tau is false: a NaN
                else m = acc[h][nt][r] > m ? acc[h][nt][r] : m;                 // score never passes, as in scan_mfma_kernel)
            }
        return m;
    };
    // NEGTAU: the accumulators hold score - ceil(tau), a score passes iff its accumulator is >= 0
    auto passes = [&](int nt, auto v) {
        if constexpr (NEGTAU) return v >= 0;
        else return (float)v >= tau[nt];
    };
    auto score_of = [&](int nt, auto v) {
        if constexpr (NEGTAU) return (float)(v + ctau[nt]);
        else return (float)v;
    };
    // does any of the lane's 8 scores per query tile in [nt_lo, nt_hi) reach its query's threshold?
    auto any_passes = [&](const acc_t (&acc)[2][QT], int nt_lo, int nt_hi) {
        bool any = false;
        if constexpr (NEGTAU) {
            int m = lane_max(acc, nt_lo);
#pragma unroll
            for (int nt = nt_lo + 1; nt < nt_hi; ++nt) {
                const int x = lane_max(acc, nt);
                m = x > m ? x : m;
            }
            any = m >= 0;
        } else {
#pragma unroll
            for (int nt = nt_lo; nt < nt_hi; ++nt) any |= passes(nt, lane_max(acc, nt));
        }
        return any;
    };
    // The append path.  A query belongs to ONE wave (q = q0 + nt * 16 + frow), so the length of its (query, block) list needs no
    // shared counter: it lives in a register — 6 bits per query tile in `lcount`, identical in the four lanes (fk = 0..3) that hold
    // the query's column — and a passing score's slot comes from a wave vote: slot = count + passing lanes of the same column below
    // this one.  The whole wave takes the excursion (the caller's test is a vote), nothing in it is waited for: no LDS atomic, no
    // load — the list's address is the kernel argument's scalar base + a 32-bit lane offset (r04 kept eight 64-bit list / spill
    // addresses per lane, which spilled: every entry paid a scratch reload + s_waitcnt vmcnt(0), i.e. a drain of the DMA ring the
    // loop never drains — 1,600 cycles per entry, profiles/r04/wide_stamps.txt), the entry leaves through a store that nobody waits
    // for, and only a list that overflows its slots (rare: the spill area) touches a returning global atomic.
    auto emit_tiles = [&](uint32_t t, int sp, const acc_t (&acc)[2][QT], int nt_lo, int nt_hi) {
#ifdef FSGPU_EXPERIMENTS
        if constexpr (DBG == 2 || DBG == 4 || DBG == 5 || SK) {  // (timing skeletons read stale bytes: keep the scores live, append nothing)
            lcnt[q0 + frow] = 0;
            return;
        }
#endif
        if (t >= ntiles) return;   // (wave-uniform) a ragged round: the slot holds the last tile again
        [[maybe_unused]] unsigned long long st_in = 0;
        if constexpr (STAMPS) st_in = __builtin_readcyclecounter();
#ifdef FSGPU_LAB_APPEND_PRIO
        // (lab) the partner wave of this SIMD keeps issuing matrix instructions; at equal priority the OLDER wave's vector instructions
        // go first and an excursion on the younger one crawls (MI355X_MICROARCH.md, "Two waves per SIMD", item 2)
        __builtin_amdgcn_s_setprio(FSGPU_LAB_APPEND_PRIO);
#endif
        // C layout: column (query) = lane & 15, row = (lane >> 4) * 4 + reg
        const uint32_t pair_row0 = __builtin_amdgcn_readfirstlane(tile_row0(t) + sp * 16);   // a multiple of 32: the pair's 32 rows share one bitmap word
        // the pair's 32 rows: bit i set = row pair_row0 + i exists, is live and allowed (scalar arithmetic)
        uint32_t ok32 = pair_row0 + 32u <= args.nrows ? ~0u : pair_row0 >= args.nrows ? 0u : (1u << (args.nrows - pair_row0)) - 1u;
        if (args.live || args.allow) {
            uint32_t wi = pair_row0 >> 6;   // (a ragged last tile: the pair may start past the last row — its rows are rejected above)
            wi = __builtin_amdgcn_readfirstlane(wi < (last_row >> 6) ? wi : (last_row >> 6));
            u64 mask = ~0ull;
#ifdef FSGPU_LAB_VECTOR_BITMAP   // (lab: the r03 state before the scalar loads — waits behind every DMA in flight)
            if (args.live) mask &= __builtin_nontemporal_load(args.live + wi);
            if (args.allow) mask &= __builtin_nontemporal_load(args.allow + wi);
#elif defined(FSGPU_LAB_BITMAP_CHECK)
            if (args.live) {
                const u64 sv = sload_u64(args.live + wi);
                bm_check(args.live + wi, sv, wi, 0, wave);
                mask &= sv;
            }
            if (args.allow) {
                const u64 sv = sload_u64(args.allow + wi);
                bm_check(args.allow + wi, sv, wi, 1, wave);
                mask &= sv;
            }
#else
            if (args.live) mask &= sload_u64(args.live + wi);
            if (args.allow) mask &= sload_u64(args.allow + wi);
#endif
            ok32 &= (uint32_t)(mask >> (pair_row0 & 32u));
        }
        // Everything per lane below derives from three values made opaque HERE, so that hipcc computes it inside the excursion
        // instead of keeping it in registers across the tile loop (which has none to spare).
        uint32_t fk_v = (uint32_t)fk, frow_v = (uint32_t)frow, list_v = list0;
        asm volatile("" : "+v"(fk_v), "+v"(frow_v), "+v"(list_v));
        const uint32_t below_mask = (1u << fk_v) - 1u;
        const uint32_t row00 = pair_row0 + fk_v * 4u;
        unsigned char* const cand_bytes = reinterpret_cast<unsigned char*>(args.cand);
        // one passing score: its slot from the vote, the entry stored, the column's count moved on in all four of its lanes
        auto take = [&](int nt, int h, int r, bool pass, u64 vote) __attribute__((always_inline)) {
            // the four lanes of this lane's query column: frow + 16 j
            const uint32_t vlo = (uint32_t)vote >> frow_v, vhi = (uint32_t)(vote >> 32) >> frow_v;
            const uint32_t col = (vlo & 1u) | ((vlo >> 15) & 2u) | ((vhi & 1u) << 2) | ((vhi >> 13) & 8u);
            const uint32_t have = (lcount >> (6 * nt)) & 63u;
            if (pass) {
                const uint32_t pos = have + (uint32_t)__builtin_popcount(col & below_mask);
                const u64 entry = pack(score_of(nt, acc[h][nt][r]), args.row_base + row00 + h * 16 + r);
                if (__builtin_expect(pos < (uint32_t)slots, 1)) {
                    *reinterpret_cast<u64*>(cand_bytes + (size_t)(uint32_t)(list_v + (uint32_t)nt * tile_step + pos * 8u)) = entry;
                } else {
                    const uint32_t q = (uint32_t)q0 + (uint32_t)nt * 16u + frow_v;
                    const uint32_t g = atomicAdd(&args.spill_count[q * kMfmaSpillCountStride], 1u);
                    if (g < args.spill_cap) args.spill[(size_t)q * args.spill_cap + g] = entry;
                    else args.overflow[q] = 1;
                }
            }
            const uint32_t now = have + (uint32_t)__builtin_popcount(col);
            lcount = (lcount & ~(63u << (6 * nt))) | ((now < 63u ? now : 63u) << (6 * nt));
        };
        if (__builtin_expect(ok32 == ~0u, 1)) {
            // every row of the pair exists, is live and allowed: the votes are the bare threshold tests, and the tests fall through
            // (a vote that finds something branches out and back: one passing score per excursion is the rule)
#pragma unroll
            for (int nt = nt_lo; nt < nt_hi; ++nt) {
                // query tiles first: the tiles without a passing score are skipped as a whole
                if (!wave_any(passes(nt, lane_max(acc, nt)))) continue;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool pass = passes(nt, acc[h][nt][r]);
                        const u64 vote = __ballot(pass);
                        if (__builtin_expect(vote != 0, 0)) take(nt, h, r, pass, vote);
                    }
            }
        } else {
            const uint32_t lb = ok32 >> (fk_v * 4u);   // this lane's 8 rows: bits h * 16 + r of its view
#pragma unroll
            for (int nt = nt_lo; nt < nt_hi; ++nt) {
                if (!wave_any(passes(nt, lane_max(acc, nt)))) continue;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool pass = passes(nt, acc[h][nt][r]) && ((lb >> (h * 16 + r)) & 1u);
                        const u64 vote = __ballot(pass);
                        if (__builtin_expect(vote != 0, 0)) take(nt, h, r, pass, vote);
                    }
            }
        }
#ifdef FSGPU_LAB_APPEND_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        if constexpr (STAMPS) {
            const unsigned long long dt = __builtin_readcyclecounter() - st_in;
            if (lane == 0) {
                unsigned long long* st = reinterpret_cast<unsigned long long*>(lcnt + NQ + 16) + wave * 2;
                atomicAdd(st, dt);
                atomicAdd(st + 1, 1ull);
            }
        }
    };
    // (the slow paths are marked unexpected: left to itself hipcc lays every emit_tiles body out INSIDE the tile loop — nine copies of
    // ~1,000 instructions between the MFMA groups, each skipped by a taken branch: the loop's 192 MFMAs spread over 7,000 instructions;
    // with the hint the loop is ~480 contiguous instructions that fall through and the append code sits behind the function's end)
    auto emit_pair = [&](uint32_t t, int sp, const acc_t (&acc)[2][QT]) {
        // one test for the whole pair; almost always negative: survivors are a few hundred rows of the slab
        if (__builtin_expect(wave_any(any_passes(acc, 0, QT)), 0)) emit_tiles(t, sp, acc, 0, QT);
    };

    // The append path reads the tombstone / allow words with SCALAR loads (sload_u64).  The scalar data cache is not invalidated
    // between the kernels of a stream the way the vector L1 is — compiler-made scalar loads only ever touch kernel arguments and
    // constants — so a line cached by an earlier kernel (the previous call's allow bitmap in the same workspace; a closed index's
    // bitmap at a reused address) could be served stale: one repetition in ~30,000 returned a filtered-out row or lost an allowed
    // one (scripts/r03/determinism.py).  Every wave therefore invalidates the scalar cache once before its first bitmap word; the
    // bitmaps do not change while the kernel runs.
#ifndef FSGPU_LAB_NO_DCACHE_INV
    if (args.live || args.allow) asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    __syncthreads();  // counters initialised (no DMA in flight yet)
    Cursor cl{0, blockIdx.x, blockIdx.x};   // next tile to fetch
    Cursor cc = cl;                         // next tile to consume
    // prologue: NSLOT - 1 tiles in flight (tile j -> slot j mod NSLOT throughout).  Past the last round the DMA count is kept
    // up with dummy tiles (the last tile again) so that the counted waits below stay exact.
    auto fetch_part = [&](uint32_t slot, int part) {   // part PARTS - 1 completes the tile and moves the cursor on
        uint32_t t = cl.n < rounds ? cursor_tile(cl) : ntiles;
        t = t < ntiles ? t : ntiles - 1;
        issue_part(t, slot, part);
        if (part == PARTS - 1 && cl.n < rounds) cursor_next(cl);
    };
    auto fetch_next = [&](uint32_t slot) {
#pragma unroll
        for (int part = 0; part < PARTS; ++part) fetch_part(slot, part);
    };
#pragma unroll
    for (int j = 0; j < NSLOT - 1; ++j) fetch_next((uint32_t)j);
    if constexpr (SPLIT) {
        constexpr int NP = RS / 2;                  // sub-tile pairs per tile
        constexpr int QA = QT / 2;                  // phase 1: query tiles [0, QA), phase 2: [QA, QT)
        half8 f[2][KS];                             // the current pair's fragments: [sub-tile of the pair][k-step]
        acc_t acc[2][QT];
        [[maybe_unused]] bool sk_loop = false;
        auto read_frag = [&](const unsigned char* base, int pair, int kk) {
            if constexpr (SK_NO_READS) {
                if (sk_loop) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) asm volatile("" : "+v"(f[h][kk]));
                    return;
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) f[h][kk] = *reinterpret_cast<const half8*>(base + ((pair * 2 + h) * KS + kk) * 1024);
        };
        auto sk_any = [&](int nt_lo, int nt_hi) {   // (skeleton without tests: the scores stay live, nothing is compared)
            if constexpr (SK_NO_TEST) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int nt = nt_lo; nt < nt_hi; ++nt) asm volatile("" ::"v"(acc[h][nt]));
                return false;
            } else {
                return any_passes(acc, nt_lo, nt_hi);
            }
        };
        auto mfma_step = [&](int kk, int nt_lo, int nt_hi) {
#pragma unroll
            for (int nt = nt_lo; nt < nt_hi; ++nt)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    acc[h][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, f[h][kk]),
                                                                       __builtin_bit_cast(i32x4, bq[nt][kk]), acc[h][nt], 0, 0, 0);
        };
        auto init_acc = [&](int nt_lo, int nt_hi) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int nt = nt_lo; nt < nt_hi; ++nt) acc[h][nt] = i32x4{-ctau[nt], -ctau[nt], -ctau[nt], -ctau[nt]};
        };
        wait_vmcnt<PW*(NSLOT - 2)>();          // this wave's share of tile 0 has landed ...
        if constexpr (FLAGS) {
            flag_post(flags, lane);
            flag_wait(flags, WPB);             // ... and everyone's
        } else {
            __builtin_amdgcn_s_barrier();      // ... and everyone's
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) read_frag(slot_base(0), 0, kk);
        init_acc(0, QT);                        // (so that the first, void, test of the second half reads defined values)
        if constexpr (SK_NO_READS) sk_loop = true;
        uint32_t slot = 0, slot_prev = NSLOT - 1;
        const unsigned char* cur = slot_base(0);
        uint32_t tB = ntiles;                   // tile of the pair whose second-half scores are still to be tested (none yet)
        // GMAX: the lane's best score per query tile so far and the first row of the sub-tile pair it was seen in
        [[maybe_unused]] int gbest[QT];
        [[maybe_unused]] uint32_t grow[QT];
        if constexpr (GMAX) {
#pragma unroll
            for (int nt = 0; nt < QT; ++nt) {
                gbest[nt] = -0x7fffffff - 1;
                grow[nt] = 0;
            }
        }
        auto gmax_update = [&](uint32_t tt, int sp, int nt_lo, int nt_hi) __attribute__((always_inline)) {
            if constexpr (GMAX) {
                const uint32_t g = tile_row0(tt < ntiles ? tt : 0u) + (uint32_t)sp * 16u;
                // (wave-uniform) a ragged round's missing tile holds the last tile again, and a pair that reaches past the last row holds
                // copies of it: neither may stand for a group — two groups must never report the same row
                if (tt < ntiles && g + 32u <= args.nrows) {
                    if (args.live || args.allow) {
                        // tombstoned / filtered-out rows do not stand for their group: the pair's 32 rows share one bitmap word (scalar
                        // loads), a lane's 8 rows are bits fk * 4 + r and 16 + fk * 4 + r of the pair's half of it
                        u64 mask = ~0ull;
                        if (args.live) mask &= sload_u64(args.live + (g >> 6));
                        if (args.allow) mask &= sload_u64(args.allow + (g >> 6));
                        const uint32_t lb = (uint32_t)(mask >> (g & 63u)) >> (fk * 4);
                        bool keep[2][4];
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) keep[h][rr] = ((lb >> (h * 16 + rr)) & 1u) != 0;
#pragma unroll
                        for (int nt = nt_lo; nt < nt_hi; ++nt) {
                            int v = -0x7fffffff - 1;
#pragma unroll
                            for (int h = 0; h < 2; ++h)
#pragma unroll
                                for (int rr = 0; rr < 4; ++rr) {
                                    const int x = keep[h][rr] ? acc[h][nt][rr] : -0x7fffffff - 1;
                                    v = x > v ? x : v;
                                }
                            const bool better = v > gbest[nt];
                            gbest[nt] = better ? v : gbest[nt];
                            grow[nt] = better ? g : grow[nt];
                        }
                    } else {
#pragma unroll
                        for (int nt = nt_lo; nt < nt_hi; ++nt) {
                            const int v = lane_max(acc, nt);
                            const bool better = v > gbest[nt];
                            gbest[nt] = better ? v : gbest[nt];
                            grow[nt] = better ? g : grow[nt];
                        }
                    }
                }
            }
        };
        if constexpr (STAMPS) st_loop = __builtin_readcyclecounter();
        while (cc.n < rounds) {
            const uint32_t slot_next = slot + 1 == NSLOT ? 0 : slot + 1;
            const unsigned char* nxt = slot_base(slot_next);
            const uint32_t t = cursor_tile(cc);   // (a ragged round's missing tile is recognised in the slow path: emit_tiles)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                constexpr int TK = KS >= 4 ? 2 : 0;   // the k-step after which a phase carries the other half's test
                // ---- phase 1: query tiles [0, QA); the previous pair's tiles [QA, QT) are tested in its shadow
                init_acc(0, QA);
                bool anyB = false;
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    mfma_step(kk, 0, QA);
                    if constexpr (GMAX) {
                        if (kk == TK) gmax_update(tB, ((p + NP - 1) % NP) * 2, QA, QT);
                    } else {
                        if (kk == TK) anyB = sk_any(QA, QT);
                    }
                    if (p == 0 && kk == 0 && !SK_NO_SYNC) {
                        // The tile's barrier.  Tile n+1: this wave's DMAs have landed (tiles n+2 .. n+NSLOT-2 may still be in
                        // flight), then everyone's; every wave is also past its last read of tile n-1 (this tile's first pair was
                        // read behind them and has been waited for), whose slot takes tile n+NSLOT-1.
                        [[maybe_unused]] unsigned long long st_a = 0;
                        if constexpr (STAMPS) st_a = __builtin_readcyclecounter();
                        wait_vmcnt<PW*(NSLOT - 3)>();
                        if constexpr (FLAGS) {
                            flag_post(flags + slot_next * 4, lane);   // this wave's share of tile n+1 is in LDS (posted a tile early)
                        } else {
                            __builtin_amdgcn_s_barrier();
                            asm volatile("" ::: "memory");
                        }
                        if constexpr (STAMPS) st_bar += __builtin_readcyclecounter() - st_a;
                    }
                    // the refill of tile n-1's slot: one group of DMA instructions per pair, behind a few MFMAs of its phase 1
                    // (all of them behind the first pair's first MFMAs was measured: 1.43 instead of 1.36 ms — the DMA's LDS writes
                    // then collide with the pair's fragment reads)
                    {
                        constexpr int IPP = (PARTS + NP - 1) / NP;   // issue points per pair
                        constexpr int K1 = KS >= 3 ? 2 : KS - 1, K2 = KS >= 5 ? 4 : K1;
                        if constexpr (FLAGS) {
                            // tile n-1's slot takes tile n+2 once EVERY wave is past its last read of tile n-1 (posted a pair before
                            // the end of that tile: in the common case long ago)
                            if (p == 0 && kk == K1 && cc.n >= 1) flag_wait(flags + 16 + slot_prev * 4, WPB * ((cc.n - 1) / NSLOT + 1));
                        }
                        if (kk == K1 && p * IPP < PARTS) fetch_part(slot_prev, p * IPP);
                        if (IPP > 1 && K2 != K1 && kk == K2 && p * IPP + 1 < PARTS) fetch_part(slot_prev, p * IPP + 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (__builtin_expect(wave_any(anyB), 0)) emit_tiles(tB, ((p + NP - 1) % NP) * 2, acc, QA, QT);
                // ---- phase 2: query tiles [QA, QT); the next pair's fragments roll in behind each k-step; this pair's tiles
                // [0, QA) are tested in its shadow
                init_acc(QA, QT);
                bool anyA = false;
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    mfma_step(kk, QA, QT);
                    if constexpr (FLAGS) {   // the next tile's first fragments: everyone's share of it must be in LDS
                        if (p + 1 == NP && kk == 0) flag_wait(flags + slot_next * 4, WPB * ((cc.n + 1) / NSLOT + 1));
                    }
                    if (p + 1 < NP) read_frag(cur, p + 1, kk);
                    else read_frag(nxt, 0, kk);
                    if constexpr (GMAX) {
                        if (kk == TK) gmax_update(t, p * 2, 0, QA);
                    } else {
                        if (kk == TK) anyA = sk_any(0, QA);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (FLAGS) {   // this wave's last read of the tile is out (the last pair's fragments): its slot may be refilled
                    if (p + 2 == NP) flag_post(flags + 16 + slot * 4, lane);
                }
                if (__builtin_expect(wave_any(anyA), 0)) emit_tiles(t, p * 2, acc, 0, QA);
                tB = t;
            }
            cursor_next(cc);
            slot_prev = slot;
            slot = slot_next;
            cur = nxt;
        }
        if constexpr (GMAX) {
            gmax_update(tB, (NP - 1) * 2, QA, QT);
            // the block's four groups per query: [q][block][lane >> 4]
            wait_vmcnt<0>();
#pragma unroll
            for (int nt = 0; nt < QT; ++nt) {
                const u64 e = gbest[nt] == -0x7fffffff - 1 ? kEmpty : pack((float)gbest[nt], args.row_base + grow[nt] + (uint32_t)fk * 4u);
                args.cand[((size_t)(q0 + nt * 16 + frow) * gridDim.x + blockIdx.x) * 4 + fk] = e;
            }
            __syncthreads();   // (no DMA may outlive the block's LDS allocation: every wave has waited for its own above)
            return;
        } else {
            if (__builtin_expect(wave_any(any_passes(acc, QA, QT)), 0)) emit_tiles(tB, (NP - 1) * 2, acc, QA, QT);
        }
    } else {
    half8 fa[2][2][CK];   // fragment double buffer: [buffer][sub-tile of the pair][k-step of the chunk]
    wait_vmcnt<PW*(NSLOT - 2)>();          // this wave's share of tile 0 has landed ...
    __builtin_amdgcn_s_barrier();          // ... and everyone's
    asm volatile("" ::: "memory");
    read_chunk(slot_base(0), 0, fa[0]);
#ifdef FSGPU_EXPERIMENTS
    if constexpr (DBG == 5) {
        read_chunk(slot_base(0), 1, fa[1]);
        frag_ready = true;
    }
#endif
    uint32_t slot = 0, slot_prev = NSLOT - 1;
    const unsigned char* cur = slot_base(0);
    while (cc.n < rounds) {
        const uint32_t slot_next = slot + 1 == NSLOT ? 0 : slot + 1;
        const unsigned char* nxt = slot_base(slot_next);
        const uint32_t t = cursor_tile(cc);   // (a ragged round's missing tile is recognised in the slow path: emit_tiles)
        acc_t acc[2][QT];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (c % NCH == 0) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int nt = 0; nt < QT; ++nt) {
                        if constexpr (NT4) acc[h][nt] = ntau4[nt];
                        else if constexpr (NEGTAU) acc[h][nt] = acc_t{-ctau[nt], -ctau[nt], -ctau[nt], -ctau[nt]};
                        else acc[h][nt] = acc_t{0, 0, 0, 0};
                    }
            }
            // the NEXT chunk's fragment reads go out before this chunk's MFMAs (the next tile's first chunk after the last)
            if (c + 1 < NC) read_chunk(cur, c + 1, fa[(c + 1) & 1]);
            else read_chunk(nxt, 0, fa[0]);
            __builtin_amdgcn_sched_barrier(0);
            mfma_chunk(c, fa[c & 1], acc);
            __builtin_amdgcn_sched_barrier(0);
            if (c % NCH == NCH - 1) emit_pair(t, (c / NCH) * 2, acc);
            // The tile's barrier.  Tile n+1: this wave's DMAs have landed (tiles n+2 .. n+NSLOT-2 may still be in flight), then
            // everyone's; every wave is also past its last read of tile n-1, whose slot takes tile n+NSLOT-1.
            if (c == 0 && DBG != 4) {   // (DBG 4: timing skeleton without the tile's wait and barrier)
                wait_vmcnt<PW*(NSLOT - 3)>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            // The DMA issue (a stretch without MFMAs) is staggered between the two waves that share a SIMD (waves w and
            // w + 4): one right after the barrier, the other a chunk later.
            if (NC >= 2 && c <= 1) {
                if ((wave >= 4) == (c == 1)) fetch_next(slot_prev);
            } else if (NC < 2 && c == 0) {
                fetch_next(slot_prev);
            }
        }
        cursor_next(cc);
        slot_prev = slot;
        slot = slot_next;
        cur = nxt;
    }
    }
    wait_vmcnt<0>();  // no DMA may outlive the block's LDS allocation
    if (fk == 0) {    // the list lengths, from the registers of the lanes that hold each query's column
#pragma unroll
        for (int nt = 0; nt < QT; ++nt) lcnt[q0 + nt * 16 + frow] = (int)((lcount >> (6 * nt)) & 63u);
    }
    __syncthreads();
    if constexpr (STAMPS) {
        if (args.dense && lane == 0) {
            const unsigned long long* st = reinterpret_cast<const unsigned long long*>(lcnt + NQ + 16) + wave * 2;
            unsigned long long* out = reinterpret_cast<unsigned long long*>(args.dense) + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * WPB + wave) * 4;
            out[0] = __builtin_readcyclecounter() - st_loop;
            out[1] = st_bar;
            out[2] = st[0];
            out[3] = st[1];
        }
    }
    if (args.cand_count) {
        // the lists' lengths instead of padding: the selection reads only what was appended (2 KB per block instead of
        // NQ x slots x 8 bytes — 128 KB at 512 queries x 32 slots)
        for (int q = tid; q < NQ; q += NT) {
            const int c = lcnt[q];
            args.cand_count[(size_t)q * gridDim.x + blockIdx.x] = (uint32_t)(c < slots ? c : slots);
        }
        return;
    }
    // pad the block's lists: [q][block][slots], kEmpty beyond the entries appended
    for (int i = tid; i < NQ * slots; i += NT) {
        const int q = i / slots, j = i - q * slots;
        if (j >= lcnt[q]) args.cand[((size_t)q * gridDim.x + blockIdx.x) * slots + j] = kEmpty;
    }
}

template <int ROWB, int EB, int QT, int NSLOT, int OPT, int DBG = 0>
__global__ __launch_bounds__(512) void scan_wide_kernel(MfmaScanArgs args) {
    scan_wide_body<ROWB, EB, QT, NSLOT, OPT, DBG, 8, 8 * QT * 16>(args, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * QT * 16);
}

// Lab (FSGPU_LAB_ASYM): the block's first four waves — the ones the oldest-first arbiter favours (profiles/r04/wide_stamps.txt: they wait
// 36 % of their loop at the tile barrier) — hold QTA query tiles each, the last four QTB: 64 x (QTA + QTB) queries per block.
template <int ROWB, int EB, int QTA, int QTB, int NSLOT, int OPT>
__global__ __launch_bounds__(512) void scan_wide_asym_kernel(MfmaScanArgs args) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wave < 4) scan_wide_body<ROWB, EB, QTA, NSLOT, OPT, 0, 8, 64 * (QTA + QTB)>(args, wave * QTA * 16);
    else scan_wide_body<ROWB, EB, QTB, NSLOT, OPT, 0, 8, 64 * (QTA + QTB)>(args, 64 * QTA + (wave - 4) * QTB * 16);
}

// ---- launcher ----------------------------------------------------------------------------------------------------

namespace {

// what the shipped kernels are built with: neg-tau + scalar-base DMA + query-tile-split loop wherever the shape's registers allow,
// 128-row tiles in a three-slot ring for the main pass of those shapes
// (measured on the bench shape, profiles/r03/wide_opt_ab_run*.txt: 1.49 ms -> 1.33 ms per 512 queries at 10M x 384)
#ifdef FSGPU_WIDE_OPT_DEFAULT
constexpr int kWideOptDefault = FSGPU_WIDE_OPT_DEFAULT;
#else
constexpr int kWideOptDefault = kOptNegTau | kOptSaddr | kOptSplit | kOptBig;
#endif

#ifdef FSGPU_EXPERIMENTS
int wide_env(const char* name) {
    const char* e = std::getenv(name);
    return e ? std::atoi(e) : -1;
}
#endif

template <int ROWB, int EB, int QT, int NSLOT, int OPT, int DBG = 0>
hipError_t launch_wide_t(const MfmaScanArgs& args, int grid, hipStream_t stream, int* occupancy) {
    constexpr bool BIG = wide_big_ok(ROWB, EB, QT, NSLOT, OPT, DBG);
    constexpr int TR = BIG ? 128 : ROWB >= 512 ? 32 : 64;
    constexpr size_t ring = (size_t)NSLOT * TR * ROWB;
    const size_t lds = ring + (size_t)QT * 128 * 4 + 64 + (DBG == 8 ? 256 : 0);   // the row-tile ring + one append counter per query + the ring's arrive / wait counters
    auto kern = scan_wide_kernel<ROWB, EB, QT, NSLOT, OPT, DBG>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (occupancy) {
        int blocks = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kern, 512, lds) != hipSuccess || blocks < 1) blocks = 1;
        *occupancy = blocks;
        return hipSuccess;
    }
    if (args.slots > kWideSlots) return hipErrorInvalidValue;
    static const std::string name = "scan_wide_kernel<" + std::to_string(ROWB) + ", " + std::to_string(EB) + ", " +
                                    std::to_string(QT) + ", " + std::to_string(NSLOT) + ", " + std::to_string(OPT) + ">";
    if (DBG != 3 && DBG != 7) note_main_pass_kernel(name.c_str());
    hipLaunchKernelGGL(kern, dim3(grid, args.groups ? args.groups : 1), dim3(512), lds, stream, args);
    return hipGetLastError();
}

#ifdef FSGPU_LAB_ASYM
template <int ROWB, int EB, int QTA, int QTB, int NSLOT, int OPT>
hipError_t launch_wide_asym(const MfmaScanArgs& args, int grid, hipStream_t stream) {
    constexpr size_t ring = (size_t)NSLOT * 64 * ROWB;
    const size_t lds = ring + (size_t)64 * (QTA + QTB) * 4 + 64;
    auto kern = scan_wide_asym_kernel<ROWB, EB, QTA, QTB, NSLOT, OPT>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (args.slots > kWideSlots) return hipErrorInvalidValue;
    note_main_pass_kernel("scan_wide_kernel<384, 1, asym>");
    hipLaunchKernelGGL(kern, dim3(grid, args.groups ? args.groups : 1), dim3(512), lds, stream, args);
    return hipGetLastError();
}
#endif

// the main pass (MODE 0) of a shape that runs the split loop takes the 128-row / three-slot form; everything else NSLOT slots
template <int ROWB, int EB, int QT, int NSLOT, int O, int MODE>
hipError_t launch_wide_pick(const MfmaScanArgs& args, int grid, hipStream_t stream, int* occupancy) {
    if constexpr (MODE == 0 && wide_big_ok(ROWB, EB, QT, 3, O, 0)) return launch_wide_t<ROWB, EB, QT, 3, O, 0>(args, grid, stream, occupancy);
    else return launch_wide_t<ROWB, EB, QT, NSLOT, O & ~(kOptBig | kOptFlags), MODE>(args, grid, stream, occupancy);
}

template <int EB, int QT, int MODE = 0>
hipError_t launch_wide_d(const MfmaScanArgs& args, int grid, hipStream_t stream, int* occupancy) {
    // resident query fragments: QT x (row bytes / 64) x 4 registers per lane; 144 is what fits next to the accumulators and
    // the fragment double buffer (f16 rows of 768 bytes: 3 tiles; int8 rows of 384 bytes: 6)
    if (QT * (int)(args.dim * EB / 64) * 4 > 144) return hipErrorInvalidValue;
    constexpr int O = kWideOptDefault;
    switch (args.dim * EB / 2) {  // row length in 2-byte units
        case 384: if constexpr (QT <= 3 && MODE != 7) {
#ifdef FSGPU_EXPERIMENTS
            if constexpr (MODE == 0 && EB == 2 && QT == 2) {   // timing skeletons: answers are NOT valid
                static const int dbg = wide_env("FSGPU_WIDE_DBG");
                if (dbg == 1) return launch_wide_t<768, EB, QT, 6, O, 1>(args, grid, stream, occupancy);
                if (dbg == 2) return launch_wide_t<768, EB, QT, 6, O, 2>(args, grid, stream, occupancy);
            }
#endif
            return launch_wide_t<768, EB, QT, 6, O & ~(kOptBig | kOptFlags), MODE>(args, grid, stream, occupancy);   // 6 x 24 KB
        } else return hipErrorInvalidValue;
        case 256: if constexpr (QT <= 4 && MODE != 7) return launch_wide_t<512, EB, QT, 8, O & ~kOptFlags, MODE>(args, grid, stream, occupancy);   // 8 x 16 KB
                  else return hipErrorInvalidValue;
        // Rows of 384 / 256 bytes: the int8 copies of 384- and 256-dimensional slabs (MiniLM / the hash embedders, potion).  Their
        // f16 forms (192 / 128 dimensions) and rows of 128 bytes are not shapes of the reference's embedders: those searches run on
        // scan_mfma_kernel (queries in LDS), and so does a group-maxima stage the split loop cannot hold (5 tiles of 384 bytes).
        case 192: if constexpr (EB != 1 || (MODE == 7 && !wide_split_ok(384, 1, QT, O, 7))) return hipErrorInvalidValue; else {
#ifdef FSGPU_EXPERIMENTS
            if constexpr (EB == 1 && QT == 4) {   // A/B runs of the kernel's options on the bench shape (int8 rows of 384 dimensions)
                static const int opt = wide_env("FSGPU_WIDE_OPT");
                static const int dbg = wide_env("FSGPU_WIDE_DBG");
                if constexpr (MODE == 0) {
                    if (dbg == 1) return launch_wide_t<384, EB, QT, 6, 6, 1>(args, grid, stream, occupancy);
                    if (dbg == 2) return launch_wide_t<384, EB, QT, 6, 6, 2>(args, grid, stream, occupancy);
                    if (dbg == 4) return launch_wide_t<384, EB, QT, 6, 6, 4>(args, grid, stream, occupancy);
                    if (dbg == 5) return launch_wide_t<384, EB, QT, 6, 6, 5>(args, grid, stream, occupancy);
                }
                switch (opt) {
                    case 0: return launch_wide_t<384, EB, QT, 6, 0, MODE>(args, grid, stream, occupancy);
                    case 2: return launch_wide_t<384, EB, QT, 6, 2, MODE>(args, grid, stream, occupancy);
                    case 4: return launch_wide_t<384, EB, QT, 6, 4, MODE>(args, grid, stream, occupancy);
                    case 6: return launch_wide_t<384, EB, QT, 6, 6, MODE>(args, grid, stream, occupancy);
                    case 14: return launch_wide_t<384, EB, QT, 6, 14, MODE>(args, grid, stream, occupancy);
                    case 30:   // 128-row tiles, three slots: the main pass only (a sample stage visits 64-row groups)
                        if constexpr (MODE == 0) return launch_wide_t<384, EB, QT, 3, 30, MODE>(args, grid, stream, occupancy);
                        else return launch_wide_t<384, EB, QT, 6, 14, MODE>(args, grid, stream, occupancy);
                    default: break;
                }
                if constexpr (MODE == 0) {   // the shipped main pass with s_memtime stamps (scripts/r04/wide_stamps.sh): one line per launch
                    if (dbg == 8 && !occupancy) {
                        static unsigned long long* buf = nullptr;
                        constexpr size_t kWords = 2 * 1024 * 8 * 4;   // [group][block][wave][4]
                        if (!buf && hipMalloc(reinterpret_cast<void**>(&buf), kWords * 8) != hipSuccess) return hipErrorOutOfMemory;
                        MfmaScanArgs a2 = args;
                        a2.dense = reinterpret_cast<u64*>(buf);
                        (void)hipMemsetAsync(buf, 0, kWords * 8, stream);
                        const hipError_t e = launch_wide_t<384, EB, QT, 3, O, 8>(a2, grid, stream, nullptr);
                        if (e != hipSuccess) return e;
                        (void)hipStreamSynchronize(stream);
                        const size_t nwaves = (size_t)grid * (args.groups ? args.groups : 1) * 8;
                        std::vector<unsigned long long> h(nwaves * 4);
                        (void)hipMemcpy(h.data(), buf, h.size() * 8, hipMemcpyDeviceToHost);
                        double loop = 0, bar[2] = {0, 0}, slow[2] = {0, 0}, ent[2] = {0, 0}, lp[2] = {0, 0};
                        int hist[10] = {0};
                        for (size_t w = 0; w < nwaves; ++w) {
                            const double L = (double)h[w * 4], B = (double)h[w * 4 + 1], S = (double)h[w * 4 + 2], E = (double)h[w * 4 + 3];
                            if (L <= 0) continue;
                            const int half = (w & 7) >= 4;
                            loop += L;
                            lp[half] += L;
                            bar[half] += B;
                            slow[half] += S;
                            ent[half] += E;
                            int b = (int)(10.0 * B / L);
                            ++hist[b < 0 ? 0 : b > 9 ? 9 : b];
                        }
                        std::fprintf(stderr, "[wide stamps] waves %zu  loop cycles/wave %.0f | at the tile's wait + barrier: waves 0-3 %.1f %%, waves 4-7 %.1f %% | in the append path: "
                                             "%.1f %% / %.1f %% (%.0f / %.0f entries per wave, %.0f cycles each) | waves by barrier share, deciles:",
                                     nwaves, loop / nwaves, 100 * bar[0] / lp[0], 100 * bar[1] / lp[1], 100 * slow[0] / lp[0], 100 * slow[1] / lp[1],
                                     ent[0] / (nwaves / 2), ent[1] / (nwaves / 2), (slow[0] + slow[1]) / std::max(1.0, ent[0] + ent[1]));
                        for (int b = 0; b < 10; ++b) std::fprintf(stderr, " %d", hist[b]);
                        std::fprintf(stderr, "\n");
                        return hipSuccess;
                    }
                    switch (dbg) {
#define FSGPU_SK(D) case D: return launch_wide_t<384, EB, QT, 3, O, D>(args, grid, stream, occupancy);
                        FSGPU_SK(16) FSGPU_SK(17) FSGPU_SK(18) FSGPU_SK(20) FSGPU_SK(24) FSGPU_SK(31) FSGPU_SK(33) FSGPU_SK(35) FSGPU_SK(36)
#undef FSGPU_SK
                        default: break;
                    }
                }
            }
#endif
#ifdef FSGPU_LAB_ASYM   // lab: 5 + 3 query tiles per SIMD pair, 64-row tiles in six slots, no 128-row form (the five-tile waves run the chunk loop)
            if constexpr (EB == 1 && QT == 4 && MODE == 0) { if (!occupancy) return launch_wide_asym<384, 1, FSGPU_LAB_ASYM, 8 - FSGPU_LAB_ASYM, 6, O & ~(kOptBig | kOptFlags)>(args, grid, stream); }
#endif
            return launch_wide_pick<384, EB, QT, 6, O, MODE>(args, grid, stream, occupancy);   // 6 x 24 KB (3 x 48 KB)
        }
        case 128: if constexpr (EB != 1 || (MODE == 7 && !wide_split_ok(256, 1, QT, O, 7))) return hipErrorInvalidValue;
                  else return launch_wide_pick<256, EB, QT, 8, O, MODE>(args, grid, stream, occupancy);   // 8 x 16 KB (3 x 32 KB)
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

#ifdef FSGPU_LAB_BITMAP_CHECK
extern "C" int fsgpu_lab_bitmap_debug(unsigned long long* out, int cap_records) {
    unsigned int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_bm_dbg_n), 4) != hipSuccess) return -1;
    const int take = (int)(n < 64u ? n : 64u) < cap_records ? (int)(n < 64u ? n : 64u) : cap_records;
    if (take > 0 && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bm_dbg), (size_t)take * 64) != hipSuccess) return -1;
    return (int)n;
}
#endif

bool scan_wide_supported(int dim, int elem_bytes) {
    const int rowb = dim * elem_bytes;   // (the shapes launch_wide_d builds)
    return rowb == 768 || rowb == 512 || (elem_bytes == 1 && (rowb == 384 || rowb == 256));
}

// int8 rows whose shape runs the query-tile-split loop (the group-maxima sample stage is built on it)
bool scan_wide_group_maxima_supported(int dim, int query_tiles) {
    return (dim == 384 || dim == 256) && query_tiles >= 2 && query_tiles <= 5 &&
           wide_split_ok(dim, 1, query_tiles, kWideOptDefault, 7);
}

// largest query_tiles a row length admits (registers: see launch_wide_d), capped at 5
int scan_wide_max_query_tiles(int dim, int elem_bytes) {
    const int per_tile = dim * elem_bytes / 64 * 4;
    const int fit = per_tile > 0 ? 144 / per_tile : 0;
    if (elem_bytes != 1) return fit > 3 ? 3 : fit;   // f16 rows: 2 and 3 are built
    return fit > 5 ? 5 : fit;   // (6 tiles of int8 rows of 384 bytes compile to 256 registers + 36 spilled)
}

// query_tiles = QT (2: 256 queries per pass, 3: 384, 4: 512, 5: 640 — as many as scan_wide_max_query_tiles allows)
hipError_t launch_scan_wide(const MfmaScanArgs& args, int query_tiles, int grid, hipStream_t stream, int* occupancy) {
    const int eb = args.elem_bytes == 1 ? 1 : 2;
    if (!scan_wide_supported((int)args.dim, eb) || query_tiles > scan_wide_max_query_tiles((int)args.dim, eb)) return hipErrorInvalidValue;
    const bool sample = args.group_count != 0;   // a sample stage: 64-row groups group_stride apart
    if (sample && args.stage == 3) {             // ... that keeps group maxima instead of thresholded lists (int8 rows, split-loop shapes)
        if (eb != 1 || !scan_wide_group_maxima_supported((int)args.dim, query_tiles)) return hipErrorInvalidValue;
        if (query_tiles == 2) return launch_wide_d<1, 2, 7>(args, grid, stream, occupancy);
        if (query_tiles == 3) return launch_wide_d<1, 3, 7>(args, grid, stream, occupancy);
        if (query_tiles == 4) return launch_wide_d<1, 4, 7>(args, grid, stream, occupancy);
        if (query_tiles == 5) return launch_wide_d<1, 5, 7>(args, grid, stream, occupancy);
        return hipErrorInvalidValue;
    }
    if (eb == 2) {
        if (query_tiles == 2) return sample ? launch_wide_d<2, 2, 3>(args, grid, stream, occupancy) : launch_wide_d<2, 2>(args, grid, stream, occupancy);
        if (query_tiles == 3) return sample ? launch_wide_d<2, 3, 3>(args, grid, stream, occupancy) : launch_wide_d<2, 3>(args, grid, stream, occupancy);
    } else {
        if (query_tiles == 2) return sample ? launch_wide_d<1, 2, 3>(args, grid, stream, occupancy) : launch_wide_d<1, 2>(args, grid, stream, occupancy);
        if (query_tiles == 3) return sample ? launch_wide_d<1, 3, 3>(args, grid, stream, occupancy) : launch_wide_d<1, 3>(args, grid, stream, occupancy);
        if (query_tiles == 4) return sample ? launch_wide_d<1, 4, 3>(args, grid, stream, occupancy) : launch_wide_d<1, 4>(args, grid, stream, occupancy);
        if (query_tiles == 5) return sample ? launch_wide_d<1, 5, 3>(args, grid, stream, occupancy) : launch_wide_d<1, 5>(args, grid, stream, occupancy);
    }
    return hipErrorInvalidValue;
}

}  // namespace fsgpu
