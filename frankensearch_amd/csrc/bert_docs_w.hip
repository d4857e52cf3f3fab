// bert_docs_w.hip — the WHOLE MiniLM-L6 forward of a batch of short texts in ONE launch.
//
// Model::embed_forward (crates/frankensearch-rerank/src/native.rs:1142-1236: embedding gather + LayerNorm, six
// encoder_layer_raw (native.rs:587-626), mean over each text's tokens, L2) couples rows only inside a text (the attention).
// When every text of a batch is at most 32 tokens long a 32-row block can therefore own WHOLE texts, and nothing of the
// forward crosses a block: the six layers chain inside one kernel with no grid-wide dependency — the 20 launches of the
// batch path (bert_gemm_w.hip: QKV, attention, post-attention per layer + embedding + pooling) become one, and the residual
// stream never leaves the CU (f32 in the registers of the lanes that own its columns, its f16 copy in LDS).
//
// The host packs consecutive texts greedily into row blocks (blk_tok / blk_doc: first token / first text of each block).
// A 512-thread block runs, per layer,
//   QKV   x tile (LDS, f16) x Wqkv (fragment order, streamed through registers) -> Q, K row-major and V transposed, f16, in LDS
//   ATT   24 work items (head, 16-query tile) over the 8 waves: S^T = K Q^T (one MFMA k-step: heads are 32 wide), softmax over
//         the keys of the query's own text (the block mask of the packing), O^T = V^T P^T with the exponentials as the B
//         operand straight from registers; the context overwrites Q in place
//   AO    x1 = LayerNorm(x + ctx Wao^T + b) — the f32 rows stay in registers (the FFN's residual), f16 copy -> x tile
//   FFN   GELU(x1 W1^T + b1) -> 32 x 1536 f16 tile in LDS -> W2, + b2 + x1, LayerNorm -> x (registers) and the x tile
// with the phases of bert_ffn_w_kernel (same weight rings, same epilogues).  LDS: 2.3 KB of statistics + 25.6 KB x tile +
// 99.3 KB shared by {Q, K, V^T | context | intermediate tile | f32 rows of the ends}: 127 KB, one block per CU.
// Same arithmetic class as the batch path (f16 x f16 -> f32 MFMA linears, f32 bias / GELU / residual / LayerNorm / softmax
// statistics); tests/test_gpu_bert.py holds it to the same tolerance against the f32 oracle.
// MiniLM-L6 shape only (hidden 384, FFN 1536, 12 heads of 32); any other model or a text longer than 32 tokens takes the
// batch path.
#include "device_util.hpp"
#include "kernels.hpp"

namespace fsgpu {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

#ifndef DOCS_ATT_UNROLL
#define DOCS_ATT_UNROLL _Pragma("unroll 1")
#endif
// a weight fragment: 16 bytes per lane, each used once per block.  (The non-temporal form of the load was measured: 0.47 instead of
// 0.31 ms per 256 queries — the fragments of the ~200 blocks that stream the same weights no longer meet in the L2s.)
#define DOCS_WLOAD(p) (*(p))
// Shapes of the weight rings (measured at 256 queries, profiles/r03/encoder_one_launch.txt): FFN-up chunks of 2 tiles (3: +6 %),
// an FFN-down ring 8 k-steps deep (6: +1 %, 12: +2 %), QKV chunks of 3 tiles (1: +2 %), 8 of the 12 attention-output k-steps requested
// in front of the attention (6 / 10 / 12: +0.5 / +1.5 / +6 % — with all twelve the attention spills), the attention's three items
// per wave not unrolled (unrolled: spills).
#ifndef DOCS_UCH
#define DOCS_UCH 2
#endif
#ifndef DOCS_R2
#define DOCS_R2 8
#endif
#ifndef DOCS_AO_PF
#define DOCS_AO_PF 8
#endif
#ifndef DOCS_PF_UP
#define DOCS_PF_UP 1
#endif

#ifdef FSGPU_EXPERIMENTS
// lab build: block 0 / thread 0 records the shader clock at every phase boundary (BertDocsArgs::stamps, null = off)
#define DOCS_STAMP(slot)                                                                                     \
    do {                                                                                                      \
        if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[(slot)] = (unsigned long long)clock64(); \
    } while (0)
#else
#define DOCS_STAMP(slot) \
    do {                 \
    } while (0)
#endif

namespace {

constexpr int DH = 384;            // hidden
constexpr int DI = 1536;           // FFN width
constexpr int DBM = 32;            // rows of a block
constexpr int DNW = 8;             // waves
constexpr int DKS = DH / 32;       // k-steps over the hidden dimension
constexpr int DNT = 3;             // 16-column tiles of a hidden-wide output per wave
constexpr int DHP = DH + 16;       // halves per row of an f16 [32][384] tile
constexpr int DXP = DH + 4;        // floats per row of the f32 [32][384] tile
constexpr int DIP = DI + 16;       // halves per row of the intermediate tile
constexpr int DVP = 40;            // halves per row of V^T ([384][32 keys]) and of a wave's P tile ([16][32 keys])
constexpr size_t kDocsBig = (size_t)DBM * DIP * 2;                       // 99,328 bytes
constexpr int DOFF = 95;           // text boundaries of a block kept in LDS (more texts than that — empty ones — are read in place)
constexpr size_t kDocsLds = 2048 + 512 + (size_t)DBM * DHP * 2 + kDocsBig;

// Block barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL load (s_waitcnt vmcnt(0)),
// i.e. for the weight fragments the next phase requested ahead of it — the prefetches below would end at the first barrier.
// The threads of this kernel share nothing through global memory; the compiler keeps its own vmcnt waits in front of each use.
__device__ __forceinline__ void d_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float d_row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));
    return v;
}
// GELU(x) = max(x, 0) - |x| 2^(q(|x| / sqrt 2) - 1), q the degree-5 fit of log2(erfc) — gelu_as_w of bert_gemm_w.hip (see there;
// scripts/r05/fit_erf.py: 1.4e-6 from the reference's Abramowitz-Stegun 7.1.26 form, native.rs:190-200, before the rounding to f16)
__device__ __forceinline__ float d_gelu(float x) {
    const float az = fabsf(x) * 0.70710678118654752440f;
    float p = fmaf(az, -0.00294418f, 0.02959011f);
    p = fmaf(p, az, -0.14866571f);
    p = fmaf(p, az, -0.91850934f);
    p = fmaf(p, az, -1.62788901f);
    const float e_half = __builtin_amdgcn_exp2f(fmaf(p, az, -1.0f));
    return fmaxf(x, 0.0f) - fabsf(x) * e_half;
}

// v = LayerNorm(v) over the 384 columns of each row (add_ln_raw's normalisation, native.rs:560-578; two passes: mean, then
// the centred variance), the f16 copy into the x tile.  Lane: rows i * 16 + fr, columns wave * 48 + j * 16 + cq .. + 3.
// Every thread of the block calls this: three barriers (the two statistics have a buffer each), the last one behind the x
// tile's stores; the gains and offsets are requested before the first, so their round trip runs underneath the statistics.
__device__ __forceinline__ void d_layer_norm(f32x4 (&v)[2][DNT], const float* __restrict__ lnw, const float* __restrict__ lnb,
                                             float eps, float* red, _Float16* Xh, int wave, int lane) {
    const int fr = lane & 15, cq = (lane >> 4) * 4;
    float* red2 = red + DBM * DNW;
    f32x4 g[DNT], b[DNT];
#pragma unroll
    for (int j = 0; j < DNT; ++j) {
        const int col = wave * 16 * DNT + j * 16 + cq;
        g[j] = *reinterpret_cast<const f32x4*>(lnw + col);
        b[j] = *reinterpret_cast<const f32x4*>(lnb + col);
    }
    float ps[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < DNT; ++j) s += (v[i][j][0] + v[i][j][1]) + (v[i][j][2] + v[i][j][3]);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        ps[i] = s;
    }
    if (lane < 16) {
        red[(0 * 16 + lane) * DNW + wave] = ps[0];
        red[(1 * 16 + lane) * DNW + wave] = ps[1];
    }
    d_barrier();
    float mu[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float* p = red + (i * 16 + fr) * DNW;
        mu[i] = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) / (float)DH;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float qs = 0.f;
#pragma unroll
        for (int j = 0; j < DNT; ++j) {
            const f32x4 d = v[i][j] - mu[i];
            qs += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        qs += __shfl_xor(qs, 16);
        qs += __shfl_xor(qs, 32);
        if (lane < 16) red2[(i * 16 + lane) * DNW + wave] = qs;
    }
    d_barrier();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float* p = red2 + (i * 16 + fr) * DNW;
        const float var = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) / (float)DH;
        const float inv = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int j = 0; j < DNT; ++j) {
            const int col = wave * 16 * DNT + j * 16 + cq;
            const f32x4 y = (v[i][j] - mu[i]) * inv * g[j] + b[j];
            v[i][j] = y;
            half4 h;
            h[0] = (_Float16)y[0];
            h[1] = (_Float16)y[1];
            h[2] = (_Float16)y[2];
            h[3] = (_Float16)y[3];
            *reinterpret_cast<half4*>(&Xh[(i * 16 + fr) * DHP + col]) = h;
        }
    }
    d_barrier();   // the x tile is complete (and every read of both statistics buffers lies two barriers behind their next writes)
}

#ifndef DOCS_QCH
#define DOCS_QCH 3
#endif
constexpr int QTPW = 9, QCH = DOCS_QCH, QNCH = QTPW / QCH;   // QKV: 72 column tiles of 16, 9 per wave, three at a time

// the first chunk of a wave's QKV weight fragments; fragment (tile t, k-step ks) at wp[(t * 12 + ks) * 64 + lane]
__device__ __forceinline__ void d_qkv_request(const void* qkv_wp, half8 (&r)[DKS][QCH], int wave, int lane) {
    const half8* w = static_cast<const half8*>(qkv_wp) + (size_t)(wave * QTPW) * DKS * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < DKS; ++ks)
#pragma unroll
        for (int j = 0; j < QCH; ++j) r[ks][j] = DOCS_WLOAD(w + ((j * DKS + ks) * 64));
}

// x tile (LDS) x Wqkv -> Q, K row-major and V transposed, f16, in LDS; ends with a block barrier
__device__ __forceinline__ void d_qkv_phase(const void* qkv_wp, const float* __restrict__ qkv_b, half8 (&r)[DKS][QCH], const _Float16* Xh,
                                            _Float16* Qs, _Float16* Ks, _Float16* Vt, int wave, int lane) {
    const int fr = lane & 15, q = lane >> 4, cq = q * 4;
    const half8* w = static_cast<const half8*>(qkv_wp) + (size_t)(wave * QTPW) * DKS * 64 + lane;
#pragma unroll 1
    for (int c = 0; c < QNCH; ++c) {
        f32x4 acc[2][QCH];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < QCH; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const half8* wn = w + (size_t)(QCH * (c + 1)) * DKS * 64;
        f32x4 bv[QCH];   // requested here: the round trip runs underneath the chunk's k-loop
#pragma unroll
        for (int j = 0; j < QCH; ++j) bv[j] = *reinterpret_cast<const f32x4*>(qkv_b + (wave * QTPW + QCH * c + j) * 16 + cq);
#pragma unroll
        for (int ks = 0; ks < DKS; ++ks) {
            half8 af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const half8*>(&Xh[(i * 16 + fr) * DHP + ks * 32 + q * 8]);
#pragma unroll
            for (int j = 0; j < QCH; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r[ks][j], af[i], acc[i][j], 0, 0, 0);
            if (c + 1 < QNCH)
#pragma unroll
                for (int j = 0; j < QCH; ++j) r[ks][j] = DOCS_WLOAD(wn + ((j * DKS + ks) * 64));
        }
#pragma unroll
        for (int j = 0; j < QCH; ++j) {
            const int tile = wave * QTPW + QCH * c + j;    // wave-uniform: 0..23 Q, 24..47 K, 48..71 V (native.rs:1500-1540)
            const int col = tile * 16 + cq;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4 y = acc[i][j] + bv[j];
                const int row = i * 16 + fr;
                half4 h;
                h[0] = (_Float16)y[0]; h[1] = (_Float16)y[1]; h[2] = (_Float16)y[2]; h[3] = (_Float16)y[3];
                if (tile < 24) *reinterpret_cast<half4*>(&Qs[row * DHP + col]) = h;
                else if (tile < 48) *reinterpret_cast<half4*>(&Ks[row * DHP + (col - DH)]) = h;
                else {
                    // V transposed (the PV product wants the keys contiguous per dimension)
                    _Float16* vp = Vt + (col - 2 * DH) * DVP + row;
                    vp[0] = h[0];
                    vp[DVP] = h[1];
                    vp[2 * DVP] = h[2];
                    vp[3 * DVP] = h[3];
                }
            }
        }
    }
    d_barrier();
}

}  // namespace

__global__ __launch_bounds__(512) void bert_docs_w_kernel(BertDocsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char docs_smem[];
    float* red = reinterpret_cast<float*>(docs_smem);                                  // 2 x [32][8]: row sums, centred squares
    int* s_doc = reinterpret_cast<int*>(docs_smem + 2048);                             // [32] text of each row (-1: padding row)
    uint32_t* s_off = reinterpret_cast<uint32_t*>(s_doc + DBM);                        // [DOFF + 1] the block's text boundaries (pooling)
    _Float16* Xh = reinterpret_cast<_Float16*>(docs_smem + 2560);                      // [32][DHP]: the x tile
    unsigned char* big = docs_smem + 2560 + (size_t)DBM * DHP * 2;
    _Float16* Is = reinterpret_cast<_Float16*>(big);                                   // [32][DIP]
    float* Xs = reinterpret_cast<float*>(big);                                         // [32][DXP] (the ends)
    _Float16* Qs = reinterpret_cast<_Float16*>(big);                                   // [32][DHP], then the context
    _Float16* Ks = Qs + DBM * DHP;                                                     // [32][DHP]
    _Float16* Vt = Ks + DBM * DHP;                                                     // [384][DVP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, q = lane >> 4, cq = q * 4;
    const uint32_t blk = blockIdx.x;
    DOCS_STAMP(0);
    half8 rq0[DKS][QCH];   // the first layer's first QKV chunk: in flight underneath the embedding stage
    d_qkv_request(a.layers[0].qkv_wp, rq0, wave, lane);
    // One round trip to the call's inputs (pinned host memory for query-sized calls): the host laid out each block's 32 rows
    // (token id, -1 = padding; rows of the block | position | text inside the block), so nothing here waits on a second read.
    const int r_emb = tid >> 4;                                         // the row this thread's 16-lane group embeds
    const int my_id = a.row_id[(size_t)blk * DBM + r_emb];
    const uint32_t my_meta = a.row_meta[(size_t)blk * DBM + r_emb];
    const uint32_t t0 = a.blk_tok[blk];
    const uint32_t d0 = a.blk_doc[blk], d1 = a.blk_doc[blk + 1];
    const int nrows = (int)((my_meta >> 16) & 0xffu);
    if (nrows <= 0) {   // only empty texts (or none): zeros (native.rs:1146-1148)
        for (uint32_t i = tid; i < (d1 - d0) * (uint32_t)DH; i += 512) a.out[(size_t)d0 * DH + i] = 0.f;
        return;
    }
    // the texts' boundaries, for the pooling at the far end (requested now: their round trip is long over by then)
    const uint32_t ndocs = d1 - d0;
    if (tid >= 64 && tid - 64 <= ndocs && tid - 64 <= (uint32_t)DOFF) s_off[tid - 64] = a.offsets[d0 + (tid - 64)];
    if ((lane & 15) == 0) s_doc[r_emb] = my_id >= 0 ? (int)(my_meta & 0xffu) : -1;
    // ---- embedding gather + LayerNorm (native.rs:1176-1192): sixteen lanes per row, as bert_embed_ln16_kernel ----
    {
        const int r = r_emb, li = tid & 15;
        const bool live = my_id >= 0;
        const float4* wr = reinterpret_cast<const float4*>(a.word + (size_t)(live ? my_id : 0) * DH);
        const float4* pr = reinterpret_cast<const float4*>(a.pos + (size_t)(live ? (my_meta >> 8) & 0xffu : 0u) * DH);   // positions restart at 0 per text (native.rs:1159-1167)
        const float4* tr = reinterpret_cast<const float4*>(a.type0);
        float4 lg[6], lb[6];   // requested with the rows: one round trip for everything the stage reads
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            lg[j] = reinterpret_cast<const float4*>(a.emb_lnw)[li + 16 * j];
            lb[j] = reinterpret_cast<const float4*>(a.emb_lnb)[li + 16 * j];
        }
        float4 v[6];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float4 x = wr[li + 16 * j], y = pr[li + 16 * j], z = tr[li + 16 * j];
            v[j] = make_float4((x.x + y.x) + z.x, (x.y + y.y) + z.y, (x.z + y.z) + z.z, (x.w + y.w) + z.w);
            s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
        const float mean = d_row16_sum(s) / (float)DH;
        float qq = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float dx = v[j].x - mean, dy = v[j].y - mean, dz = v[j].z - mean, dw = v[j].w - mean;
            qq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        const float inv = 1.0f / sqrtf(d_row16_sum(qq) / (float)DH + a.eps);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float4 g = lg[j], b = lb[j];
            float4 y = make_float4((v[j].x - mean) * inv * g.x + b.x, (v[j].y - mean) * inv * g.y + b.y,
                                   (v[j].z - mean) * inv * g.z + b.z, (v[j].w - mean) * inv * g.w + b.w);
            if (!live) y = make_float4(0.f, 0.f, 0.f, 0.f);   // padding rows: zeros (finite everywhere downstream, masked in the attention)
            const int c0 = (li + 16 * j) * 4;
            *reinterpret_cast<float4*>(&Xs[r * DXP + c0]) = y;
            half4 h;
            h[0] = (_Float16)y.x; h[1] = (_Float16)y.y; h[2] = (_Float16)y.z; h[3] = (_Float16)y.w;
            *reinterpret_cast<half4*>(&Xh[r * DHP + c0]) = h;
        }
    }
    d_barrier();
    // the residual stream: this lane's elements (rows i * 16 + fr, columns wave * 48 + j * 16 + cq .. + 3), in registers for all layers
    f32x4 xres[2][DNT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < DNT; ++j) xres[i][j] = *reinterpret_cast<const f32x4*>(&Xs[(i * 16 + fr) * DXP + wave * 16 * DNT + j * 16 + cq]);
    d_barrier();   // the f32 tile's space becomes Q, K, V^T
    // the attention's block mask, in registers: the text of this lane's keys (cj * 16 + cq + r) and of its query rows (rt * 16 + fr)
    int kdoc[2][4], qdoc[2];
#pragma unroll
    for (int cj = 0; cj < 2; ++cj) {
#pragma unroll
        for (int r = 0; r < 4; ++r) kdoc[cj][r] = s_doc[cj * 16 + cq + r];
        qdoc[cj] = s_doc[cj * 16 + fr];
    }

    // Every weight phase starts on fragments requested BEFORE the phase in front of it that streams nothing (attention, the
    // LayerNorms): the round trip of a phase's first ring — and the whole attention-output slice — runs underneath that phase.
    // (The QKV projection of layer l + 1 therefore sits at the END of layer l's iteration, its first chunk requested in front of
    // the LayerNorm that closes the layer.)
    DOCS_STAMP(1);
    d_qkv_phase(a.layers[0].qkv_wp, a.layers[0].qkv_b, rq0, Xh, Qs, Ks, Vt, wave, lane);
    for (int layer = 0; layer < a.nlayers; ++layer) {
        const BertDocsLayer L = a.layers[layer];
        DOCS_STAMP(2 + 8 * layer);
        half8 r0[DKS][DNT];   // the wave's whole slice of the attention-output weights, in flight underneath the attention
        // (the first DOCS_AO_PF k-steps: the rest is requested when the projection starts and lands underneath its first MFMAs —
        // all twelve in front of the attention left it too few registers and it spilled)
        const half8* w0 = static_cast<const half8*>(L.ao_wp) + (size_t)(wave * DNT) * DKS * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < DOCS_AO_PF; ++ks)
#pragma unroll
            for (int j = 0; j < DNT; ++j) r0[ks][j] = DOCS_WLOAD(w0 + ((j * DKS + ks) * 64));
        // ---- attention (fused_attention / fast_softmax_inplace, native.rs:82-163,366-432; the scale goes inside the exponential) ----
        // S^T = K Q^T: a lane holds the scores of ONE query (row rt * 16 + fr) against the keys cj * 16 + cq + r — the row maximum
        // and sum are in-lane plus two xor-shuffles — and its exponentials, rounded to f16, ARE the B fragment of O^T = V^T P^T
        // (the reduction index of an MFMA may be permuted as long as both operands agree: the V^T fragment is read as the two
        // 8-byte runs of the same keys).  No P tile in LDS; the output layout has the lane's query again: four consecutive
        // dimensions per store.
        {
DOCS_ATT_UNROLL
            for (int it = 0; it < 24 / DNW; ++it) {
                const int item = wave + it * DNW;
                const int head = item >> 1, rt = item & 1;
                const int fk = q * 8;
                const half8 qf = *reinterpret_cast<const half8*>(&Qs[(rt * 16 + fr) * DHP + head * 32 + fk]);
                f32x4 sc[2];
#pragma unroll
                for (int cj = 0; cj < 2; ++cj) {
                    const half8 kf = *reinterpret_cast<const half8*>(&Ks[(cj * 16 + fr) * DHP + head * 32 + fk]);
                    sc[cj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                }
                const int qd = qdoc[rt];
                float v[2][4], m = -INFINITY;
#pragma unroll
                for (int cj = 0; cj < 2; ++cj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[cj][r] = (qd >= 0 && kdoc[cj][r] == qd) ? sc[cj][r] : -INFINITY;
                        m = fmaxf(m, v[cj][r]);
                    }
                m = fmaxf(m, __shfl_xor(m, 16));
                m = fmaxf(m, __shfl_xor(m, 32));
                float sum = 0.f;
                half8 pf;
#pragma unroll
                for (int cj = 0; cj < 2; ++cj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = v[cj][r] == -INFINITY ? 0.f : __expf((v[cj][r] - m) * a.attn_scale);
                        sum += e;
                        pf[cj * 4 + r] = (_Float16)e;
                    }
                sum += __shfl_xor(sum, 16);
                sum += __shfl_xor(sum, 32);
                const float inv = sum > 0.f ? __builtin_amdgcn_rcpf(sum) : 0.f;   // (1 ulp: the context is rounded to f16 next)
#pragma unroll
                for (int dj = 0; dj < 2; ++dj) {
                    const _Float16* vrow = Vt + (head * 32 + dj * 16 + fr) * DVP + cq;
                    const half4 v0 = *reinterpret_cast<const half4*>(vrow), v1 = *reinterpret_cast<const half4*>(vrow + 16);
                    half8 vf;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        vf[e] = v0[e];
                        vf[4 + e] = v1[e];
                    }
                    const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    // O[query rt * 16 + fr][head * 32 + dj * 16 + cq .. + 3]: the context takes the place of this item's queries
                    half4 h;
                    h[0] = (_Float16)(o[0] * inv);
                    h[1] = (_Float16)(o[1] * inv);
                    h[2] = (_Float16)(o[2] * inv);
                    h[3] = (_Float16)(o[3] * inv);
                    *reinterpret_cast<half4*>(&Qs[(rt * 16 + fr) * DHP + head * 32 + dj * 16 + cq]) = h;
                }
            }
        }
        d_barrier();
        DOCS_STAMP(3 + 8 * layer);
        // ---- attention output projection + residual + LayerNorm (add_ln_raw, native.rs:560-578) ----
        f32x4 x1[2][DNT];
        {
#pragma unroll
            for (int ks = DOCS_AO_PF; ks < DKS; ++ks)
#pragma unroll
                for (int j = 0; j < DNT; ++j) r0[ks][j] = DOCS_WLOAD(w0 + ((j * DKS + ks) * 64));
            f32x4 b0[DNT];
#pragma unroll
            for (int j = 0; j < DNT; ++j) b0[j] = *reinterpret_cast<const f32x4*>(L.ao_b + wave * 16 * DNT + j * 16 + cq);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < DNT; ++j) x1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < DKS; ++ks) {
                half8 af[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const half8*>(&Qs[(i * 16 + fr) * DHP + ks * 32 + q * 8]);
#pragma unroll
                for (int j = 0; j < DNT; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) x1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r0[ks][j], af[i], x1[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < DNT; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) x1[i][j] = (x1[i][j] + b0[j]) + xres[i][j];
        }
        DOCS_STAMP(4 + 8 * layer);
        constexpr int UCH = DOCS_UCH, UTPW = DI / 16 / DNW, UNCH = UTPW / UCH;
        const half8* w1 = static_cast<const half8*>(L.i_wp) + (size_t)(wave * UTPW) * DKS * 64 + lane;
        half8 r1[DKS][UCH];   // the up-projection's first chunk, in flight underneath the LayerNorm
        if (DOCS_PF_UP) {
#pragma unroll
            for (int ks = 0; ks < DKS; ++ks)
#pragma unroll
                for (int j = 0; j < UCH; ++j) r1[ks][j] = DOCS_WLOAD(w1 + ((j * DKS + ks) * 64));
        }
        d_layer_norm(x1, L.ln1_w, L.ln1_b, a.eps, red, Xh, wave, lane);   // (its first barrier also ends the reads of the context)
        DOCS_STAMP(5 + 8 * layer);
        // ---- FFN up + GELU -> the intermediate tile (encoder_layer_raw, native.rs:606-626) ----
        {
            constexpr int CH = UCH, TPW = UTPW, NCH = UNCH;
            if (!DOCS_PF_UP) {
#pragma unroll
                for (int ks = 0; ks < DKS; ++ks)
#pragma unroll
                    for (int j = 0; j < CH; ++j) r1[ks][j] = DOCS_WLOAD(w1 + ((j * DKS + ks) * 64));
            }
#pragma unroll 1
            for (int c = 0; c < NCH; ++c) {
                f32x4 acc[2][CH];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < CH; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                const half8* wn = w1 + (size_t)(CH * (c + 1)) * DKS * 64;
                f32x4 bv[CH];   // requested here: the round trip runs underneath the chunk's k-loop
#pragma unroll
                for (int j = 0; j < CH; ++j) bv[j] = *reinterpret_cast<const f32x4*>(L.i_b + (wave * TPW + CH * c + j) * 16 + cq);
#pragma unroll
                for (int ks = 0; ks < DKS; ++ks) {
                    half8 af[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const half8*>(&Xh[(i * 16 + fr) * DHP + ks * 32 + q * 8]);
#pragma unroll
                    for (int j = 0; j < CH; ++j)
#pragma unroll
                        for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r1[ks][j], af[i], acc[i][j], 0, 0, 0);
                    if (c + 1 < NCH)
#pragma unroll
                        for (int j = 0; j < CH; ++j) r1[ks][j] = DOCS_WLOAD(wn + ((j * DKS + ks) * 64));
                }
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int col = (wave * TPW + CH * c + j) * 16 + cq;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const f32x4 y = acc[i][j] + bv[j];
                        half4 h;
                        h[0] = (_Float16)d_gelu(y[0]);
                        h[1] = (_Float16)d_gelu(y[1]);
                        h[2] = (_Float16)d_gelu(y[2]);
                        h[3] = (_Float16)d_gelu(y[3]);
                        *reinterpret_cast<half4*>(&Is[(i * 16 + fr) * DIP + col]) = h;
                    }
                }
            }
        }
        DOCS_STAMP(6 + 8 * layer);
        // ---- FFN down + residual + LayerNorm: W2 fragment (tile j, k-step ks) at w2[(j * 48 + ks) * 64] ----
        {
            constexpr int KS2 = DI / 32, R2 = DOCS_R2;
            const half8* w2 = static_cast<const half8*>(L.o_wp) + (size_t)(wave * DNT) * KS2 * 64 + lane;
            half8 r2[R2][DNT];
#pragma unroll
            for (int d = 0; d < R2; ++d)
#pragma unroll
                for (int j = 0; j < DNT; ++j) r2[d][j] = DOCS_WLOAD(w2 + (((size_t)j * KS2 + d) * 64));
            f32x4 b2[DNT];
#pragma unroll
            for (int j = 0; j < DNT; ++j) b2[j] = *reinterpret_cast<const f32x4*>(L.o_b + wave * 16 * DNT + j * 16 + cq);
            d_barrier();   // the intermediate tile is complete
            f32x4 acc[2][DNT];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < DNT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int ks0 = 0; ks0 < KS2; ks0 += R2) {
#pragma unroll
                for (int d = 0; d < R2; ++d) {
                    const int ks = ks0 + d;
                    half8 af[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const half8*>(&Is[(i * 16 + fr) * DIP + ks * 32 + q * 8]);
#pragma unroll
                    for (int j = 0; j < DNT; ++j)
#pragma unroll
                        for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r2[d][j], af[i], acc[i][j], 0, 0, 0);
                    if (ks + R2 < KS2)
#pragma unroll
                        for (int j = 0; j < DNT; ++j) r2[d][j] = DOCS_WLOAD(w2 + (((size_t)j * KS2 + ks + R2) * 64));
                }
            }
#pragma unroll
            for (int j = 0; j < DNT; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) xres[i][j] = (acc[i][j] + b2[j]) + x1[i][j];
        }
        DOCS_STAMP(7 + 8 * layer);
        // (the LayerNorm's first barrier ends every wave's reads of the intermediate tile: the next layer's Q, K, V^T go there)
        if (layer + 1 < a.nlayers) {
            const BertDocsLayer Ln = a.layers[layer + 1];
            half8 r[DKS][QCH];
            d_qkv_request(Ln.qkv_wp, r, wave, lane);   // in flight underneath the LayerNorm
            d_layer_norm(xres, L.ln2_w, L.ln2_b, a.eps, red, Xh, wave, lane);
            DOCS_STAMP(8 + 8 * layer);
            d_qkv_phase(Ln.qkv_wp, Ln.qkv_b, r, Xh, Qs, Ks, Vt, wave, lane);
        } else {
            d_layer_norm(xres, L.ln2_w, L.ln2_b, a.eps, red, Xh, wave, lane);
        }
    }
    DOCS_STAMP(2 + 8 * a.nlayers);
    // ---- mean over each text's tokens, L2 with the zero guard (native.rs:1209-1235; fastembed_embedder.rs:416-426) ----
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < DNT; ++j) *reinterpret_cast<f32x4*>(&Xs[(i * 16 + fr) * DXP + wave * 16 * DNT + j * 16 + cq]) = xres[i][j];
    d_barrier();
    // one text at a time, the whole block: thread t sums dimension t over the text's rows (consecutive LDS words), the squared norm
    // meets in LDS
    for (uint32_t j = 0; j < ndocs; ++j) {   // block-uniform
        const uint32_t o0 = j <= (uint32_t)DOFF ? s_off[j] : a.offsets[d0 + j];
        const uint32_t o1 = j + 1 <= (uint32_t)DOFF ? s_off[j + 1] : a.offsets[d0 + j + 1];
        const int r0 = (int)(o0 - t0), r1 = (int)(o1 - t0);
        const int n = r1 - r0;
        float val = 0.f;
        if (tid < DH && n > 0) {
            for (int r = r0; r < r1; ++r) val += Xs[r * DXP + tid];
            val *= 1.0f / (float)n;
        }
        float sq = val * val;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
        if (lane == 0) red[(j & 1) * DNW + wave] = sq;
        d_barrier();   // (two statistics slots alternate: the next text's writes cannot pass this text's reads)
        const float* p = red + (j & 1) * DNW;
        const float norm_sq = ((p[0] + p[1]) + (p[2] + p[3])) + (p[4] + p[5]);   // waves 6, 7 hold no dimension
        float scale = 0.f;
        if (__builtin_isfinite(norm_sq) && norm_sq > 1.1920929e-7f) scale = 1.0f / sqrtf(norm_sq);
        if (tid < DH) a.out[(size_t)(d0 + j) * DH + tid] = val * scale;
    }
    DOCS_STAMP(3 + 8 * a.nlayers);
}

bool bert_docs_w_supported(int hidden, int inter, int heads) { return hidden == DH && inter == DI && heads * 32 == DH; }

hipError_t launch_bert_docs_w(const BertDocsArgs& a, uint32_t nblocks, hipStream_t stream) {
    if (nblocks == 0) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bert_docs_w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)kDocsLds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bert_docs_w_kernel, dim3(nblocks), dim3(512), kDocsLds, stream, a);
    return hipGetLastError();
}

}  // namespace fsgpu
