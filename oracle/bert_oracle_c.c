/* bert_oracle_c.c -- f32 CPU restatement of the MiniLM-class embedder forward.  TEST INFRASTRUCTURE ONLY
 * (the encoder `cpu_baseline` leg of bench.py and tests/test_oracle_bert.py; never linked into the product).
 *
 * Follows `Model::embed_forward` of the reference's native backend
 *   crates/frankensearch-rerank/src/native.rs:1142-1236  embed_forward (gather + LN, L layers, mean pool, L2)
 *   native.rs:587-626    encoder_layer_raw  (fused QKV -> per-document attention -> out-proj -> add+LN -> FFN GELU -> add+LN)
 *   native.rs:366-432    fused_attention    (per head, no mask, softmax(scale * QK^T) V, head dim 32, scale 0.17677669)
 *   native.rs:82-147     softmax_row_fused  (exp((x - max) * scale) / sum)
 *   native.rs:190-200    gelu_scalar        (exact-form GELU, Abramowitz-Stegun 7.1.26 erf)
 *   native.rs:560-578    add_ln_raw         (LayerNorm(a + b), eps 1e-12)
 * with the adapter's zero guard on the final normalisation (crates/frankensearch-embed/src/fastembed_embedder.rs:416-426).
 * The reference's linears are int8 dynamic-quantised through frankentorch (not vendored): this is the f32 form of that
 * forward, the same arithmetic as oracle/bert_oracle.py (numpy), which tests/test_oracle_bert.py holds it against.
 * Documents are independent, so the thread pool splits the batch by document (the reference serialises callers on a mutex,
 * native_embedder.rs:40-50; its rayon pool works inside the GEMMs).  Dot products use 4 x 8-lane FMA accumulators. */
#include <immintrin.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "fs_oracle.h"

#define HEAD_DIM 32

static float dotf(const float *a, const float *b, int n) {
    __m256 s0 = _mm256_setzero_ps(), s1 = s0, s2 = s0, s3 = s0;
    int i = 0;
    for (; i + 32 <= n; i += 32) {
        s0 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i), s0);
        s1 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i + 8), _mm256_loadu_ps(b + i + 8), s1);
        s2 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i + 16), _mm256_loadu_ps(b + i + 16), s2);
        s3 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i + 24), _mm256_loadu_ps(b + i + 24), s3);
    }
    for (; i + 8 <= n; i += 8) s0 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i), s0);
    __m256 s = _mm256_add_ps(_mm256_add_ps(s0, s1), _mm256_add_ps(s2, s3));
    float t[8];
    _mm256_storeu_ps(t, s);
    float r = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    for (; i < n; ++i) r += a[i] * b[i];
    return r;
}

static float hsum8(__m256 s) {
    float t[8];
    _mm256_storeu_ps(t, s);
    return ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
}

/* four token rows against one weight row: the weight vector is loaded once per four dot products (K % 8 == 0) */
static void dotf4(const float *x, size_t ldx, const float *w, int K, float *o0, float *o1, float *o2, float *o3) {
    __m256 a0 = _mm256_setzero_ps(), a1 = a0, a2 = a0, a3 = a0;
    for (int i = 0; i < K; i += 8) {
        const __m256 wv = _mm256_loadu_ps(w + i);
        a0 = _mm256_fmadd_ps(_mm256_loadu_ps(x + i), wv, a0);
        a1 = _mm256_fmadd_ps(_mm256_loadu_ps(x + ldx + i), wv, a1);
        a2 = _mm256_fmadd_ps(_mm256_loadu_ps(x + 2 * ldx + i), wv, a2);
        a3 = _mm256_fmadd_ps(_mm256_loadu_ps(x + 3 * ldx + i), wv, a3);
    }
    *o0 = hsum8(a0); *o1 = hsum8(a1); *o2 = hsum8(a2); *o3 = hsum8(a3);
}

/* y[s, n] = x[s, :] . w[n, :] + b[n]   (HF layout: w is [N, K]); a weight row is read once per token block */
static void linear(const float *x, int S, int K, const float *w, const float *b, int N, float *y) {
    for (int n = 0; n < N; ++n) {
        const float *wr = w + (size_t)n * K;
        int s = 0;
        for (; s + 4 <= S; s += 4) {
            float o0, o1, o2, o3;
            dotf4(x + (size_t)s * K, (size_t)K, wr, K, &o0, &o1, &o2, &o3);
            y[(size_t)s * N + n] = o0 + b[n];
            y[(size_t)(s + 1) * N + n] = o1 + b[n];
            y[(size_t)(s + 2) * N + n] = o2 + b[n];
            y[(size_t)(s + 3) * N + n] = o3 + b[n];
        }
        for (; s < S; ++s) y[(size_t)s * N + n] = dotf(x + (size_t)s * K, wr, K) + b[n];
    }
}

static float gelu1(float x) { /* native.rs:190-200 */
    const float z = x * 0.70710678118654752440f;
    const float az = fabsf(z);
    const float t = 1.0f / (1.0f + 0.3275911f * az);
    const float poly = t * (0.2548296f + t * (-0.28449673f + t * (1.4214137f + t * (-1.453152f + t * 1.0614054f))));
    const float e = 1.0f - poly * expf(-(z * z));
    return 0.5f * x * (1.0f + copysignf(e, z));
}

/* x = LayerNorm(x + d) over H (d may be NULL), statistics in double as oracle/bert_oracle.py keeps them */
static void add_ln(float *x, const float *d, int S, int H, const float *w, const float *b, float eps) {
    for (int s = 0; s < S; ++s) {
        float *r = x + (size_t)s * H;
        double mean = 0.0, var = 0.0;
        for (int i = 0; i < H; ++i) {
            if (d) r[i] = r[i] + d[(size_t)s * H + i];
            mean += r[i];
        }
        mean /= H;
        for (int i = 0; i < H; ++i) var += ((double)r[i] - mean) * ((double)r[i] - mean);
        var /= H;
        const double inv = 1.0 / sqrt(var + (double)eps);
        for (int i = 0; i < H; ++i) r[i] = (float)((((double)r[i] - mean) * inv)) * w[i] + b[i];
    }
}

static void attention(const float *qkv, int S, int H, float scale, float *ctx, float *row) {
    const int NH = H / HEAD_DIM;
    for (int h = 0; h < NH; ++h)
        for (int i = 0; i < S; ++i) {
            const float *q = qkv + (size_t)i * 3 * H + h * HEAD_DIM;
            float m = -INFINITY;
            for (int j = 0; j < S; ++j) {
                row[j] = dotf(q, qkv + (size_t)j * 3 * H + H + h * HEAD_DIM, HEAD_DIM);
                if (row[j] > m) m = row[j];
            }
            float sum = 0.0f;
            for (int j = 0; j < S; ++j) {
                row[j] = expf((row[j] - m) * scale);
                sum += row[j];
            }
            const float inv = 1.0f / sum;
            float *o = ctx + (size_t)i * H + h * HEAD_DIM;
            for (int c = 0; c < HEAD_DIM; ++c) o[c] = 0.0f;
            for (int j = 0; j < S; ++j) {
                const float p = row[j] * inv;
                const float *v = qkv + (size_t)j * 3 * H + 2 * H + h * HEAD_DIM;
                for (int c = 0; c < HEAD_DIM; ++c) o[c] += p * v[c];
            }
        }
}

/* a block of consecutive documents d0 .. d1 as ONE token block (native.rs:1142-1236 keeps the whole batch flat the same way) */
static void forward_block(const fso_bert_weights *w, const int32_t *ids, const uint32_t *offsets, uint32_t d0, uint32_t d1,
                          float *out) {
    const int H = w->hidden, I = w->inter;
    const uint32_t t0 = offsets[d0];
    const int T = (int)(offsets[d1] - t0);
    memset(out + (size_t)d0 * H, 0, sizeof(float) * (size_t)(d1 - d0) * H);
    if (T <= 0) return;
    int smax = 0;
    for (uint32_t d = d0; d < d1; ++d)
        if ((int)(offsets[d + 1] - offsets[d]) > smax) smax = (int)(offsets[d + 1] - offsets[d]);
    float *x = malloc(sizeof(float) * (size_t)T * H), *qkv = malloc(sizeof(float) * (size_t)T * 3 * H);
    float *ctx = malloc(sizeof(float) * (size_t)T * H), *tmp = malloc(sizeof(float) * (size_t)T * H);
    float *mid = malloc(sizeof(float) * (size_t)T * I), *row = malloc(sizeof(float) * (size_t)smax);
    for (uint32_t d = d0; d < d1; ++d)
        for (uint32_t t = offsets[d]; t < offsets[d + 1]; ++t) {
            int32_t id = ids[t];
            if (id < 0 || id >= w->vocab) id = 0;
            const int pp = (int)(t - offsets[d]);
            const int p = pp < w->max_pos ? pp : w->max_pos - 1;
            float *xr = x + (size_t)(t - t0) * H;
            for (int i = 0; i < H; ++i) xr[i] = (w->word[(size_t)id * H + i] + w->pos[(size_t)p * H + i]) + w->type0[i];
        }
    add_ln(x, NULL, T, H, w->emb_ln_w, w->emb_ln_b, w->eps);
    for (int l = 0; l < w->layers; ++l) {
        const fso_bert_layer *L = &w->layer[l];
        linear(x, T, H, L->wqkv, L->bqkv, 3 * H, qkv);
        for (uint32_t d = d0; d < d1; ++d) {
            const int S = (int)(offsets[d + 1] - offsets[d]);
            if (S > 0) attention(qkv + (size_t)(offsets[d] - t0) * 3 * H, S, H, 0.17677669f, ctx + (size_t)(offsets[d] - t0) * H, row);
        }
        linear(ctx, T, H, L->wo, L->bo, H, tmp);
        add_ln(x, tmp, T, H, L->ln1_w, L->ln1_b, w->eps);
        linear(x, T, H, L->w1, L->b1, I, mid);
        for (size_t i = 0; i < (size_t)T * I; ++i) mid[i] = gelu1(mid[i]);
        linear(mid, T, I, L->w2, L->b2, H, tmp);
        add_ln(x, tmp, T, H, L->ln2_w, L->ln2_b, w->eps);
    }
    for (uint32_t d = d0; d < d1; ++d) {
        const int S = (int)(offsets[d + 1] - offsets[d]);
        if (S <= 0) continue;
        float *o = out + (size_t)d * H;
        for (int s = 0; s < S; ++s)
            for (int i = 0; i < H; ++i) o[i] += x[((size_t)(offsets[d] - t0) + s) * H + i];
        const float inv = 1.0f / (float)S;
        float nsq = 0.0f;
        for (int i = 0; i < H; ++i) {
            o[i] *= inv;
            nsq += o[i] * o[i];
        }
        if (isfinite(nsq) && nsq > 1.1920929e-7f) {
            const float sc = 1.0f / sqrtf(nsq);
            for (int i = 0; i < H; ++i) o[i] *= sc;
        } else {
            memset(o, 0, sizeof(float) * (size_t)H);
        }
    }
    free(x); free(qkv); free(ctx); free(tmp); free(mid); free(row);
}

typedef struct {
    const fso_bert_weights *w;
    const int32_t *ids;
    const uint32_t *offsets;
    uint32_t n_blocks;
    const uint32_t *block_start;   /* [n_blocks + 1] document index where each block starts */
    float *out;
    volatile uint32_t next;
} bert_job;

static void *bert_worker(void *arg) {
    bert_job *j = arg;
    for (;;) {
        const uint32_t b = __sync_fetch_and_add(&j->next, 1u);
        if (b >= j->n_blocks) break;
        forward_block(j->w, j->ids, j->offsets, j->block_start[b], j->block_start[b + 1], j->out);
    }
    return NULL;
}

int fso_bert_forward(const fso_bert_weights *w, const int32_t *ids, const uint32_t *offsets, uint32_t n_docs, int nthreads,
                     float *out) {
    if (!w || !offsets || !out || w->hidden % HEAD_DIM || w->hidden % 8 || w->inter % 8) return 2;
    if (n_docs == 0) return 0;
    if (nthreads > 256) nthreads = 256;
    if (nthreads < 1) nthreads = 1;
    /* blocks of consecutive documents, ~128 tokens each (a block's activations stay in the core's L2 while the weights stream
     * past once per block), sized so that every thread gets the same number of blocks */
    uint32_t total = offsets[n_docs] - offsets[0];
    const uint32_t per_round = 128u * (uint32_t)nthreads;
    const uint32_t rounds = (total + per_round - 1) / per_round;
    const uint32_t target = total / ((rounds ? rounds : 1) * (uint32_t)nthreads) + 1;   /* an even number of blocks per thread */
    uint32_t *starts = malloc(sizeof(uint32_t) * ((size_t)n_docs + 2));
    uint32_t nb = 0, acc = 0;
    starts[0] = 0;
    for (uint32_t d = 0; d < n_docs; ++d) {
        acc += offsets[d + 1] - offsets[d];
        if (acc >= target || d + 1 == n_docs) {
            starts[++nb] = d + 1;
            acc = 0;
        }
    }
    bert_job job = {w, ids, offsets, nb, starts, out, 0};
    if ((uint32_t)nthreads > nb) nthreads = (int)nb;
    pthread_t th[256];
    int started = 0;
    for (int t = 1; t < nthreads; ++t)
        if (pthread_create(&th[started], NULL, bert_worker, &job) == 0) ++started;
    bert_worker(&job);
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    free(starts);
    return 0;
}
