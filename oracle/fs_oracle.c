/*
 * fs_oracle.c — CPU restatement of frankensearch's f16 cosine scan + top-k, FSVI v1
 * format, f32->f16 encode and Model2Vec pool.  TEST INFRASTRUCTURE ONLY — see
 * fs_oracle.h for the usage rule and the parity-pin statement.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; the AVX2/F16C variant lives in
 * fs_oracle_avx2.c so only that translation unit is compiled with -mavx2 -mf16c).
 * Every function cites the reference file:line (relative to /root/reference/) it follows.
 */
#include "fs_oracle.h"

#include <errno.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* f16 <-> f32                                                                */
/* ------------------------------------------------------------------------- */

static inline uint32_t f32_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
static inline float bits_f32(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* Exact IEEE binary16 -> binary32 widening.  The reference reaches the same values with
 * the "magic multiply" trick (crates/frankensearch-index/src/simd.rs:63-82) and with
 * half::f16::to_f32; both are exact for finite/zero/subnormal inputs and map inf->inf,
 * nan->nan keeping the sign.  Here the classic field-by-field expansion is used. */
float fso_f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t out;
    if (exp == 0) {
        if (man == 0) {
            out = sign;
        } else {
            /* subnormal: normalise */
            int shift = 0;
            while ((man & 0x400u) == 0) {
                man <<= 1;
                ++shift;
            }
            man &= 0x3ffu;
            out = sign | ((uint32_t)(127 - 15 - shift + 1) << 23) | (man << 13);
        }
    } else if (exp == 0x1f) {
        out = sign | 0x7f800000u | (man << 13);
        if (man != 0) out |= 0x00400000u; /* quiet */
    } else {
        out = sign | ((exp + (127 - 15)) << 23) | (man << 13);
    }
    return bits_f32(out);
}

/* f32 -> f16 round-to-nearest-even == half::f16::from_f32 == vcvtps2ph RNE
 * (crates/frankensearch-index/src/simd.rs:2245-2305). */
uint16_t fso_f32_to_f16(float f) {
    uint32_t x = f32_bits(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t exp = (x >> 23) & 0xffu;
    uint32_t man = x & 0x007fffffu;
    if (exp == 0xff) {
        if (man == 0) return (uint16_t)(sign | 0x7c00u);
        /* NaN: keep top mantissa bits, force quiet */
        return (uint16_t)(sign | 0x7c00u | 0x0200u | (man >> 13));
    }
    int32_t e = (int32_t)exp - 127 + 15;
    if (e >= 0x1f) return (uint16_t)(sign | 0x7c00u); /* overflow -> inf */
    if (e <= 0) {
        /* result is subnormal or zero */
        if (e < -10) return (uint16_t)sign; /* too small: rounds to zero */
        man |= 0x00800000u;                 /* implicit 1 */
        uint32_t shift = (uint32_t)(14 - e); /* 14..24 */
        uint32_t half_man = man >> shift;
        uint32_t round_bit = 1u << (shift - 1);
        if ((man & round_bit) != 0 && (man & (3u * round_bit - 1u)) != 0) {
            /* (man & (round_bit-1)) != 0  ||  lsb(half_man) set  -> round up */
            half_man += 1;
        }
        return (uint16_t)(sign | half_man);
    }
    uint32_t half_exp = (uint32_t)e << 10;
    uint32_t half_man = man >> 13;
    uint32_t round_bit = 0x00001000u;
    uint16_t out = (uint16_t)(sign | half_exp | half_man);
    if ((man & round_bit) != 0 && (man & (3u * round_bit - 1u)) != 0) {
        out = (uint16_t)(out + 1); /* may carry into exponent (correct, incl. -> inf) */
    }
    return out;
}

void fso_encode_f32_to_f16(const float *src, size_t n, uint16_t *dst) {
    for (size_t i = 0; i < n; ++i) dst[i] = fso_f32_to_f16(src[i]);
}

/* ------------------------------------------------------------------------- */
/* dot_product_f16_bytes_f32                                                  */
/* ------------------------------------------------------------------------- */

static inline uint16_t load_le16(const uint8_t *p) { return (uint16_t)(p[0] | ((uint16_t)p[1] << 8)); }

static inline float hreduce8(const float v[8], int mode) {
    /* wide::f32x8::reduce_add — third-party, see fs_oracle.h. */
    if (mode == FSO_HREDUCE_SEQ) {
        float a = ((v[0] + v[1]) + v[2]) + v[3];
        float b = ((v[4] + v[5]) + v[6]) + v[7];
        return a + b;
    }
    if (mode == FSO_HREDUCE_AVX) {
        float a = v[0] + v[4], b = v[1] + v[5], c = v[2] + v[6], d = v[3] + v[7];
        float lo = a + c, hi = b + d;
        return lo + hi;
    }
    float a = (v[0] + v[2]) + (v[1] + v[3]);
    float b = (v[4] + v[6]) + (v[5] + v[7]);
    return a + b;
}

/* crates/frankensearch-index/src/simd.rs:532-571 (generic) == :398-446 (AVX2):
 * 8-lane chunks; chunk c of each group of four goes to accumulator s_{c mod 4} as a
 * separate multiply then add (no FMA); leftover chunks go to s0; (s0+s1)+(s2+s3);
 * horizontal reduce; scalar tail is a fused mul_add. */
float fso_dot_f16_f32(const uint8_t *row, const float *q, size_t dim, int hreduce) {
    size_t chunks = dim / 8;
    float s[4][8];
    memset(s, 0, sizeof s);
    size_t c = 0;
    while (c + 4 <= chunks) {
        for (int a = 0; a < 4; ++a) {
            const uint8_t *b = row + (c + (size_t)a) * 16;
            const float *qq = q + (c + (size_t)a) * 8;
            for (int j = 0; j < 8; ++j) {
                float w = fso_f16_to_f32(load_le16(b + 2 * j));
                float p = w * qq[j];
                s[a][j] = s[a][j] + p;
            }
        }
        c += 4;
    }
    while (c < chunks) {
        const uint8_t *b = row + c * 16;
        const float *qq = q + c * 8;
        for (int j = 0; j < 8; ++j) {
            float w = fso_f16_to_f32(load_le16(b + 2 * j));
            float p = w * qq[j];
            s[0][j] = s[0][j] + p;
        }
        c += 1;
    }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = (s[0][j] + s[1][j]) + (s[2][j] + s[3][j]);
    float result = hreduce8(v, hreduce);
    for (size_t i = chunks * 8; i < dim; ++i) {
        float w = fso_f16_to_f32(load_le16(row + 2 * i));
        result = fmaf(w, q[i], result);
    }
    return result;
}

/* dot_product_f32_bytes_f32 (crates/frankensearch-index/src/simd.rs:581-702): rows of a Quantization::F32 slab.
 * Four 8-lane accumulators over groups of 32 elements (separate multiply and add), (acc0+acc1)+(acc2+acc3), the
 * leftover 8-element chunks added to that SUM (not to acc0 as in the f16 kernel), reduce_add, fused scalar tail. */
float fso_dot_f32_bytes_f32(const uint8_t *row, const float *q, size_t dim, int hreduce) {
    size_t groups = dim / 32, chunks = dim / 8;
    float acc[4][8];
    memset(acc, 0, sizeof acc);
    for (size_t g = 0; g < groups; ++g)
        for (int x = 0; x < 4; ++x)
            for (int j = 0; j < 8; ++j) {
                size_t o = g * 32 + (size_t)x * 8 + (size_t)j;
                float w;
                memcpy(&w, row + 4 * o, 4); /* f32::from_le_bytes on a little-endian host */
                float p = w * q[o];
                acc[x][j] = acc[x][j] + p;
            }
    float sum[8];
    for (int j = 0; j < 8; ++j) sum[j] = (acc[0][j] + acc[1][j]) + (acc[2][j] + acc[3][j]);
    for (size_t c = groups * 4; c < chunks; ++c)
        for (int j = 0; j < 8; ++j) {
            size_t o = c * 8 + (size_t)j;
            float w;
            memcpy(&w, row + 4 * o, 4);
            float p = w * q[o];
            sum[j] = sum[j] + p;
        }
    float result = hreduce8(sum, hreduce);
    for (size_t i = chunks * 8; i < dim; ++i) {
        float w;
        memcpy(&w, row + 4 * i, 4);
        result = fmaf(w, q[i], result);
    }
    return result;
}

/* dot_product_f32_f32 (crates/frankensearch-index/src/simd.rs:134-222). */
float fso_dot_f32_f32(const float *a, const float *b, size_t n, int hreduce) {
    size_t groups = n / 32, chunks = n / 8;
    float acc[4][8];
    memset(acc, 0, sizeof acc);
    for (size_t g = 0; g < groups; ++g)
        for (int x = 0; x < 4; ++x)
            for (int j = 0; j < 8; ++j) {
                size_t o = g * 32 + (size_t)x * 8 + (size_t)j;
                float p = a[o] * b[o];
                acc[x][j] = acc[x][j] + p;
            }
    float sum[8];
    for (int j = 0; j < 8; ++j) sum[j] = (acc[0][j] + acc[1][j]) + (acc[2][j] + acc[3][j]);
    for (size_t c = groups * 4; c < chunks; ++c)
        for (int j = 0; j < 8; ++j) {
            float p = a[c * 8 + (size_t)j] * b[c * 8 + (size_t)j];
            sum[j] = sum[j] + p;
        }
    float result = hreduce8(sum, hreduce);
    for (size_t i = chunks * 8; i < n; ++i) {
        float p = a[i] * b[i];
        result = result + p;
    }
    return result;
}

/* provided by fs_oracle_avx2.c */
float fso_dot_f16_f32_avx2_impl(const uint8_t *row, const float *q, size_t dim, int hreduce);

int fso_has_avx2_f16c(void) {
#if defined(__x86_64__)
    return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("f16c");
#else
    return 0;
#endif
}

float fso_dot_f16_f32_fast(const uint8_t *row, const float *q, size_t dim, int hreduce) {
    static int have = -1;
    if (have < 0) have = fso_has_avx2_f16c();
    if (have) return fso_dot_f16_f32_avx2_impl(row, q, dim, hreduce);
    return fso_dot_f16_f32(row, q, dim, hreduce);
}

/* ------------------------------------------------------------------------- */
/* ordering + bounded heap                                                    */
/* ------------------------------------------------------------------------- */

typedef struct {
    uint64_t row;
    float score;
} entry_t;

/* score_key (search.rs:1655-1661): NaN ranks as -inf. */
static inline float score_key(float s) { return isnan(s) ? -INFINITY : s; }

/* f32::total_cmp on the keys: sign-magnitude -> two's-complement-ordered integer. */
static inline int32_t total_order_i32(float f) {
    int32_t b = (int32_t)f32_bits(f);
    return b ^ (int32_t)(((uint32_t)(b >> 31)) >> 1);
}
static inline int total_cmp(float a, float b) {
    int32_t x = total_order_i32(a), y = total_order_i32(b);
    return (x > y) - (x < y);
}

/* candidate_is_better (search.rs:1680-1686). */
static inline int is_better(entry_t l, entry_t r) {
    int c = total_cmp(score_key(l.score), score_key(r.score));
    if (c > 0) return 1;
    if (c < 0) return 0;
    return l.row < r.row;
}

int fso_ranks_before(uint64_t row_a, float score_a, uint64_t row_b, float score_b) {
    entry_t a = {row_a, score_a}, b = {row_b, score_b};
    return is_better(a, b);
}

/* compare_best_first (search.rs:1673-1678) as a qsort comparator. */
static int cmp_best_first(const void *pa, const void *pb) {
    const entry_t *a = (const entry_t *)pa, *b = (const entry_t *)pb;
    if (is_better(*a, *b)) return -1;
    if (is_better(*b, *a)) return 1;
    return 0;
}

/* Binary heap whose top is the WORST retained entry (HeapEntry::cmp, search.rs:116-126). */
typedef struct {
    entry_t *v;
    size_t len, cap;
} heap_t;

static void heap_init(heap_t *h, size_t cap) {
    h->cap = cap ? cap : 1;
    h->len = 0;
    h->v = (entry_t *)malloc(h->cap * sizeof(entry_t));
}
static void heap_free(heap_t *h) {
    free(h->v);
    h->v = NULL;
    h->len = h->cap = 0;
}
/* "worse" == greater in heap order */
static inline int heap_gt(entry_t a, entry_t b) { return is_better(b, a); }
static void heap_push(heap_t *h, entry_t e) {
    if (h->len == h->cap) {
        h->cap *= 2;
        h->v = (entry_t *)realloc(h->v, h->cap * sizeof(entry_t));
    }
    size_t i = h->len++;
    h->v[i] = e;
    while (i > 0) {
        size_t p = (i - 1) / 2;
        if (!heap_gt(h->v[i], h->v[p])) break;
        entry_t t = h->v[i];
        h->v[i] = h->v[p];
        h->v[p] = t;
        i = p;
    }
}
static void heap_pop(heap_t *h) {
    if (h->len == 0) return;
    h->v[0] = h->v[--h->len];
    size_t i = 0;
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < h->len && heap_gt(h->v[l], h->v[m])) m = l;
        if (r < h->len && heap_gt(h->v[r], h->v[m])) m = r;
        if (m == i) break;
        entry_t t = h->v[i];
        h->v[i] = h->v[m];
        h->v[m] = t;
        i = m;
    }
}

/* insert_candidate (search.rs:1688-1702). */
static void insert_candidate(heap_t *h, entry_t c, size_t limit) {
    if (limit == 0) return;
    if (h->len < limit) {
        heap_push(h, c);
        return;
    }
    if (is_better(c, h->v[0])) {
        heap_pop(h);
        heap_push(h, c);
    }
}

static inline int row_live(const uint64_t *live, uint64_t row) {
    return live == NULL || ((live[row >> 6] >> (row & 63)) & 1u);
}

typedef float (*dot_fn)(const uint8_t *, const float *, size_t, int);

/* scan_range_chunk (search.rs:1257-1327), F16 arm, including the `>= cutoff` fast path. */
static void scan_range_chunk(const uint8_t *slab, uint32_t dim, size_t stride, const uint64_t *live, uint64_t start,
                             uint64_t end, const float *q, size_t limit, int hreduce, dot_fn dot,
                             heap_t *heap) {
    float cutoff = -INFINITY;
    for (uint64_t index = start; index < end; ++index) {
        if (!row_live(live, index)) continue;
        float score = dot(slab + index * stride, q, dim, hreduce);
        if (heap->len < limit || score_key(score) >= cutoff) {
            entry_t e = {index, score};
            insert_candidate(heap, e, limit);
            if (heap->len >= limit && heap->len > 0) cutoff = score_key(heap->v[0].score);
        }
    }
}

typedef struct {
    const uint8_t *slab;
    uint32_t dim;
    size_t stride;       /* bytes per row: dim * 2 (F16) or dim * 4 (F32) */
    const uint64_t *live;
    uint64_t nrows;
    const float *q;
    size_t limit, chunk_size, chunk_count;
    int hreduce;
    dot_fn dot;
    heap_t *heaps;       /* one per chunk (top-k mode) */
    entry_t *collect;    /* collect-all mode: slot per row, row==UINT64_MAX when dead */
    size_t next;         /* work counter (atomic) */
} scan_job_t;

static void scan_worker_body(scan_job_t *job) {
    for (;;) {
        size_t c = __atomic_fetch_add(&job->next, 1, __ATOMIC_RELAXED);
        if (c >= job->chunk_count) break;
        uint64_t start = (uint64_t)c * job->chunk_size;
        uint64_t end = start + job->chunk_size;
        if (end > job->nrows) end = job->nrows;
        if (job->collect) {
            size_t stride = job->stride;
            for (uint64_t r = start; r < end; ++r) {
                if (!row_live(job->live, r)) {
                    job->collect[r].row = UINT64_MAX;
                    continue;
                }
                job->collect[r].row = r;
                job->collect[r].score = job->dot(job->slab + r * stride, job->q, job->dim, job->hreduce);
            }
        } else {
            size_t cap = job->limit < (size_t)(end - start) ? job->limit : (size_t)(end - start);
            heap_init(&job->heaps[c], cap + 1);
            scan_range_chunk(job->slab, job->dim, job->stride, job->live, start, end, job->q, job->limit,
                             job->hreduce, job->dot, &job->heaps[c]);
        }
    }
}

/* Persistent worker pool (the reference scans chunks on rayon's persistent pool, search.rs:1013-1036): workers are
 * created once and parked on a condition variable, so a query does not pay thread creation.  Jobs are serialised. */
#define FSO_MAX_WORKERS 512
static struct {
    pthread_mutex_t mu;
    pthread_cond_t start, done;
    pthread_t th[FSO_MAX_WORKERS];
    int nworkers;          /* threads created so far */
    unsigned long gen;     /* job generation */
    int want;              /* workers that take part in the current job */
    int remaining;         /* participants still running */
    scan_job_t *job;
} g_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, {0}, 0, 0, 0, 0, NULL};
static pthread_mutex_t g_job_mu = PTHREAD_MUTEX_INITIALIZER;

static void *pool_worker(void *arg) {
    const int me = (int)(intptr_t)arg;
    unsigned long seen = 0;
    pthread_mutex_lock(&g_pool.mu);
    for (;;) {
        while (g_pool.gen == seen) pthread_cond_wait(&g_pool.start, &g_pool.mu);
        seen = g_pool.gen;
        if (me >= g_pool.want) continue;
        scan_job_t *job = g_pool.job;
        pthread_mutex_unlock(&g_pool.mu);
        scan_worker_body(job);
        pthread_mutex_lock(&g_pool.mu);
        if (--g_pool.remaining == 0) pthread_cond_signal(&g_pool.done);
    }
    return NULL;
}

static void run_job(scan_job_t *job, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > job->chunk_count) nthreads = (int)(job->chunk_count ? job->chunk_count : 1);
    if (nthreads > FSO_MAX_WORKERS + 1) nthreads = FSO_MAX_WORKERS + 1;
    job->next = 0;
    if (nthreads == 1) {
        scan_worker_body(job);
        return;
    }
    pthread_mutex_lock(&g_job_mu);
    pthread_mutex_lock(&g_pool.mu);
    while (g_pool.nworkers < nthreads - 1) {
        if (pthread_create(&g_pool.th[g_pool.nworkers], NULL, pool_worker, (void *)(intptr_t)g_pool.nworkers) != 0) break;
        pthread_detach(g_pool.th[g_pool.nworkers]);
        g_pool.nworkers++;
    }
    g_pool.job = job;
    g_pool.want = nthreads - 1 < g_pool.nworkers ? nthreads - 1 : g_pool.nworkers;
    g_pool.remaining = g_pool.want;
    g_pool.gen++;
    pthread_cond_broadcast(&g_pool.start);
    pthread_mutex_unlock(&g_pool.mu);
    scan_worker_body(job);  /* the caller is a worker too */
    pthread_mutex_lock(&g_pool.mu);
    while (g_pool.remaining > 0) pthread_cond_wait(&g_pool.done, &g_pool.mu);
    pthread_mutex_unlock(&g_pool.mu);
    pthread_mutex_unlock(&g_job_mu);
}

/* search_top_k_internal (search.rs:426-494) for a main index with no WAL and no filter,
 * scan_parallel (:1013-1036), merge_partial_heaps (:1704-1720), resolve_hits sort (:1493-1501). */
static size_t search_top_k_impl(const uint8_t *slab, size_t stride, dot_fn dot, uint64_t nrows, uint32_t dim,
                                const uint64_t *live, const float *q, size_t k, size_t parallel_threshold,
                                size_t chunk_size, int parallel_enabled, int nthreads, int hreduce,
                                uint32_t *out_rows, float *out_scores);

size_t fso_search_top_k(const uint8_t *slab, uint64_t nrows, uint32_t dim, const uint64_t *live,
                        const float *q, size_t k, size_t parallel_threshold, size_t chunk_size,
                        int parallel_enabled, int nthreads, int hreduce, uint32_t *out_rows,
                        float *out_scores) {
    dot_fn dot = fso_has_avx2_f16c() ? fso_dot_f16_f32_fast : fso_dot_f16_f32;
    return search_top_k_impl(slab, (size_t)dim * 2, dot, nrows, dim, live, q, k, parallel_threshold, chunk_size,
                             parallel_enabled, nthreads, hreduce, out_rows, out_scores);
}

/* The same search over a Quantization::F32 slab (scan_range_chunk's F32 arm, search.rs:1293-1325). */
size_t fso_search_top_k_f32(const uint8_t *slab, uint64_t nrows, uint32_t dim, const uint64_t *live,
                            const float *q, size_t k, int nthreads, int hreduce, uint32_t *out_rows,
                            float *out_scores) {
    return search_top_k_impl(slab, (size_t)dim * 4, fso_dot_f32_bytes_f32, nrows, dim, live, q, k, 10000, 1024, 1,
                             nthreads, hreduce, out_rows, out_scores);
}

static size_t search_top_k_impl(const uint8_t *slab, size_t stride, dot_fn dot, uint64_t nrows, uint32_t dim,
                                const uint64_t *live, const float *q, size_t k, size_t parallel_threshold,
                                size_t chunk_size, int parallel_enabled, int nthreads, int hreduce,
                                uint32_t *out_rows, float *out_scores) {
    if (k == 0 || nrows == 0) return 0;
    if (chunk_size == 0) chunk_size = 1;
    int use_parallel = parallel_enabled && nrows >= parallel_threshold;

    scan_job_t job;
    memset(&job, 0, sizeof job);
    job.slab = slab;
    job.dim = dim;
    job.stride = stride;
    job.live = live;
    job.nrows = nrows;
    job.q = q;
    job.limit = k;
    job.hreduce = hreduce;
    job.dot = dot;

    entry_t *winners = NULL;
    size_t nwin = 0;

    if (k >= nrows) {
        /* collect-all path (search.rs:449-473): score every live row, sort best-first. */
        job.chunk_size = use_parallel ? chunk_size : (size_t)nrows;
        job.chunk_count = (size_t)((nrows + job.chunk_size - 1) / job.chunk_size);
        job.collect = (entry_t *)malloc(sizeof(entry_t) * (size_t)nrows);
        run_job(&job, use_parallel ? nthreads : 1);
        winners = job.collect;
        for (uint64_t r = 0; r < nrows; ++r)
            if (winners[r].row != UINT64_MAX) winners[nwin++] = winners[r];
    } else {
        job.chunk_size = use_parallel ? chunk_size : (size_t)nrows;
        job.chunk_count = (size_t)((nrows + job.chunk_size - 1) / job.chunk_size);
        job.heaps = (heap_t *)calloc(job.chunk_count, sizeof(heap_t));
        run_job(&job, use_parallel ? nthreads : 1);
        heap_t merged;
        heap_init(&merged, k + 1);
        for (size_t c = 0; c < job.chunk_count; ++c) {
            for (size_t i = 0; i < job.heaps[c].len; ++i) insert_candidate(&merged, job.heaps[c].v[i], k);
            heap_free(&job.heaps[c]);
        }
        free(job.heaps);
        winners = merged.v;
        nwin = merged.len;
    }
    qsort(winners, nwin, sizeof(entry_t), cmp_best_first);
    size_t n = nwin < k ? nwin : k;
    for (size_t i = 0; i < n; ++i) {
        out_rows[i] = (uint32_t)winners[i].row;
        out_scores[i] = winners[i].score;
    }
    free(winners);
    return n;
}

/* search_top_k_classified validation (search.rs:227-261), ensure_query_dimension (:1602-1610). */
int fso_classify_query(const float *q, size_t qlen, uint32_t dim, size_t k, int *zero_signal) {
    *zero_signal = 0;
    if (qlen != dim) return FSO_ERR_DIMENSION_MISMATCH;
    if (k == 0) {
        *zero_signal = 1;
        return FSO_OK;
    }
    for (size_t i = 0; i < qlen; ++i)
        if (!isfinite(q[i])) return FSO_ERR_INVALID_CONFIG;
    int all_zero = 1;
    for (size_t i = 0; i < qlen; ++i)
        if (q[i] != 0.0f) all_zero = 0;
    if (all_zero) *zero_signal = 2;
    return FSO_OK;
}

/* ------------------------------------------------------------------------- */
/* int8 two-pass                                                              */
/* ------------------------------------------------------------------------- */

static inline int8_t quant_i8(float x, float scale) {
    /* (x * scale).round().clamp(-127, 127) as i8 — Rust `as` maps NaN to 0 */
    float v = roundf(x * scale);
    if (isnan(v)) return 0;
    if (v > 127.0f) v = 127.0f;
    if (v < -127.0f) v = -127.0f;
    return (int8_t)v;
}

/* quantize_f16_le_bytes_to_i8_generic (simd.rs:1865-1886) */
void fso_quantize_slab_i8(const uint8_t *slab, uint64_t n_values, int8_t *out) {
    float max_abs = 0.0f;
    for (uint64_t i = 0; i < n_values; ++i) {
        float v = fabsf(fso_f16_to_f32(load_le16(slab + 2 * i)));
        if (v > max_abs) max_abs = v; /* f32::max ignores NaN */
    }
    if (!(max_abs > 0.0f)) {
        memset(out, 0, (size_t)n_values);
        return;
    }
    float scale = 127.0f / max_abs;
    for (uint64_t i = 0; i < n_values; ++i) out[i] = quant_i8(fso_f16_to_f32(load_le16(slab + 2 * i)), scale);
}

/* quantize_i8_query (search.rs:1616-1626) */
void fso_quantize_query_i8(const float *q, size_t dim, int8_t *out) {
    float max_abs = 0.0f;
    for (size_t i = 0; i < dim; ++i) {
        float v = fabsf(q[i]);
        if (v > max_abs) max_abs = v;
    }
    if (!(max_abs > 0.0f)) {
        memset(out, 0, dim);
        return;
    }
    float scale = 127.0f / max_abs;
    for (size_t i = 0; i < dim; ++i) out[i] = quant_i8(q[i], scale);
}

/* dot_i8_i8 (simd.rs:757, 1240-1286): exact i32 sum */
int32_t fso_dot_i8_i8(const int8_t *a, const int8_t *b, size_t n) {
    int32_t s = 0;
    for (size_t i = 0; i < n; ++i) s += (int32_t)a[i] * (int32_t)b[i];
    return s;
}

typedef struct {
    uint64_t row;
    uint32_t desc_key; /* int8_heap_key's high word: smaller = better */
} i8cand_t;

static int cmp_i8cand(const void *pa, const void *pb) {
    const i8cand_t *a = (const i8cand_t *)pa, *b = (const i8cand_t *)pb;
    if (a->desc_key != b->desc_key) return a->desc_key < b->desc_key ? -1 : 1;
    return a->row < b->row ? -1 : (a->row > b->row);
}

/* search_top_k_int8_two_pass_impl (search.rs:589-661), int8_heap_key / _from_f32 (:141-156),
 * MAX_EXACT_I8_DOT_DIM = 1040 (:130-134). */
size_t fso_search_int8_two_pass(const uint8_t *slab, const int8_t *slab_i8, uint64_t nrows, uint32_t dim,
                                const uint64_t *live, const float *q, size_t k, size_t candidate_multiplier,
                                int hreduce, uint32_t *out_rows, float *out_scores) {
    if (k == 0 || nrows == 0) return 0;
    size_t mult = candidate_multiplier ? candidate_multiplier : 1;
    size_t cc = k * mult;
    if (cc > nrows) cc = (size_t)nrows;
    size_t kmin = k < nrows ? k : (size_t)nrows;
    if (cc < kmin) cc = kmin;
    int8_t *qi = (int8_t *)malloc(dim ? dim : 1);
    fso_quantize_query_i8(q, dim, qi);
    i8cand_t *all = (i8cand_t *)malloc(sizeof(i8cand_t) * (size_t)nrows);
    size_t n = 0;
    for (uint64_t r = 0; r < nrows; ++r) {
        if (!row_live(live, r)) continue;
        int32_t dot = fso_dot_i8_i8(slab_i8 + r * dim, qi, dim);
        uint32_t asc;
        if (dim <= 1040) {
            asc = (uint32_t)dot ^ 0x80000000u;
        } else {
            float f = (float)dot;
            uint32_t bits = f32_bits(f);
            uint32_t mask = (uint32_t)(-(int32_t)(bits >> 31)) | 0x80000000u;
            asc = bits ^ mask;
        }
        all[n].row = r;
        all[n].desc_key = ~asc;
        ++n;
    }
    qsort(all, n, sizeof(i8cand_t), cmp_i8cand); /* the bounded heap keeps exactly the cc smallest keys */
    if (n > cc) n = cc;
    heap_t heap;
    heap_init(&heap, k + 1);
    size_t stride = (size_t)dim * 2;
    for (size_t i = 0; i < n; ++i) {
        entry_t e = {all[i].row, fso_dot_f16_f32(slab + all[i].row * stride, q, dim, hreduce)};
        insert_candidate(&heap, e, k);
    }
    qsort(heap.v, heap.len, sizeof(entry_t), cmp_best_first);
    size_t outn = heap.len;
    for (size_t i = 0; i < outn; ++i) {
        out_rows[i] = (uint32_t)heap.v[i].row;
        out_scores[i] = heap.v[i].score;
    }
    heap_free(&heap);
    free(all);
    free(qi);
    return outn;
}

/* ---- 4-bit two-pass (search.rs:860-1000; simd.rs:1286-1556, 1886-1900, 2153-2215) ---- */
/* nibble_of_4bit / nibble_of (simd.rs:1892-1896, search.rs:1632-1635): round half away from zero, clamp +-7, NaN -> 0
 * (Rust `as i8`), 4-bit two's complement in the low nibble. */
static uint8_t nibble_of(float value, float scale) {
    float v = roundf(value * scale);
    if (v != v) return 0;
    if (v < -7.0f) v = -7.0f;
    if (v > 7.0f) v = 7.0f;
    return (uint8_t)((int8_t)v) & 0x0F;
}
static int32_t nibble_lo(uint8_t b) { return (int32_t)(int8_t)((b & 0x0F) ^ 0x08) - 8; }   /* simd.rs:1290-1298 */
static int32_t nibble_hi(uint8_t b) { return (int32_t)(int8_t)((b >> 4) ^ 0x08) - 8; }

/* pack_f16_le_bytes_to_4bit (simd.rs:2153-2215): ONE corpus-wide max-abs scale 7/max (0 when max <= 1e-9; f32::max
 * ignores NaN), dim.div_ceil(2) bytes per vector, low nibble = even dim. */
void fso_pack_slab_4bit(const uint8_t *slab_f16, uint64_t count, uint32_t dim, uint8_t *out) {
    if (dim == 0) return;
    float max_abs = 0.0f;
    uint64_t n = count * dim;
    for (uint64_t i = 0; i < n; ++i) {
        float v = fabsf(fso_f16_to_f32((uint16_t)(slab_f16[2 * i] | (slab_f16[2 * i + 1] << 8))));
        if (v > max_abs) max_abs = v; /* NaN compares false: ignored, like f32::max */
    }
    float scale = max_abs > 1e-9f ? 7.0f / max_abs : 0.0f;
    size_t bpv = (dim + 1) / 2;
    memset(out, 0, (size_t)count * bpv);
    for (uint64_t v = 0; v < count; ++v)
        for (uint32_t d = 0; d < dim; ++d) {
            uint64_t i = v * dim + d;
            float x = fso_f16_to_f32((uint16_t)(slab_f16[2 * i] | (slab_f16[2 * i + 1] << 8)));
            uint8_t nib = nibble_of(x, scale);
            out[v * bpv + d / 2] |= (d % 2 == 0) ? nib : (uint8_t)(nib << 4);
        }
}

/* pack_4bit_query (search.rs:1640-1653): the query's own max-abs scale. */
void fso_pack_query_4bit(const float *q, uint32_t dim, uint8_t *out) {
    float max_abs = 0.0f;
    for (uint32_t d = 0; d < dim; ++d) {
        float v = fabsf(q[d]);
        if (v > max_abs) max_abs = v;
    }
    float scale = max_abs > 1e-9f ? 7.0f / max_abs : 0.0f;
    memset(out, 0, (dim + 1) / 2);
    for (uint32_t d = 0; d < dim; ++d) {
        uint8_t nib = nibble_of(q[d], scale);
        out[d / 2] |= (d % 2 == 0) ? nib : (uint8_t)(nib << 4);
    }
}

/* dot_packed_4bit / dot_4bit_prepared (simd.rs:1338-1556): exact integer sum of nibble products. */
int32_t fso_dot_4bit(const uint8_t *stored, const uint8_t *query, size_t nbytes) {
    int32_t sum = 0;
    for (size_t i = 0; i < nbytes; ++i)
        sum += nibble_lo(stored[i]) * nibble_lo(query[i]) + nibble_hi(stored[i]) * nibble_hi(query[i]);
    return sum;
}

/* search_top_k_4bit_two_pass (search.rs:876-946) + nibble_scan_range (:951-983): pass 1 keeps the top
 * candidate_count = max(min(k * max(mult, 1), N), min(k, N)) by (dot as f32) under the HeapEntry order (score desc, row
 * asc), pass 2 re-scores them with the exact f16 dot and selects k.  Raw-slab form: no doc ids, so resolve_hits'
 * dedup has nothing to merge. */
size_t fso_search_4bit_two_pass(const uint8_t *slab, const uint8_t *slab_4bit, uint64_t nrows, uint32_t dim,
                                const uint64_t *live, const float *q, size_t k, size_t candidate_multiplier, int hreduce,
                                uint32_t *out_rows, float *out_scores) {
    if (k == 0 || nrows == 0) return 0;
    size_t mult = candidate_multiplier ? candidate_multiplier : 1;
    size_t cc = k * mult;
    if (cc > nrows) cc = (size_t)nrows;
    size_t kmin = k < nrows ? k : (size_t)nrows;
    if (cc < kmin) cc = kmin;
    size_t bpv = (dim + 1) / 2;
    uint8_t *qp = (uint8_t *)malloc(bpv ? bpv : 1);
    fso_pack_query_4bit(q, dim, qp);
    heap_t cand;
    heap_init(&cand, cc + 1);
    float cutoff = -INFINITY;
    for (uint64_t r = 0; r < nrows; ++r) {
        if (!row_live(live, r)) continue;
        float score = (float)fso_dot_4bit(slab_4bit + r * bpv, qp, bpv);
        if (cand.len < cc || score_key(score) >= cutoff) {
            entry_t e = {r, score};
            insert_candidate(&cand, e, cc);
            if (cand.len >= cc && cand.len > 0) cutoff = score_key(cand.v[0].score);
        }
    }
    heap_t heap;
    heap_init(&heap, k + 1);
    size_t stride = (size_t)dim * 2;
    for (size_t i = 0; i < cand.len; ++i) {
        entry_t e = {cand.v[i].row, fso_dot_f16_f32(slab + cand.v[i].row * stride, q, dim, hreduce)};
        insert_candidate(&heap, e, k);
    }
    qsort(heap.v, heap.len, sizeof(entry_t), cmp_best_first);
    size_t outn = heap.len;
    for (size_t i = 0; i < outn; ++i) {
        out_rows[i] = (uint32_t)heap.v[i].row;
        out_scores[i] = heap.v[i].score;
    }
    heap_free(&heap);
    heap_free(&cand);
    free(qp);
    return outn;
}

/* VectorIndex::mrl_search_with_stats (crates/frankensearch-index/src/mrl.rs:241-395): truncated scan over the first
 * search_dims dimensions keeping the top rescore_top_k (0 = 3 * limit) under (nan_safe(score) desc, index asc)
 * (:160-207, :468-537), resident WAL entries scored with the truncated f32 dot and skipped when non-finite (:539-583),
 * every candidate re-scored over rescore_dims (0 or > dim = full; never fewer than search_dims, :92-105) with the f16 dot
 * on the row's leading bytes or the f32 dot for WAL entries (:587-618), sorted by (score desc, index asc), truncated to
 * limit, WAL hits reported at the virtual index nrows + i (:642-683).  No doc-id dedup and no WAL shadowing here.
 * The caller handles search_dims >= dim (plain search_top_k) and limit == 0.  wal_vecs may be NULL. */
size_t fso_mrl_search(const uint8_t *slab, uint64_t nrows, uint32_t dim, const uint64_t *live, const float *const *wal_vecs,
                      size_t wal_len, const float *q, size_t limit, size_t search_dims, size_t rescore_dims,
                      size_t rescore_top_k, int hreduce, uint32_t *out_rows, float *out_scores) {
    if (limit == 0 || (nrows == 0 && wal_len == 0) || search_dims == 0 || search_dims >= dim) return 0;
    size_t rdims = (rescore_dims == 0 || rescore_dims > dim) ? dim : rescore_dims;
    if (rdims < search_dims) rdims = search_dims;
    size_t rtop = rescore_top_k ? rescore_top_k : limit * 3;
    const uint64_t wal_tag = 1ull << 63;
    size_t stride = (size_t)dim * 2;
    heap_t heap;
    heap_init(&heap, rtop + 1);
    float cutoff = -INFINITY;
    for (uint64_t r = 0; r < nrows; ++r) {
        if (!row_live(live, r)) continue;
        float score = fso_dot_f16_f32(slab + r * stride, q, search_dims, hreduce);
        if (heap.len < rtop || score_key(score) >= cutoff) {
            entry_t e = {r, score};
            insert_candidate(&heap, e, rtop);
            if (heap.len >= rtop && heap.len > 0) cutoff = score_key(heap.v[0].score);
        }
    }
    for (size_t w = 0; w < wal_len; ++w) {
        float score = fso_dot_f32_f32(wal_vecs[w], q, search_dims, hreduce);
        if (!isfinite(score)) continue;
        entry_t e = {wal_tag | (uint64_t)w, score};
        insert_candidate(&heap, e, rtop);
    }
    for (size_t i = 0; i < heap.len; ++i) {
        uint64_t idx = heap.v[i].row;
        if (idx & wal_tag) heap.v[i].score = fso_dot_f32_f32(wal_vecs[idx & ~wal_tag], q, rdims, hreduce);
        else heap.v[i].score = fso_dot_f16_f32(slab + idx * stride, q, rdims, hreduce);
    }
    qsort(heap.v, heap.len, sizeof(entry_t), cmp_best_first);
    size_t n = heap.len < limit ? heap.len : limit;
    for (size_t i = 0; i < n; ++i) {
        uint64_t idx = heap.v[i].row;
        out_rows[i] = (idx & wal_tag) ? (uint32_t)(nrows + (idx & ~wal_tag)) : (uint32_t)idx;
        out_scores[i] = heap.v[i].score;
    }
    heap_free(&heap);
    return n;
}



/* VectorIndex::dot_query_at (lib.rs:3229-3239) over a row list (two_tier.rs:1566-1631). */
void fso_gather_dot(const uint8_t *slab, uint32_t dim, const float *q, const uint32_t *rows,
                    size_t n, int hreduce, float *out) {
    for (size_t i = 0; i < n; ++i)
        out[i] = fso_dot_f16_f32(slab + (size_t)rows[i] * dim * 2, q, dim, hreduce);
}

/* ------------------------------------------------------------------------- */
/* hashes                                                                     */
/* ------------------------------------------------------------------------- */

/* fnv1a_hash (lib.rs:6120-6127). */
uint64_t fso_fnv1a64(const uint8_t *bytes, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) {
        h ^= bytes[i];
        h *= 0x00000100000001b3ull;
    }
    return h;
}

/* CRC-32/IEEE (crc32fast, lib.rs:6115-6118). */
uint32_t fso_crc32(const uint8_t *bytes, size_t n) {
    static uint32_t table[256];
    static int init = 0;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? (0xedb88320u ^ (c >> 1)) : (c >> 1);
            table[i] = c;
        }
        init = 1;
    }
    uint32_t c = 0xffffffffu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ bytes[i]) & 0xffu] ^ (c >> 8);
    return c ^ 0xffffffffu;
}

/* align_up (lib.rs:6028-6045; tests :10201-10216). */
uint64_t fso_align_up(uint64_t value, uint64_t alignment) {
    if (alignment == 0) return value;
    uint64_t rem = value % alignment;
    return rem == 0 ? value : value + (alignment - rem);
}

int fso_vector_signal_usable(const float *v, size_t n) {
    float norm_sq = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        if (!isfinite(v[i])) return 0;
        float p = v[i] * v[i];
        norm_sq = norm_sq + p;
    }
    return norm_sq > 0.0f && isfinite(norm_sq);
}

/* l2_normalize_in_place (crates/frankensearch-core/src/traits.rs:590-618). */
void fso_l2_normalize(float *v, size_t n) {
    float norm_sq = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        float p = v[i] * v[i];
        norm_sq = norm_sq + p;
    }
    if (!isfinite(norm_sq) || norm_sq < 1.1920929e-7f) {
        for (size_t i = 0; i < n; ++i) v[i] = 0.0f;
        return;
    }
    float inv = 1.0f / sqrtf(norm_sq);
    for (size_t i = 0; i < n; ++i) v[i] = v[i] * inv;
}

/* ------------------------------------------------------------------------- */
/* FSVI v1                                                                    */
/* ------------------------------------------------------------------------- */

struct fso_fsvi {
    uint8_t *data;
    size_t len;
    uint32_t dim;
    uint8_t quant;
    uint64_t record_count;
    uint64_t vectors_offset;
    size_t records_offset;
    size_t strings_offset;
    /* resident WAL entries (wal.rs WalEntry: doc_id, doc_id_hash, f32 embedding) */
    char **wal_ids;
    float **wal_vecs;
    size_t wal_len, wal_cap;
};

typedef struct {
    uint64_t hash;
    const char *doc_id;
    size_t doc_len;
    const float *vec;
    size_t seq; /* insertion order, for stable sort */
} pending_t;

static int cmp_pending(const void *pa, const void *pb) {
    const pending_t *a = (const pending_t *)pa, *b = (const pending_t *)pb;
    if (a->hash != b->hash) return a->hash < b->hash ? -1 : 1;
    size_t m = a->doc_len < b->doc_len ? a->doc_len : b->doc_len;
    int c = memcmp(a->doc_id, b->doc_id, m);
    if (c != 0) return c;
    if (a->doc_len != b->doc_len) return a->doc_len < b->doc_len ? -1 : 1;
    return a->seq < b->seq ? -1 : (a->seq > b->seq); /* stable (lib.rs:3753-3762) */
}

static void put16(uint8_t *p, uint16_t v) {
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
}
static void put32(uint8_t *p, uint32_t v) {
    for (int i = 0; i < 4; ++i) p[i] = (uint8_t)(v >> (8 * i));
}
static void put64(uint8_t *p, uint64_t v) {
    for (int i = 0; i < 8; ++i) p[i] = (uint8_t)(v >> (8 * i));
}
static uint16_t get16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static uint32_t get32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint64_t get64(const uint8_t *p) { return (uint64_t)get32(p) | ((uint64_t)get32(p + 4) << 32); }

/* VectorIndexWriter::write_record validation (lib.rs:3637-3672) + finish (lib.rs:3752-3943):
 * header (lib.rs:5714-5768) | record table | string table | pad to 64 | f16 slab. */
int fso_fsvi_write(const char *path, const char *embedder_id, const char *embedder_revision,
                   uint32_t dim, uint64_t n, const char *const *doc_ids, const float *vectors,
                   uint8_t compaction_gen) {
    return fso_fsvi_write_quant(path, embedder_id, embedder_revision, dim, n, doc_ids, vectors, compaction_gen, 1);
}

/* quantization: 1 = F16 (the default), 0 = F32 (raw little-endian f32 rows, write_vector_slab lib.rs:6017-6024). */
int fso_fsvi_write_quant(const char *path, const char *embedder_id, const char *embedder_revision,
                         uint32_t dim, uint64_t n, const char *const *doc_ids, const float *vectors,
                         uint8_t compaction_gen, uint8_t quantization) {
    if (dim == 0 || quantization > 1) return FSO_ERR_INVALID_CONFIG;
    const size_t elem = quantization == 1 ? 2 : 4;
    pending_t *recs = (pending_t *)malloc(sizeof(pending_t) * (size_t)(n ? n : 1));
    size_t strings_len = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const float *v = vectors + (size_t)i * dim;
        for (uint32_t d = 0; d < dim; ++d)
            if (!isfinite(v[d])) {
                free(recs);
                return FSO_ERR_INVALID_CONFIG;
            }
        if (!fso_vector_signal_usable(v, dim)) {
            free(recs);
            return FSO_ERR_INVALID_CONFIG;
        }
        size_t dl = strlen(doc_ids[i]);
        if (dl > 0xffff) {
            free(recs);
            return FSO_ERR_INVALID_CONFIG;
        }
        recs[i].hash = fso_fnv1a64((const uint8_t *)doc_ids[i], dl);
        recs[i].doc_id = doc_ids[i];
        recs[i].doc_len = dl;
        recs[i].vec = v;
        recs[i].seq = (size_t)i;
        strings_len += dl;
    }
    qsort(recs, (size_t)n, sizeof(pending_t), cmp_pending);

    size_t idl = strlen(embedder_id), rvl = strlen(embedder_revision);
    size_t header_len = 4 + 2 + 2 + idl + 2 + rvl + 4 + 1 + 3 + 8 + 8 + 4;
    size_t records_bytes = (size_t)n * 16;
    uint64_t pre = (uint64_t)header_len + records_bytes + strings_len;
    uint64_t vectors_offset = fso_align_up(pre, 64);
    size_t total = (size_t)vectors_offset + (size_t)n * dim * elem;
    uint8_t *buf = (uint8_t *)calloc(total ? total : 1, 1);

    size_t c = 0;
    memcpy(buf + c, "FSVI", 4);
    c += 4;
    put16(buf + c, 1);
    c += 2;
    put16(buf + c, (uint16_t)idl);
    c += 2;
    memcpy(buf + c, embedder_id, idl);
    c += idl;
    put16(buf + c, (uint16_t)rvl);
    c += 2;
    memcpy(buf + c, embedder_revision, rvl);
    c += rvl;
    put32(buf + c, dim);
    c += 4;
    buf[c++] = quantization; /* Quantization::{F32 = 0, F16 = 1} (lib.rs:203-208) */
    buf[c++] = compaction_gen;
    put16(buf + c, 0); /* publication nonce */
    c += 2;
    put64(buf + c, n);
    c += 8;
    put64(buf + c, vectors_offset);
    c += 8;
    put32(buf + c, fso_crc32(buf, c));
    c += 4;

    size_t str_off = 0;
    uint8_t *rec = buf + c;
    uint8_t *str = buf + c + records_bytes;
    for (uint64_t i = 0; i < n; ++i) {
        put64(rec + i * 16, recs[i].hash);
        put32(rec + i * 16 + 8, (uint32_t)str_off);
        put16(rec + i * 16 + 12, (uint16_t)recs[i].doc_len);
        put16(rec + i * 16 + 14, 0);
        memcpy(str + str_off, recs[i].doc_id, recs[i].doc_len);
        str_off += recs[i].doc_len;
    }
    uint8_t *slab = buf + vectors_offset;
    for (uint64_t i = 0; i < n; ++i)
        for (uint32_t d = 0; d < dim; ++d) {
            if (quantization == 1) put16(slab + ((size_t)i * dim + d) * 2, fso_f32_to_f16(recs[i].vec[d]));
            else memcpy(slab + ((size_t)i * dim + d) * 4, &recs[i].vec[d], 4); /* to_le_bytes, little-endian host */
        }
    free(recs);

    FILE *f = fopen(path, "wb");
    if (!f) {
        free(buf);
        return FSO_ERR_IO;
    }
    size_t w = fwrite(buf, 1, total, f);
    fclose(f);
    free(buf);
    return w == total ? FSO_OK : FSO_ERR_IO;
}

/* parse_header (lib.rs:4049-4144) + open-time layout checks (lib.rs:1780-1816, 3510-3537). */
int fso_fsvi_open(const char *path, fso_fsvi **out) {
    *out = NULL;
    FILE *f = fopen(path, "rb");
    if (!f) return FSO_ERR_IO;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *data = (uint8_t *)malloc((size_t)(sz > 0 ? sz : 1));
    size_t rd = fread(data, 1, (size_t)sz, f);
    fclose(f);
    if ((long)rd != sz) {
        free(data);
        return FSO_ERR_IO;
    }
    size_t len = (size_t)sz, c = 0;
#define NEED(nbytes)                        \
    do {                                    \
        if (c + (size_t)(nbytes) > len) {   \
            free(data);                     \
            return FSO_ERR_INDEX_CORRUPTED; \
        }                                   \
    } while (0)
    NEED(4);
    if (memcmp(data, "FSVI", 4) != 0) {
        free(data);
        return FSO_ERR_INDEX_CORRUPTED;
    }
    c = 4;
    NEED(2);
    uint16_t version = get16(data + c);
    c += 2;
    if (version != 1) {
        free(data);
        return FSO_ERR_INDEX_VERSION_MISMATCH;
    }
    NEED(2);
    size_t idl = get16(data + c);
    c += 2;
    NEED(idl);
    c += idl;
    NEED(2);
    size_t rvl = get16(data + c);
    c += 2;
    NEED(rvl);
    c += rvl;
    NEED(4);
    uint32_t dim = get32(data + c);
    c += 4;
    if (dim == 0) {
        free(data);
        return FSO_ERR_INDEX_CORRUPTED;
    }
    NEED(1);
    uint8_t quant = data[c++];
    if (quant > 1) {
        free(data);
        return FSO_ERR_INDEX_CORRUPTED;
    }
    NEED(3);
    c += 3;
    NEED(8);
    uint64_t record_count = get64(data + c);
    c += 8;
    NEED(8);
    uint64_t vectors_offset = get64(data + c);
    c += 8;
    NEED(4);
    uint32_t crc = get32(data + c);
    if (fso_crc32(data, c) != crc) {
        free(data);
        return FSO_ERR_INDEX_CORRUPTED;
    }
    c += 4;
#undef NEED
    size_t elem = quant == 1 ? 2 : 4;
    size_t records_offset = c;
    if (records_offset + record_count * 16 > vectors_offset || vectors_offset % 64 != 0 ||
        vectors_offset + record_count * dim * elem > len) {
        free(data);
        return FSO_ERR_INDEX_CORRUPTED;
    }
    fso_fsvi *idx = (fso_fsvi *)calloc(1, sizeof *idx);
    idx->data = data;
    idx->len = len;
    idx->dim = dim;
    idx->quant = quant;
    idx->record_count = record_count;
    idx->vectors_offset = vectors_offset;
    idx->records_offset = records_offset;
    idx->strings_offset = records_offset + (size_t)record_count * 16;
    for (uint64_t r = 0; r < record_count; ++r) {
        const uint8_t *rec = data + records_offset + r * 16;
        uint64_t off = get32(rec + 8), dl = get16(rec + 12);
        if (idx->strings_offset + off + dl > vectors_offset) {
            fso_fsvi_close(idx);
            return FSO_ERR_INDEX_CORRUPTED;
        }
    }
    *out = idx;
    return FSO_OK;
}

void fso_fsvi_close(fso_fsvi *idx) {
    if (!idx) return;
    for (size_t i = 0; i < idx->wal_len; ++i) {
        free(idx->wal_ids[i]);
        free(idx->wal_vecs[i]);
    }
    free(idx->wal_ids);
    free(idx->wal_vecs);
    free(idx->data);
    free(idx);
}

uint64_t fso_fsvi_wal_count(const fso_fsvi *idx) { return idx->wal_len; }
uint32_t fso_fsvi_wal_doc_id(const fso_fsvi *idx, uint64_t i, const char **ptr) {
    *ptr = idx->wal_ids[i];
    return (uint32_t)strlen(idx->wal_ids[i]);
}

/* VectorIndex::soft_delete_batch (lib.rs:2313-2397) for one doc id: every live main record with that id is
 * tombstoned (step 1, :2328-2356), every resident WAL entry with that id is dropped (step 2, :2358-2373); the return
 * value is the number of records that went live -> deleted, WAL entries included. */
size_t fso_fsvi_soft_delete(fso_fsvi *idx, const char *doc_id) {
    size_t deleted = 0, dl = strlen(doc_id);
    for (uint64_t r = 0; r < idx->record_count; ++r) {
        const char *p;
        uint32_t l = fso_fsvi_doc_id(idx, r, &p);
        if (l == dl && memcmp(p, doc_id, dl) == 0 && (fso_fsvi_flags(idx, r) & 1u) == 0) {
            fso_fsvi_set_flags(idx, r, (uint16_t)(fso_fsvi_flags(idx, r) | 1u));
            ++deleted;
        }
    }
    size_t w = 0;
    for (size_t i = 0; i < idx->wal_len; ++i) {
        if (strcmp(idx->wal_ids[i], doc_id) == 0) {
            free(idx->wal_ids[i]);
            free(idx->wal_vecs[i]);
            ++deleted;
        } else {
            idx->wal_ids[w] = idx->wal_ids[i];
            idx->wal_vecs[w] = idx->wal_vecs[i];
            ++w;
        }
    }
    idx->wal_len = w;
    return deleted;
}

/* VectorIndex::append -> append_batch_impl (lib.rs:2569-2720) for one entry. */
int fso_fsvi_append(fso_fsvi *idx, const char *doc_id, const float *vector, size_t len) {
    if (len != idx->dim) return FSO_ERR_DIMENSION_MISMATCH;
    for (size_t i = 0; i < len; ++i)
        if (!isfinite(vector[i])) return FSO_ERR_INVALID_CONFIG;
    if (!fso_vector_signal_usable(vector, len)) return FSO_ERR_INVALID_CONFIG;
    size_t dl = strlen(doc_id);
    if (dl > 0xffff) return FSO_ERR_INVALID_CONFIG;
    /* supersede older resident copies */
    size_t w = 0;
    for (size_t i = 0; i < idx->wal_len; ++i) {
        if (strcmp(idx->wal_ids[i], doc_id) == 0) {
            free(idx->wal_ids[i]);
            free(idx->wal_vecs[i]);
        } else {
            idx->wal_ids[w] = idx->wal_ids[i];
            idx->wal_vecs[w] = idx->wal_vecs[i];
            ++w;
        }
    }
    idx->wal_len = w;
    if (idx->wal_len == idx->wal_cap) {
        idx->wal_cap = idx->wal_cap ? idx->wal_cap * 2 : 8;
        idx->wal_ids = (char **)realloc(idx->wal_ids, idx->wal_cap * sizeof(char *));
        idx->wal_vecs = (float **)realloc(idx->wal_vecs, idx->wal_cap * sizeof(float *));
    }
    idx->wal_ids[idx->wal_len] = strdup(doc_id);
    idx->wal_vecs[idx->wal_len] = (float *)malloc(len * sizeof(float));
    memcpy(idx->wal_vecs[idx->wal_len], vector, len * sizeof(float));
    idx->wal_len++;
    /* tombstone the first live main row with this doc id */
    for (uint64_t r = 0; r < idx->record_count; ++r) {
        const char *p;
        uint32_t l = fso_fsvi_doc_id(idx, r, &p);
        if (l == dl && memcmp(p, doc_id, dl) == 0 && (fso_fsvi_flags(idx, r) & 1u) == 0) {
            fso_fsvi_set_flags(idx, r, (uint16_t)(fso_fsvi_flags(idx, r) | 1u));
            break;
        }
    }
    return FSO_OK;
}
uint8_t fso_fsvi_quantization(const fso_fsvi *idx) { return idx->quant; }
uint64_t fso_fsvi_record_count(const fso_fsvi *idx) { return idx->record_count; }
uint32_t fso_fsvi_dimension(const fso_fsvi *idx) { return idx->dim; }
uint64_t fso_fsvi_vectors_offset(const fso_fsvi *idx) { return idx->vectors_offset; }
const uint8_t *fso_fsvi_slab(const fso_fsvi *idx) { return idx->data + idx->vectors_offset; }
uint32_t fso_fsvi_doc_id(const fso_fsvi *idx, uint64_t row, const char **ptr) {
    const uint8_t *rec = idx->data + idx->records_offset + row * 16;
    *ptr = (const char *)(idx->data + idx->strings_offset + get32(rec + 8));
    return get16(rec + 12);
}
uint16_t fso_fsvi_flags(const fso_fsvi *idx, uint64_t row) {
    return get16(idx->data + idx->records_offset + row * 16 + 14);
}
void fso_fsvi_set_flags(fso_fsvi *idx, uint64_t row, uint16_t flags) {
    put16(idx->data + idx->records_offset + row * 16 + 14, flags);
}

/* search_top_k (search.rs:192-206, 426-494) -> resolve_sorted_entries (search.rs:1503-1558):
 * main scan (tombstones skipped), WAL entries scored with dot_product_f32_f32 and merged into the same
 * size-k selection with their WAL-tagged index (always greater than a main index, wal.rs:557-569), then
 * winners resolved: WAL hits get the virtual index record_count + i, main winners shadowed by a resident WAL
 * entry with the same doc id are dropped, and only the first (best) hit per doc id is kept. */
size_t fso_fsvi_search(const fso_fsvi *idx, const float *q, size_t k, int hreduce,
                       uint32_t *out_rows, float *out_scores) {
    uint64_t n = idx->record_count;
    if (k == 0 || (n == 0 && idx->wal_len == 0)) return 0;
    uint64_t words = (n + 63) / 64;
    uint64_t *live = (uint64_t *)calloc((size_t)(words ? words : 1), 8);
    for (uint64_t r = 0; r < n; ++r)
        if ((fso_fsvi_flags(idx, r) & 1u) == 0) live[r >> 6] |= 1ull << (r & 63);
    size_t cap = (k < n ? k : (size_t)n) + idx->wal_len + 1;
    entry_t *cand = (entry_t *)malloc(sizeof(entry_t) * cap);
    size_t nc = 0;
    if (n > 0) {
        size_t kk = k < n ? k : (size_t)n;
        uint32_t *rows = (uint32_t *)malloc(sizeof(uint32_t) * kk);
        float *scores = (float *)malloc(sizeof(float) * kk);
        size_t got = idx->quant == 1 ? fso_search_top_k(fso_fsvi_slab(idx), n, idx->dim, live, q, k, 10000, 1024, 1, 1,
                                                        hreduce, rows, scores)
                                     : fso_search_top_k_f32(fso_fsvi_slab(idx), n, idx->dim, live, q, k, 1, hreduce, rows,
                                                            scores);
        for (size_t i = 0; i < got; ++i) {
            cand[nc].row = rows[i];
            cand[nc].score = scores[i];
            ++nc;
        }
        free(rows);
        free(scores);
    }
    const uint64_t wal_tag = 1ull << 63;
    for (size_t i = 0; i < idx->wal_len; ++i) {
        float s = fso_dot_f32_f32(idx->wal_vecs[i], q, idx->dim, hreduce);
        if (!isfinite(s)) continue; /* search.rs:1466-1470 */
        cand[nc].row = wal_tag | i;
        cand[nc].score = s;
        ++nc;
    }
    qsort(cand, nc, sizeof(entry_t), cmp_best_first);
    if (nc > k) nc = k; /* the size-k heap keeps exactly the k best of main U wal */
    size_t outn = 0;
    const char **seen = (const char **)malloc(sizeof(char *) * (nc ? nc : 1));
    uint32_t *seen_len = (uint32_t *)malloc(sizeof(uint32_t) * (nc ? nc : 1));
    for (size_t i = 0; i < nc; ++i) {
        const char *di;
        uint32_t dl;
        uint32_t index;
        if (cand[i].row & wal_tag) {
            size_t wi = (size_t)(cand[i].row & ~wal_tag);
            di = idx->wal_ids[wi];
            dl = (uint32_t)strlen(di);
            index = (uint32_t)(n + wi);
        } else {
            if (fso_fsvi_flags(idx, cand[i].row) & 1u) continue;
            dl = fso_fsvi_doc_id(idx, cand[i].row, &di);
            int shadowed = 0;
            for (size_t w = 0; w < idx->wal_len && !shadowed; ++w)
                if (strlen(idx->wal_ids[w]) == dl && memcmp(idx->wal_ids[w], di, dl) == 0) shadowed = 1;
            if (shadowed) continue;
            index = (uint32_t)cand[i].row;
        }
        int dup = 0;
        for (size_t j = 0; j < outn && !dup; ++j)
            if (seen_len[j] == dl && memcmp(seen[j], di, dl) == 0) dup = 1;
        if (dup) continue;
        seen[outn] = di;
        seen_len[outn] = dl;
        out_rows[outn] = index;
        out_scores[outn] = cand[i].score;
        ++outn;
    }
    free(seen);
    free(seen_len);
    free(cand);
    free(live);
    return outn;
}

/* ------------------------------------------------------------------------- */
/* fixtures                                                                   */
/* ------------------------------------------------------------------------- */

/* search.rs:1823-1834: s=(i*2654435761)^(j*40503); s^=s>>13; ((s&0xffff)/65535)-0.5 */
float fso_fixture_hashmix(uint64_t i, uint64_t j) {
    uint64_t s = (i * 2654435761ull) ^ (j * 40503ull);
    s ^= s >> 13;
    return ((float)(s & 0xffffu) / 65535.0f) - 0.5f;
}

/* frankensearch/benches/fsvi_4bit_vs_incumbent.rs:66-76 */
void fso_raw_vector(uint64_t seed, uint32_t dim, float *out) {
    uint64_t state = seed | 1u;
    for (uint32_t d = 0; d < dim; ++d) {
        state ^= state << 13;
        state ^= state >> 7;
        state ^= state << 17;
        out[d] = (float)(state >> 40) / (float)(1ull << 23) - 1.0f;
    }
}

/* :78-86 — norm = sqrt(sum x*x) (left-to-right f32), divide when norm > 1e-12 */
void fso_normalize_bench(float *v, uint32_t dim) {
    float acc = 0.0f;
    for (uint32_t d = 0; d < dim; ++d) {
        float p = v[d] * v[d];
        acc = acc + p;
    }
    float norm = sqrtf(acc);
    if (norm > 1e-12f)
        for (uint32_t d = 0; d < dim; ++d) v[d] = v[d] / norm;
}

static void make_vector(const float *centroid, uint32_t dim, uint64_t noise_seed, float noise, float *scratch,
                        float *out) {
    fso_raw_vector(noise_seed, dim, scratch);
    for (uint32_t d = 0; d < dim; ++d) {
        float p = noise * scratch[d];
        out[d] = centroid[d] + p;
    }
    fso_normalize_bench(out, dim);
}

/* :344-365 — centroids normalize(raw_vector(0xc0000000+i)); row i = make_vector(i%clusters, i+1) */
void fso_clustered_corpus_f16(uint64_t row0, uint64_t n, uint32_t dim, uint32_t clusters, float noise,
                              uint16_t *out) {
    float *cent = (float *)malloc(sizeof(float) * (size_t)clusters * dim);
    float *scratch = (float *)malloc(sizeof(float) * dim);
    float *v = (float *)malloc(sizeof(float) * dim);
    for (uint32_t c = 0; c < clusters; ++c) {
        fso_raw_vector(0xc0000000ull + c, dim, cent + (size_t)c * dim);
        fso_normalize_bench(cent + (size_t)c * dim, dim);
    }
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t row = row0 + i;
        make_vector(cent + (size_t)(row % clusters) * dim, dim, row + 1, noise, scratch, v);
        fso_encode_f32_to_f16(v, dim, out + (size_t)i * dim);
    }
    free(cent);
    free(scratch);
    free(v);
}

void fso_clustered_query(uint64_t q, uint32_t dim, uint32_t clusters, float noise, float *out) {
    float *cent = (float *)malloc(sizeof(float) * dim);
    float *scratch = (float *)malloc(sizeof(float) * dim);
    fso_raw_vector(0xc0000000ull + (q % clusters), dim, cent);
    fso_normalize_bench(cent, dim);
    make_vector(cent, dim, 0xdead0000ull + q, noise, scratch, out);
    free(cent);
    free(scratch);
}

/* ------------------------------------------------------------------------- */
/* Model2Vec                                                                  */
/* ------------------------------------------------------------------------- */

/* embed_token_ids (embed/src/model2vec_embedder.rs:310-335), accumulate_model2vec_rows_base
 * (embed/src/simd.rs:273-289), finish_mean_pool_and_normalize (model2vec_embedder.rs:435-451). */
void fso_m2v_embed(const float *table, uint32_t vocab, uint32_t dim, const uint32_t *ids, size_t n_ids,
                   float *out) {
    for (uint32_t d = 0; d < dim; ++d) out[d] = 0.0f;
    size_t count = 0;
    for (size_t t = 0; t < n_ids; ++t) {
        if (ids[t] >= vocab) continue;
        const float *row = table + (size_t)ids[t] * dim;
        for (uint32_t d = 0; d < dim; ++d) out[d] = out[d] + row[d];
        ++count;
    }
    if (count == 0) return; /* zeros */
    float inv = 1.0f / (float)count;
    float norm_sq = 0.0f;
    for (uint32_t d = 0; d < dim; ++d) {
        out[d] = out[d] * inv;
        float p = out[d] * out[d];
        norm_sq = norm_sq + p;
    }
    if (isfinite(norm_sq) && norm_sq > 1.1920929e-7f) {
        float inv_norm = 1.0f / sqrtf(norm_sq);
        for (uint32_t d = 0; d < dim; ++d) out[d] = out[d] * inv_norm;
    } else {
        for (uint32_t d = 0; d < dim; ++d) out[d] = 0.0f;
    }
}
