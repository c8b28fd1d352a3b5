/*
 * vrod_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, see vrod_oracle.h).
 *
 * PARITY UNPINNED BY THE REFERENCE: vRod has no scan to restate
 * (src/command/types.rs:127-132 is empty, src/database/mod.rs:19-21 is todo!()).
 * What the reference does fix, and this file follows:
 *   - element type and row layout: Vec<Vec<f32>>, one row per vector, ids are the
 *     implicit row order            (src/utils/embeddings.rs:29, :55-61)
 *   - single-threaded execution:    Rc<RefCell<Database>> (src/command/types.rs:10)
 * Everything else is the build-defined spec of DESIGN.md "Scan spec" (SURVEY.md 8c).
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off -fno-fast-math: no FMA
 * contraction, no reassociation, so the loops below are evaluated exactly as written).
 */
#include "vrod_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ synthetic */

uint64_t orc_splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* Element (row, col) of stream `seed`: hash the element index under a key derived
 * from the seed (so streams with nearby seeds do not share hash inputs), add the
 * four 16-bit fields (Irwin-Hall, approximately Gaussian) and centre. */
int32_t orc_synth_int(uint64_t seed, uint64_t row, uint32_t dim, uint32_t col) {
    uint64_t key = orc_splitmix64(seed);
    uint64_t h = orc_splitmix64(key ^ (row * (uint64_t)dim + col));
    int32_t s = (int32_t)(h & 0xFFFF) + (int32_t)((h >> 16) & 0xFFFF) +
                (int32_t)((h >> 32) & 0xFFFF) + (int32_t)(h >> 48);
    return s - 131070;
}

void orc_synth_row_f32(uint64_t seed, uint64_t row, uint32_t dim, float *out) {
    /* sum of squares is an exact integer < 2^53: order-independent */
    double ss = 0.0;
    for (uint32_t j = 0; j < dim; ++j) {
        double v = (double)orc_synth_int(seed, row, dim, j);
        ss += v * v;
    }
    if (ss == 0.0) {
        for (uint32_t j = 0; j < dim; ++j) out[j] = 0.0f;
        return;
    }
    double nrm = sqrt(ss);
    for (uint32_t j = 0; j < dim; ++j)
        out[j] = (float)((double)orc_synth_int(seed, row, dim, j) / nrm);
}

typedef struct {
    uint64_t seed, first, n;
    uint32_t dim;
    float *out;
} synth_job;

static void *synth_worker(void *p) {
    synth_job *j = (synth_job *)p;
    for (uint64_t r = 0; r < j->n; ++r)
        orc_synth_row_f32(j->seed, j->first + r, j->dim, j->out + r * (uint64_t)j->dim);
    return NULL;
}

void orc_synth_rows_f32(uint64_t seed, uint64_t first_row, uint64_t n, uint32_t dim,
                        float *out, int threads) {
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > n) threads = n ? (int)n : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    synth_job *jobs = (synth_job *)malloc(sizeof(synth_job) * (size_t)threads);
    for (int t = 0; t < threads; ++t) {
        uint64_t a = n * (uint64_t)t / (uint64_t)threads, b = n * (uint64_t)(t + 1) / (uint64_t)threads;
        jobs[t].seed = seed;
        jobs[t].first = first_row + a;
        jobs[t].n = b - a;
        jobs[t].dim = dim;
        jobs[t].out = out + a * (uint64_t)dim;
        if (t + 1 < threads) pthread_create(&th[t], NULL, synth_worker, &jobs[t]);
    }
    synth_worker(&jobs[threads - 1]);
    for (int t = 0; t + 1 < threads; ++t) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}

/* ------------------------------------------------------------------ conversions */

uint16_t orc_f32_to_bf16(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x0040u); /* NaN stays NaN */
    u += 0x7FFFu + ((u >> 16) & 1u); /* round to nearest, ties to even */
    return (uint16_t)(u >> 16);
}

float orc_bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* ------------------------------------------------------------------ preparation */

void orc_prepare_row(const float *in, uint32_t dim, int dtype, int metric, float *out) {
    if (metric == ORC_METRIC_COSINE) {
        double ss = 0.0;
        for (uint32_t j = 0; j < dim; ++j) { /* left to right, fp64 */
            double v = (double)in[j];
            ss = ss + v * v; /* v*v is exact in fp64 for fp32 v, so FMA or not is the same */
        }
        if (ss == 0.0) {
            for (uint32_t j = 0; j < dim; ++j) out[j] = 0.0f;
        } else {
            double nrm = sqrt(ss);
            for (uint32_t j = 0; j < dim; ++j) out[j] = (float)((double)in[j] / nrm);
        }
    } else {
        for (uint32_t j = 0; j < dim; ++j) out[j] = in[j];
    }
    if (dtype == ORC_DTYPE_BF16)
        for (uint32_t j = 0; j < dim; ++j) out[j] = orc_bf16_to_f32(orc_f32_to_bf16(out[j]));
}

typedef struct {
    const float *in;
    float *out;
    uint64_t n;
    uint32_t dim;
    int dtype, metric;
} prep_job;

static void *prep_worker(void *p) {
    prep_job *j = (prep_job *)p;
    for (uint64_t r = 0; r < j->n; ++r)
        orc_prepare_row(j->in + r * (uint64_t)j->dim, j->dim, j->dtype, j->metric,
                        j->out + r * (uint64_t)j->dim);
    return NULL;
}

void orc_prepare_rows(const float *in, uint64_t n, uint32_t dim, int dtype, int metric,
                      float *out, int threads) {
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > n) threads = n ? (int)n : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    prep_job *jobs = (prep_job *)malloc(sizeof(prep_job) * (size_t)threads);
    for (int t = 0; t < threads; ++t) {
        uint64_t a = n * (uint64_t)t / (uint64_t)threads, b = n * (uint64_t)(t + 1) / (uint64_t)threads;
        jobs[t].in = in + a * (uint64_t)dim;
        jobs[t].out = out + a * (uint64_t)dim;
        jobs[t].n = b - a;
        jobs[t].dim = dim;
        jobs[t].dtype = dtype;
        jobs[t].metric = metric;
        if (t + 1 < threads) pthread_create(&th[t], NULL, prep_worker, &jobs[t]);
    }
    prep_worker(&jobs[threads - 1]);
    for (int t = 0; t + 1 < threads; ++t) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}

/* ------------------------------------------------------------------ canonical scores */

/* What `q.iter().zip(x).map(|(a, b)| a * b).sum::<f32>()` evaluates to: the product
 * rounded to fp32, then the add rounded to fp32, strictly left to right. */
float orc_dot_canonical(const float *q, const float *x, uint32_t dim) {
    float acc = 0.0f;
    for (uint32_t j = 0; j < dim; ++j) {
        float p = q[j] * x[j];
        acc = acc + p;
    }
    return acc;
}

float orc_l2_canonical(const float *q, const float *x, uint32_t dim) {
    float acc = 0.0f;
    for (uint32_t j = 0; j < dim; ++j) {
        float df = q[j] - x[j];
        float p = df * df;
        acc = acc + p;
    }
    return acc;
}

/* ------------------------------------------------------------------ ordering */

/* 1 if (sa, ia) ranks strictly before (sb, ib). NaN ranks last; ties -> smaller id. */
static int ranks_before(int metric, float sa, uint64_t ia, float sb, uint64_t ib) {
    int na = (sa != sa), nb = (sb != sb);
    if (na || nb) {
        if (na != nb) return nb; /* the non-NaN one first */
        return ia < ib;
    }
    if (sa != sb) return metric == ORC_METRIC_COSINE ? (sa > sb) : (sa < sb);
    return ia < ib;
}

typedef struct {
    uint32_t k, count;
    int metric;
    uint64_t *ids;
    float *scores;
} topk_list;

static void topk_push(topk_list *l, float s, uint64_t id) {
    if (l->k == 0) return;
    if (l->count == l->k &&
        !ranks_before(l->metric, s, id, l->scores[l->k - 1], l->ids[l->k - 1]))
        return;
    uint32_t pos = l->count < l->k ? l->count : l->k - 1;
    while (pos > 0 && ranks_before(l->metric, s, id, l->scores[pos - 1], l->ids[pos - 1])) {
        l->scores[pos] = l->scores[pos - 1];
        l->ids[pos] = l->ids[pos - 1];
        --pos;
    }
    l->scores[pos] = s;
    l->ids[pos] = id;
    if (l->count < l->k) l->count++;
}

static void topk_fill_rest(topk_list *l) {
    uint32_t nanbits = ORC_SCORE_NONE_BITS;
    for (uint32_t i = l->count; i < l->k; ++i) {
        l->ids[i] = ORC_ID_NONE;
        memcpy(&l->scores[i], &nanbits, 4);
    }
}

/* ------------------------------------------------------------------ scan */

static void scan_range(const float *corpus, uint64_t r0, uint64_t r1, uint32_t dim,
                       const float *queries, uint32_t nq, uint32_t k, int metric,
                       uint64_t id_offset, uint64_t *ids, float *scores) {
    for (uint32_t qi = 0; qi < nq; ++qi) {
        topk_list l = {k, 0, metric, ids + (uint64_t)qi * k, scores + (uint64_t)qi * k};
        const float *q = queries + (uint64_t)qi * dim;
        for (uint64_t r = r0; r < r1; ++r) {
            const float *x = corpus + r * (uint64_t)dim;
            float s = metric == ORC_METRIC_COSINE ? orc_dot_canonical(q, x, dim)
                                                  : orc_l2_canonical(q, x, dim);
            topk_push(&l, s, r + id_offset);
        }
        topk_fill_rest(&l);
    }
}

typedef struct {
    const float *corpus, *queries;
    uint64_t r0, r1, id_offset;
    uint32_t dim, nq, k;
    int metric;
    uint64_t *ids;
    float *scores;
} scan_job;

static void *scan_worker(void *p) {
    scan_job *j = (scan_job *)p;
    scan_range(j->corpus, j->r0, j->r1, j->dim, j->queries, j->nq, j->k, j->metric,
               j->id_offset, j->ids, j->scores);
    return NULL;
}

int orc_merge_topk(const uint64_t *ids, const float *scores, uint32_t n_lists, uint32_t nq,
                   uint32_t k, int metric, uint64_t *out_ids, float *out_scores) {
    for (uint32_t qi = 0; qi < nq; ++qi) {
        topk_list l = {k, 0, metric, out_ids + (uint64_t)qi * k, out_scores + (uint64_t)qi * k};
        for (uint32_t li = 0; li < n_lists; ++li) {
            const uint64_t *pi = ids + ((uint64_t)li * nq + qi) * k;
            const float *ps = scores + ((uint64_t)li * nq + qi) * k;
            for (uint32_t i = 0; i < k; ++i)
                if (pi[i] != ORC_ID_NONE) topk_push(&l, ps[i], pi[i]);
        }
        topk_fill_rest(&l);
    }
    return 0;
}

int orc_scan_topk(const float *corpus, uint64_t n, uint32_t dim, const float *queries,
                  uint32_t nq, uint32_t k, int metric, uint64_t id_offset,
                  uint64_t *out_ids, float *out_scores, int threads) {
    if (metric != ORC_METRIC_COSINE && metric != ORC_METRIC_L2) return 1;
    if (threads <= 1 || n < 2) {
        scan_range(corpus, 0, n, dim, queries, nq, k, metric, id_offset, out_ids, out_scores);
        return 0;
    }
    if ((uint64_t)threads > n) threads = (int)n;
    size_t per = (size_t)nq * k;
    uint64_t *tids = (uint64_t *)malloc(sizeof(uint64_t) * per * (size_t)threads);
    float *tsc = (float *)malloc(sizeof(float) * per * (size_t)threads);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    scan_job *jobs = (scan_job *)malloc(sizeof(scan_job) * (size_t)threads);
    if (!tids || !tsc || !th || !jobs) return 2;
    for (int t = 0; t < threads; ++t) {
        jobs[t].corpus = corpus;
        jobs[t].queries = queries;
        jobs[t].r0 = n * (uint64_t)t / (uint64_t)threads;
        jobs[t].r1 = n * (uint64_t)(t + 1) / (uint64_t)threads;
        jobs[t].id_offset = id_offset;
        jobs[t].dim = dim;
        jobs[t].nq = nq;
        jobs[t].k = k;
        jobs[t].metric = metric;
        jobs[t].ids = tids + per * (size_t)t;
        jobs[t].scores = tsc + per * (size_t)t;
        if (t + 1 < threads) pthread_create(&th[t], NULL, scan_worker, &jobs[t]);
    }
    scan_worker(&jobs[threads - 1]);
    for (int t = 0; t + 1 < threads; ++t) pthread_join(th[t], NULL);
    orc_merge_topk(tids, tsc, (uint32_t)threads, nq, k, metric, out_ids, out_scores);
    free(tids);
    free(tsc);
    free(th);
    free(jobs);
    return 0;
}
