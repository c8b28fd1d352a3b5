// kernels_select.hip -- top-k select, candidate-list compaction, final ordering and the
// exactness certificate, shard merge (gfx950).
//
// Fills the "top-k select" and "shard merge" slots of SURVEY.md 8a (rows a6, a7); the
// reference has nothing here (SearchSimilarCommand::execute is empty,
// src/command/types.rs:127-132).
//
// Every ordering in this file is on one 64-bit composite key
//        key = order_preserving_u32(score) << 32  |  ~row
// so that "larger key" == "better score, then smaller id" -- the spec's ordering
// (DESIGN.md "Scan spec" rule 5) -- and a select is a descending sort of keys.
// Blocks sort up to kSelectChunk keys in LDS (64 KB) with a bitonic network.
#include "vrod_common.h"
#include "vrod_kernels.h"

namespace vrod {

constexpr int kSortThreads = 1024;

// ------------------------------------------------------------------ block bitonic sort (descending)
// n is a power of two, n <= kSelectChunk.  All kSortThreads threads call this.
__device__ __forceinline__ void bitonic_sort_desc(uint64_t* __restrict__ a, uint32_t n) {
    const uint32_t tid = threadIdx.x;
    for (uint32_t k = 2; k <= n; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < (n >> 1); t += kSortThreads) {
                const uint32_t i = 2 * t - (t & (j - 1));
                const uint32_t l = i + j;
                const bool up = (i & k) == 0;
                const uint64_t x = a[i], y = a[l];
                if ((x < y) == up) { a[i] = y; a[l] = x; }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ uint32_t pow2_ceil(uint32_t n) {
    uint32_t p = 1;
    while (p < n) p <<= 1;
    return p;
}

// ------------------------------------------------------------------ chunked select
// grid = (nchunks, nq).  Chunk c of query q: elements [c*C, min((c+1)*C, n)).
// FROM_SCORES: element i is (score[q][i], row i).  Else: element i is key[q][i].
// Writes the chunk's best `kp` keys (descending, zero-padded) to out[q][c*kp ...].
template <int METRIC, bool FROM_SCORES>
__global__ __launch_bounds__(kSortThreads) void select_chunk_kernel(
    const float* __restrict__ scores, const uint64_t* __restrict__ keys_in, uint64_t in_ld,
    uint64_t n, uint32_t kp, uint64_t* __restrict__ out, uint64_t out_ld) {
    extern __shared__ __attribute__((aligned(16))) uint64_t skeys[];
    const uint32_t q = blockIdx.y;
    const uint64_t c0 = (uint64_t)blockIdx.x * kSelectChunk;
    const uint32_t cnt = (uint32_t)(n - c0 < kSelectChunk ? n - c0 : kSelectChunk);
    const uint32_t np2 = pow2_ceil(cnt < 2 ? 2 : cnt);
    for (uint32_t i = threadIdx.x; i < np2; i += kSortThreads) {
        uint64_t key = 0;
        if (i < cnt) {
            if constexpr (FROM_SCORES)
                key = make_key(score_key<METRIC>(scores[(uint64_t)q * in_ld + c0 + i]), (uint32_t)(c0 + i));
            else
                key = keys_in[(uint64_t)q * in_ld + c0 + i];
        }
        skeys[i] = key;
    }
    __syncthreads();
    bitonic_sort_desc(skeys, np2);
    uint64_t* o = out + (uint64_t)q * out_ld + (uint64_t)blockIdx.x * kp;
    for (uint32_t i = threadIdx.x; i < kp; i += kSortThreads) o[i] = i < np2 ? skeys[i] : 0ull;
}

static inline size_t sort_lds_bytes(uint64_t n) {
    uint64_t p = 2;
    while (p < n) p <<= 1;
    if (p > kSelectChunk) p = kSelectChunk;
    return (size_t)p * sizeof(uint64_t);
}

uint64_t launch_select_from_scores(const float* d_scores, uint64_t score_ld, uint64_t n, int nq,
                                   int metric, uint32_t kp, uint64_t* d_out, uint64_t out_ld,
                                   hipStream_t s) {
    const uint64_t nchunks = (n + kSelectChunk - 1) / kSelectChunk;
    dim3 grid((unsigned)nchunks, nq);
    const size_t lds = sort_lds_bytes(n);
    if (metric == M_COSINE)
        select_chunk_kernel<M_COSINE, true><<<grid, kSortThreads, lds, s>>>(d_scores, nullptr, score_ld, n, kp, d_out, out_ld);
    else
        select_chunk_kernel<M_L2, true><<<grid, kSortThreads, lds, s>>>(d_scores, nullptr, score_ld, n, kp, d_out, out_ld);
    return nchunks * kp;
}

uint64_t launch_select_from_keys(const uint64_t* d_in, uint64_t in_ld, uint64_t n, int nq,
                                 uint32_t kp, uint64_t* d_out, uint64_t out_ld, hipStream_t s) {
    const uint64_t nchunks = (n + kSelectChunk - 1) / kSelectChunk;
    dim3 grid((unsigned)nchunks, nq);
    select_chunk_kernel<M_COSINE, false><<<grid, kSortThreads, sort_lds_bytes(n), s>>>(nullptr, d_in, in_ld, n, kp, d_out, out_ld);
    return nchunks * kp;
}

// ------------------------------------------------------------------ keys -> candidates
// One block per query, n <= kSelectChunk keys.  Sort, emit kp rows + fast scores + T.
__global__ __launch_bounds__(kSortThreads) void keys_to_candidates_kernel(
    const uint64_t* __restrict__ keys, uint64_t key_ld, uint32_t n, int metric, uint32_t kp,
    uint32_t* __restrict__ cand_rows, float* __restrict__ cand_fast, float* __restrict__ T) {
    extern __shared__ __attribute__((aligned(16))) uint64_t skeys[];
    const uint32_t q = blockIdx.x;
    const uint32_t np2 = pow2_ceil(n < 2 ? 2 : n);
    for (uint32_t i = threadIdx.x; i < np2; i += kSortThreads)
        skeys[i] = i < n ? keys[(uint64_t)q * key_ld + i] : 0ull;
    __syncthreads();
    bitonic_sort_desc(skeys, np2);
    for (uint32_t i = threadIdx.x; i < kp; i += kSortThreads) {
        const uint64_t key = i < np2 ? skeys[i] : 0ull;
        cand_rows[(uint64_t)q * kp + i] = key ? key_row(key) : 0xFFFFFFFFu;
        cand_fast[(uint64_t)q * kp + i] = key ? key_to_score_rt(key_skey(key), metric) : __uint_as_float(kScoreNoneBits);
    }
    if (threadIdx.x == 0) {
        // kp-th candidate present -> rows left out all have a fast score no better than it
        const uint64_t last = kp <= np2 ? skeys[kp - 1] : 0ull;
        T[q] = last ? key_to_score_rt(key_skey(last), metric) : worst_score(metric);
    }
}

void launch_keys_to_candidates(const uint64_t* d_keys, uint64_t key_ld, uint64_t n, int nq,
                               int metric, uint32_t kp, uint32_t* d_cand_rows, float* d_cand_fast,
                               float* d_T, hipStream_t s) {
    keys_to_candidates_kernel<<<nq, kSortThreads, sort_lds_bytes(n), s>>>(d_keys, key_ld, (uint32_t)n, metric, kp, d_cand_rows, d_cand_fast, d_T);
}

// ------------------------------------------------------------------ candidate lists (MFMA path)
// lists[q][cap] of {fast score bits, row}; counts[q] may exceed cap (overflow -> status bit 1).
__global__ __launch_bounds__(kSortThreads) void list_compact_kernel(
    uint2* __restrict__ lists, uint32_t* __restrict__ counts, uint32_t cap, int metric,
    uint32_t keep, float* __restrict__ thr, uint32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) uint64_t skeys[];
    const uint32_t q = blockIdx.x;
    const uint32_t c = counts[q];
    const uint32_t n = c < cap ? c : cap;
    uint2* l = lists + (uint64_t)q * cap;
    const uint32_t np2 = pow2_ceil(n < 2 ? 2 : n);
    for (uint32_t i = threadIdx.x; i < np2; i += kSortThreads) {
        uint64_t key = 0;
        if (i < n) {
            const uint2 e = l[i];
            key = make_key(score_key_rt(__uint_as_float(e.x), metric), e.y);
        }
        skeys[i] = key;
    }
    __syncthreads();
    bitonic_sort_desc(skeys, np2);
    const uint32_t m = n < keep ? n : keep;
    for (uint32_t i = threadIdx.x; i < m; i += kSortThreads) {
        const uint64_t key = skeys[i];
        l[i] = make_uint2(__float_as_uint(key_to_score_rt(key_skey(key), metric)), key_row(key));
    }
    if (threadIdx.x == 0) {
        counts[q] = m;
        thr[q] = n >= keep ? key_to_score_rt(key_skey(skeys[keep - 1]), metric) : worst_score(metric);
        if (c > cap) atomicOr(&status[q], 2u);
    }
}

void launch_list_compact(uint2* d_lists, uint32_t* d_counts, uint32_t cap, int nq, int metric,
                         uint32_t keep, float* d_thr, uint32_t* d_status, hipStream_t s) {
    list_compact_kernel<<<nq, kSortThreads, sort_lds_bytes(cap), s>>>(d_lists, d_counts, cap, metric, keep, d_thr, d_status);
}

__global__ __launch_bounds__(256) void list_to_candidates_kernel(
    const uint2* __restrict__ lists, const uint32_t* __restrict__ counts, uint32_t cap,
    uint32_t kp, uint32_t* __restrict__ cand_rows, float* __restrict__ cand_fast) {
    const uint32_t q = blockIdx.x;
    const uint32_t n = counts[q] < kp ? counts[q] : kp;
    for (uint32_t i = threadIdx.x; i < kp; i += blockDim.x) {
        uint2 e = make_uint2(kScoreNoneBits, 0xFFFFFFFFu);
        if (i < n) e = lists[(uint64_t)q * cap + i];
        cand_rows[(uint64_t)q * kp + i] = e.y;
        cand_fast[(uint64_t)q * kp + i] = __uint_as_float(e.x);
    }
}

void launch_list_to_candidates(const uint2* d_lists, const uint32_t* d_counts, uint32_t cap, int nq,
                               int metric, uint32_t kp, uint32_t* d_cand_rows, float* d_cand_fast,
                               float* d_T, hipStream_t s) {
    (void)metric; (void)d_T;  // T[q] is the thr[] written by the last list_compact(keep = kp)
    list_to_candidates_kernel<<<nq, 256, 0, s>>>(d_lists, d_counts, cap, kp, d_cand_rows, d_cand_fast);
}

// ------------------------------------------------------------------ final ordering + certificate
// One block per query.  Sort candidates by (canonical score, id); emit the best k.
// Certificate (DESIGN.md "Exactness certificate"): every row NOT among the candidates has a
// fast score no better than T.  With |fast - canonical| <= eps_abs + eps_rel*|T| the k-th
// canonical score s_k beats every left-out row strictly iff
//     COSINE: s_k > T + eps          L2: s_k < T - eps
// T == worst score means nothing was left out.  status bit 0 is set when NOT certified.
__global__ __launch_bounds__(kSortThreads) void final_topk_kernel(
    const uint32_t* __restrict__ cand_rows, const float* __restrict__ cand_fast,
    const float* __restrict__ cand_canon, const float* __restrict__ T, uint32_t kp, uint32_t k,
    int metric, uint64_t id_offset, float eps_abs, float eps_rel, uint64_t* __restrict__ out_ids,
    float* __restrict__ out_scores, uint32_t* __restrict__ status, float* __restrict__ max_err) {
    extern __shared__ __attribute__((aligned(16))) uint64_t skeys[];
    const uint32_t q = blockIdx.x;
    const uint32_t np2 = pow2_ceil(kp < 2 ? 2 : kp);
    float err = 0.0f;
    for (uint32_t i = threadIdx.x; i < np2; i += kSortThreads) {
        uint64_t key = 0;
        if (i < kp) {
            const uint32_t row = cand_rows[(uint64_t)q * kp + i];
            if (row != 0xFFFFFFFFu) {
                const float c = cand_canon[(uint64_t)q * kp + i];
                key = make_key(score_key_rt(c, metric), row);
                const float e = __builtin_fabsf(cand_fast[(uint64_t)q * kp + i] - c);
                if (e == e) err = __builtin_fmaxf(err, e);
            }
        }
        skeys[i] = key;
    }
    for (int o = 32; o > 0; o >>= 1) err = __builtin_fmaxf(err, __shfl_xor(err, o));
    if ((threadIdx.x & 63) == 0 && err > 0.0f) atomicMax((uint32_t*)max_err, __float_as_uint(err));
    __syncthreads();
    bitonic_sort_desc(skeys, np2);
    for (uint32_t i = threadIdx.x; i < k; i += kSortThreads) {
        const uint64_t key = i < np2 ? skeys[i] : 0ull;
        out_ids[(uint64_t)q * k + i] = key ? (uint64_t)key_row(key) + id_offset : UINT64_MAX;
        out_scores[(uint64_t)q * k + i] = key ? key_to_score_rt(key_skey(key), metric) : __uint_as_float(kScoreNoneBits);
    }
    if (threadIdx.x == 0) {
        const float t = T[q];
        bool ok;
        if (t == worst_score(metric)) {
            ok = true;  // every row of the shard was a candidate
        } else {
            const uint64_t kk = (k >= 1 && k <= np2) ? skeys[k - 1] : 0ull;
            if (!kk) {
                ok = false;  // fewer than k candidates although rows were left out
            } else {
                const float sk = key_to_score_rt(key_skey(kk), metric);
                const float eps = eps_abs + eps_rel * __builtin_fabsf(t);
                ok = metric == M_COSINE ? (sk > t + eps) : (sk < t - eps);
            }
        }
        if (!ok) atomicOr(&status[q], 1u);
    }
}

void launch_final_topk(const uint32_t* d_cand_rows, const float* d_cand_fast,
                       const float* d_cand_canon, const float* d_T, int nq, uint32_t kp, uint32_t k,
                       int metric, uint64_t nrows_total, uint64_t id_offset, float eps_abs,
                       float eps_rel, uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_status,
                       float* d_max_err, hipStream_t s) {
    (void)nrows_total;
    if (!nq) return;
    final_topk_kernel<<<nq, kSortThreads, sort_lds_bytes(kp), s>>>(d_cand_rows, d_cand_fast, d_cand_canon, d_T, kp, k, metric, id_offset, eps_abs, eps_rel, d_out_ids, d_out_scores, d_status, d_max_err);
}

// ------------------------------------------------------------------ exact path output
// keys: n <= kSelectChunk composite keys built from CANONICAL scores of one query.
__global__ __launch_bounds__(kSortThreads) void keys_to_output_kernel(
    const uint64_t* __restrict__ keys, uint32_t n, int metric, uint32_t k, uint64_t id_offset,
    uint64_t* __restrict__ out_ids, float* __restrict__ out_scores) {
    extern __shared__ __attribute__((aligned(16))) uint64_t skeys[];
    const uint32_t np2 = pow2_ceil(n < 2 ? 2 : n);
    for (uint32_t i = threadIdx.x; i < np2; i += kSortThreads) skeys[i] = i < n ? keys[i] : 0ull;
    __syncthreads();
    bitonic_sort_desc(skeys, np2);
    for (uint32_t i = threadIdx.x; i < k; i += kSortThreads) {
        const uint64_t key = i < np2 ? skeys[i] : 0ull;
        out_ids[i] = key ? (uint64_t)key_row(key) + id_offset : UINT64_MAX;
        out_scores[i] = key ? key_to_score_rt(key_skey(key), metric) : __uint_as_float(kScoreNoneBits);
    }
}

void launch_keys_to_output(const uint64_t* d_keys, uint64_t n, int metric, uint32_t k,
                           uint64_t id_offset, uint64_t* d_out_ids, float* d_out_scores,
                           hipStream_t s) {
    keys_to_output_kernel<<<1, kSortThreads, sort_lds_bytes(n), s>>>(d_keys, (uint32_t)n, metric, k, id_offset, d_out_ids, d_out_scores);
}

// ------------------------------------------------------------------ shard merge (after the all-gather)
// Each of the n_lists inputs is sorted best-first with unique ids, so an element's output
// position is its own index plus, for every other list, the number of that list's
// elements that rank before it (binary search).  One block per query.
__device__ __forceinline__ bool ranks_before(uint32_t ka, uint64_t ia, uint32_t kb, uint64_t ib) {
    return ka > kb || (ka == kb && ia < ib);
}

__global__ __launch_bounds__(256) void merge_topk_kernel(int metric, const uint64_t* __restrict__ ids,
                                                         const float* __restrict__ scores,
                                                         uint32_t n_lists, uint32_t nq, uint32_t k,
                                                         uint64_t* __restrict__ out_ids,
                                                         float* __restrict__ out_scores) {
    const uint32_t q = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
        out_ids[(uint64_t)q * k + i] = UINT64_MAX;
        out_scores[(uint64_t)q * k + i] = __uint_as_float(kScoreNoneBits);
    }
    __syncthreads();
    const uint32_t total = n_lists * k;
    for (uint32_t e = threadIdx.x; e < total; e += blockDim.x) {
        const uint32_t li = e / k, pi = e - li * k;
        const uint64_t base = ((uint64_t)li * nq + q) * k;
        const uint64_t id = ids[base + pi];
        if (id == UINT64_MAX) continue;
        const float sc = scores[base + pi];
        const uint32_t key = score_key_rt(sc, metric);
        uint32_t rank = pi;
        for (uint32_t m = 0; m < n_lists; ++m) {
            if (m == li) continue;
            const uint64_t mb = ((uint64_t)m * nq + q) * k;
            uint32_t lo = 0, hi = k;  // first index whose element does NOT rank before (key, id)
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                const uint64_t mid_id = ids[mb + mid];
                const bool before = mid_id != UINT64_MAX &&
                                    ranks_before(score_key_rt(scores[mb + mid], metric), mid_id, key, id);
                if (before) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            out_ids[(uint64_t)q * k + rank] = id;
            out_scores[(uint64_t)q * k + rank] = sc;
        }
    }
}

void launch_merge_topk(int metric, const uint64_t* d_ids, const float* d_scores, uint32_t n_lists,
                       uint32_t nq, uint32_t k, uint64_t* d_out_ids, float* d_out_scores,
                       hipStream_t s) {
    if (!nq || !k) return;
    merge_topk_kernel<<<nq, 256, 0, s>>>(metric, d_ids, d_scores, n_lists, nq, k, d_out_ids, d_out_scores);
}

}  // namespace vrod
