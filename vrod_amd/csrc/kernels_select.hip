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
// n is a power of two, 2 <= n <= kSelectChunk.  All kSortThreads threads call this.
// Each of the W active waves owns a contiguous segment of S = n/W >= 128 keys.  A stage whose
// stride j is < S only pairs keys inside one segment, so it needs no block barrier: LDS
// operations of one wave execute in order, a compiler fence is enough.  Only the log2(W)
// largest strides of each merge step are block-wide (10 of 91 stages at n = 8192).
template <uint32_t NT = kSortThreads>
__device__ __forceinline__ void bitonic_sort_desc(uint64_t* __restrict__ a, uint32_t n) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr uint32_t kWaves = NT / 64;
    const uint32_t W = n / 128 < kWaves ? (n / 128 ? n / 128 : 1) : kWaves;  // active waves
    const uint32_t S = n / W;                                                 // keys per segment
    const bool active = wave < W;
    bool dirty_local = false;  // local stages ran since the last block barrier
    for (uint32_t k = 2; k <= n; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            if (j >= S) {
                // block-wide stage
                if (dirty_local) { __syncthreads(); dirty_local = false; }
                for (uint32_t t = tid; t < (n >> 1); t += NT) {
                    const uint32_t i = 2 * t - (t & (j - 1));
                    const uint32_t l = i + j;
                    const bool up = (i & k) == 0;
                    const uint64_t x = a[i], y = a[l];
                    if ((x < y) == up) { a[i] = y; a[l] = x; }
                }
                __syncthreads();
            } else {
                // segment-local stage: wave `wave` handles compare-exchanges of its own segment
                if (active) {
                    const uint32_t base = wave * S;
                    for (uint32_t t = lane; t < (S >> 1); t += 64) {
                        const uint32_t i = base + 2 * t - (t & (j - 1));
                        const uint32_t l = i + j;
                        const bool up = (i & k) == 0;
                        const uint64_t x = a[i], y = a[l];
                        if ((x < y) == up) { a[i] = y; a[l] = x; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                dirty_local = true;
            }
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------ small sets: rank sort
// Keys are unique (they embed the row), so an element's position in descending order is the
// number of keys larger than it.  Every lane reads the same LDS word per step (broadcast,
// conflict-free) -- no barriers, n^2/threads compares: the cheaper sort below ~2K keys.
// a[0..n) unsorted in LDS, tmp[0..n) receives the sorted keys.  Ends with a barrier.
constexpr uint32_t kRankSortMax = 128;
__device__ __forceinline__ void rank_sort_desc(const uint64_t* __restrict__ a, uint64_t* __restrict__ tmp, uint32_t n) {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t mine = a[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) rank += a[j] > mine;
        // duplicates can only be padding zeros: give them distinct slots at the end
        if (mine == 0ull) { for (uint32_t j = 0; j < i; ++j) rank += a[j] == 0ull; }
        tmp[rank] = mine;
    }
    __syncthreads();
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
// One block per query, n <= kSelectChunk keys (n = min(cnt[q], n_max) when cnt is given).
// Sort, emit kp rows + fast scores + T.
__global__ __launch_bounds__(kSortThreads) void keys_to_candidates_kernel(
    const uint64_t* __restrict__ keys, uint64_t key_ld, uint32_t n_max, const uint32_t* __restrict__ cnt,
    int metric, uint32_t kp, uint32_t* __restrict__ cand_rows, float* __restrict__ cand_fast,
    float* __restrict__ T) {
    extern __shared__ __attribute__((aligned(16))) uint64_t skeys[];
    const uint32_t q = blockIdx.x;
    uint32_t n = n_max;
    if (cnt) n = cnt[q] < n_max ? cnt[q] : n_max;
    const uint64_t* sorted;
    uint32_t nsorted;
    if (n <= kRankSortMax) {
        for (uint32_t i = threadIdx.x; i < n; i += kSortThreads) skeys[i] = keys[(uint64_t)q * key_ld + i];
        __syncthreads();
        rank_sort_desc(skeys, skeys + kRankSortMax, n);
        sorted = skeys + kRankSortMax;
        nsorted = n;
    } else {
        const uint32_t np2 = pow2_ceil(n);
        for (uint32_t i = threadIdx.x; i < np2; i += kSortThreads)
            skeys[i] = i < n ? keys[(uint64_t)q * key_ld + i] : 0ull;
        __syncthreads();
        bitonic_sort_desc(skeys, np2);
        sorted = skeys;
        nsorted = np2;
    }
    for (uint32_t i = threadIdx.x; i < kp; i += kSortThreads) {
        const uint64_t key = i < nsorted ? sorted[i] : 0ull;
        cand_rows[(uint64_t)q * kp + i] = key ? key_row(key) : 0xFFFFFFFFu;
        cand_fast[(uint64_t)q * kp + i] = key ? key_to_score_rt(key_skey(key), metric) : __uint_as_float(kScoreNoneBits);
    }
    if (threadIdx.x == 0) {
        // kp-th candidate present -> rows left out all have a fast score no better than it
        const uint64_t last = kp <= nsorted ? sorted[kp - 1] : 0ull;
        T[q] = last ? key_to_score_rt(key_skey(last), metric) : worst_score(metric);
    }
}

void launch_keys_to_candidates(const uint64_t* d_keys, uint64_t key_ld, uint64_t n, int nq,
                               int metric, uint32_t kp, uint32_t* d_cand_rows, float* d_cand_fast,
                               float* d_T, const uint32_t* d_cnt, hipStream_t s) {
    // LDS: bitonic needs pow2(n) keys; the rank sort needs 2 * kRankSortMax
    size_t lds = sort_lds_bytes(n);
    if (lds < 2 * kRankSortMax * sizeof(uint64_t)) lds = 2 * kRankSortMax * sizeof(uint64_t);
    keys_to_candidates_kernel<<<nq, kSortThreads, lds, s>>>(d_keys, key_ld, (uint32_t)n, d_cnt, metric, kp, d_cand_rows, d_cand_fast, d_T);
}

// ------------------------------------------------------------------ radix-select step 2 (stream path)
// grid = (blocks over the score array, nq).  Every block re-derives the cut bin from the
// global histogram (a few KB, L2-resident) -- cheaper than one more launch -- and then
// compacts its slice: rows whose key is in the cut bin or a better one.
constexpr uint32_t kCompactSlice = 8192;
template <int METRIC>
__global__ __launch_bounds__(256) void hist_compact_kernel(const float* __restrict__ scores, uint64_t score_ld,
                                                           uint64_t n, const uint32_t* __restrict__ ghist,
                                                           int hist_bits, uint32_t kp, uint64_t* __restrict__ keys,
                                                           uint32_t cap, uint32_t* __restrict__ cnt,
                                                           uint32_t* __restrict__ status) {
    __shared__ uint32_t ctl[8];
    const uint32_t q = blockIdx.y;
    const int nbins = 1 << hist_bits;
    block_find_cut_bin(ghist + (size_t)q * nbins, nbins, kp, ctl);
    const uint32_t cut = ctl[4], n_ge = ctl[5];
    if (n_ge > cap) {  // too many rows share the cut bin (heavy ties): the exact path takes over
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&status[q], 2u);
        return;
    }
    const uint32_t thr_key = cut << (32 - hist_bits);
    const uint64_t i0 = (uint64_t)blockIdx.x * kCompactSlice;
    const float* sc = scores + (uint64_t)q * score_ld;
    const int lane = threadIdx.x & 63;
    for (uint32_t off = threadIdx.x; off < kCompactSlice; off += 256) {
        const uint64_t i = i0 + off;
        uint32_t skey = 0;
        bool take = false;
        if (i < n) {
            skey = score_key<METRIC>(sc[i]);
            take = skey >= thr_key;
        }
        const unsigned long long m = __ballot(take);
        if (m) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&cnt[q], (uint32_t)__popcll(m));
            base = __shfl(base, 0);
            if (take) {
                const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (pos < cap) keys[(uint64_t)q * cap + pos] = make_key(skey, (uint32_t)i);
            }
        }
    }
}

void launch_hist_compact(const float* d_scores, uint64_t score_ld, uint64_t n, int nq, int metric,
                         const uint32_t* d_hist, int hist_bits, uint32_t kp, uint64_t* d_keys,
                         uint32_t cap, uint32_t* d_cnt, uint32_t* d_status, hipStream_t s) {
    dim3 grid((unsigned)((n + kCompactSlice - 1) / kCompactSlice), nq);
    if (metric == M_COSINE)
        hist_compact_kernel<M_COSINE><<<grid, 256, 0, s>>>(d_scores, score_ld, n, d_hist, hist_bits, kp, d_keys, cap, d_cnt, d_status);
    else
        hist_compact_kernel<M_L2><<<grid, 256, 0, s>>>(d_scores, score_ld, n, d_hist, hist_bits, kp, d_keys, cap, d_cnt, d_status);
}

// ------------------------------------------------------------------ candidate lists (MFMA path)
// lists[q][cap] of {fast score bits, row}; counts[q] may exceed cap (overflow -> status bit 1).
// One 256-thread block per query with an LDS window of `wsize` keys (2048 = 16 KB when
// keep <= 1024 -- lists normally hold a few hundred entries -- else 8192): the best `keep` so
// far stay at the front of the window and the list streams through the rest of it chunk by
// chunk, so any count <= cap is handled.  Requires keep <= wsize / 2.
constexpr uint32_t kCompactThreads = 256;
__global__ __launch_bounds__(kCompactThreads) void list_compact_kernel(
    uint2* __restrict__ lists, uint32_t* __restrict__ counts, uint32_t cap, int metric,
    uint32_t keep, uint32_t wsize, float* __restrict__ thr, uint32_t* __restrict__ status,
    uint32_t* __restrict__ cand_rows, float* __restrict__ cand_fast, float* __restrict__ T) {
    extern __shared__ __attribute__((aligned(16))) uint64_t win[];
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const uint32_t c = counts[q];
    const uint32_t n = c < cap ? c : cap;
    uint2* l = lists + (uint64_t)q * cap;
    const uint32_t kk = keep;
    uint32_t have = 0;          // sorted best-so-far at win[0..have)
    uint32_t done = 0;
    while (done < n) {
        const uint32_t room = wsize - have;
        const uint32_t take = n - done < room ? n - done : room;
        for (uint32_t i = tid; i < take; i += kCompactThreads) {
            const uint2 e = l[done + i];
            win[have + i] = make_key(score_key_rt(__uint_as_float(e.x), metric), e.y);
        }
        const uint32_t m = have + take;
        const uint32_t np2 = pow2_ceil(m < 2 ? 2 : m);
        for (uint32_t i = m + tid; i < np2; i += kCompactThreads) win[i] = 0ull;
        __syncthreads();
        bitonic_sort_desc<kCompactThreads>(win, np2);
        have = m < kk ? m : kk;
        done += take;
        __syncthreads();
    }
    for (uint32_t i = tid; i < have; i += kCompactThreads) {
        const uint64_t key = win[i];
        l[i] = make_uint2(__float_as_uint(key_to_score_rt(key_skey(key), metric)), key_row(key));
    }
    // last compaction of a search: also emit the candidate arrays (rows, fast scores, T)
    if (cand_rows) {
        for (uint32_t i = tid; i < keep; i += kCompactThreads) {
            const uint64_t key = i < have ? win[i] : 0ull;
            cand_rows[(uint64_t)q * keep + i] = i < have ? key_row(key) : 0xFFFFFFFFu;
            cand_fast[(uint64_t)q * keep + i] = i < have ? key_to_score_rt(key_skey(key), metric) : __uint_as_float(kScoreNoneBits);
        }
    }
    if (tid == 0) {
        counts[q] = have;
        float t = thr[q];
        if (n >= keep && have >= keep) { t = key_to_score_rt(key_skey(win[keep - 1]), metric); thr[q] = t; }  // else: unchanged
        if (T) T[q] = t;
        if (c > cap) atomicOr(&status[q], 2u);
    }
}

void launch_list_compact(uint2* d_lists, uint32_t* d_counts, uint32_t cap, int nq, int metric,
                         uint32_t keep, float* d_thr, uint32_t* d_status, uint32_t* d_cand_rows,
                         float* d_cand_fast, float* d_T, hipStream_t s) {
    const uint32_t wsize = keep <= 1024 ? 2048u : kSelectChunk;
    list_compact_kernel<<<nq, kCompactThreads, (size_t)wsize * 8, s>>>(d_lists, d_counts, cap, metric, keep, wsize, d_thr, d_status,
                                                                     d_cand_rows, d_cand_fast, d_T);
}

// ------------------------------------------------------------------ sample pass: exact j-th best
// One block per query over its n sample scores (L2-resident).  j is tiny next to n, so:
//   pass 1: every thread takes the best key of its own strided slice; the j-th largest of
//           those 256 keys is a LOWER bound L of the answer (they are j distinct elements);
//   pass 2: the few keys >= L are collected in LDS (expected ~j..3j of them);
//   then the exact j-th best among them (rank by counting).
// If more than kSampleListCap keys pass (heavy ties) the block falls back to three radix
// passes (12 + 12 + 8 bits) with LDS histograms.
constexpr uint32_t kSampleListCap = 2048;
template <int METRIC>
__global__ __launch_bounds__(256) void sample_select_kernel(const float* __restrict__ scores, uint64_t score_ld,
                                                            uint32_t n, uint32_t j, float* __restrict__ thr) {
    __shared__ __attribute__((aligned(16))) uint32_t hist[4096];   // keys [<= 2048] / pass-1 maxima [256] + candidate list [2048] / radix histogram
    __shared__ uint32_t ctl[8];
    const float* sc = scores + (uint64_t)blockIdx.x * score_ld;
    const uint32_t tid = threadIdx.x;
    if (j <= 256) {
        // (rows of the dense block start 16-B aligned and are padded to 256 floats: 16-B loads,
        //  8 of them in flight per thread; elements at or beyond n are masked, not trusted)
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 none4 = {__uint_as_float(kScoreNoneBits), __uint_as_float(kScoreNoneBits), __uint_as_float(kScoreNoneBits), __uint_as_float(kScoreNoneBits)};
        const uint32_t n4 = (n + 3) / 4;
        auto load4 = [&](uint32_t i4) -> f4 {
            if (i4 >= n4) return none4;
            f4 v = *reinterpret_cast<const f4*>(sc + (uint64_t)i4 * 4);
            if (i4 * 4 + 3 >= n) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (i4 * 4 + e >= n) v[e] = __uint_as_float(kScoreNoneBits);
            }
            return v;
        };
        uint32_t best = 0;
        for (uint32_t i0 = 0; i0 < n4; i0 += 256 * 8) {
            f4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = load4(i0 + u * 256 + tid);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const uint32_t k = score_key<METRIC>(v[u][e]); best = k > best ? k : best; }
        }
        hist[tid] = best;
        if (tid == 0) ctl[6] = 0u;
        __syncthreads();
        // rank of my maximum among the 256 (ties by thread id): the one with rank j-1 is L
        uint32_t rank = 0;
        for (uint32_t t = 0; t < 256; ++t) { const uint32_t o = hist[t]; rank += (o > best) || (o == best && t < tid); }
        if (rank == j - 1) ctl[7] = best;
        __syncthreads();
        const uint32_t L = ctl[7];
        uint32_t* list = hist + 1024;   // [kSampleListCap]
        for (uint32_t i0 = 0; i0 < n4; i0 += 256 * 8) {
            f4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = load4(i0 + u * 256 + tid);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t k = score_key<METRIC>(v[u][e]);   // NaN padding -> key 0 < L
                    if (k >= L && k != 0u) { const uint32_t p = atomicAdd(&ctl[6], 1u); if (p < kSampleListCap) list[p] = k; }
                }
        }
        __syncthreads();
        const uint32_t m = ctl[6];
        if (m <= kSampleListCap) {
            // exact j-th largest of list[0..m): count keys greater (ties: earlier index first)
            for (uint32_t i = tid; i < m; i += 256) {
                const uint32_t mine = list[i];
                uint32_t r = 0;
                for (uint32_t t = 0; t < m; ++t) { const uint32_t o = list[t]; r += (o > mine) || (o == mine && t < i); }
                if (r == j - 1) thr[blockIdx.x] = key_to_score_rt(mine, METRIC);
            }
            return;
        }
        __syncthreads();
    }
    uint32_t prefix = 0, prefix_bits = 0, remaining = j;
#pragma unroll 1
    for (int pass = 0; pass < 3; ++pass) {
        const uint32_t bits = pass == 2 ? 8u : 12u;
        const uint32_t nb = 1u << bits, shift = 32u - prefix_bits - bits;
        for (uint32_t i = tid; i < nb; i += 256) hist[i] = 0u;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += 256) {
            const uint32_t key = score_key<METRIC>(sc[i]);
            if (prefix_bits == 0 || (key >> (32u - prefix_bits)) == prefix) atomicAdd(&hist[(key >> shift) & (nb - 1)], 1u);
        }
        __syncthreads();
        block_find_cut_bin(hist, (int)nb, remaining, ctl);
        const uint32_t cut = ctl[4], cum = ctl[5];
        remaining -= cum - hist[cut];  // rows in strictly better bins are accounted for
        prefix = (prefix << bits) | cut;
        prefix_bits += bits;
        __syncthreads();
    }
    if (tid == 0) thr[blockIdx.x] = key_to_score_rt(prefix, METRIC);
}

// The grouped sample pass leaves at most a few hundred group bests per query (512 at batch 1024): one WAVE per query, the
// keys in registers (up to 16 per lane), the j-th largest built bit by bit from the top -- 32 rounds of compare + ballot +
// popcount, no LDS, no barrier, one dependent load.  Same value as the block kernel above (the exact j-th largest key;
// nothing written when fewer than j keys are valid), 7 us instead of 20 in front of every batch's first filtered stage.
constexpr uint32_t kSampleWaveMax = 1024;
template <int METRIC>
__global__ __launch_bounds__(256) void sample_select_wave_kernel(const float* __restrict__ scores, uint64_t score_ld, uint32_t n,
                                                                 uint32_t j, uint32_t nq, float* __restrict__ thr) {
    const uint32_t q = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (q >= nq) return;
    const float* sc = scores + (uint64_t)q * score_ld;
    uint32_t key[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const uint32_t i = (uint32_t)u * 64u + lane;
        key[u] = i < n ? score_key<METRIC>(sc[i]) : 0u;   // NaN padding and the slots past n: key 0, never counted
    }
    uint32_t valid = 0;
#pragma unroll
    for (int u = 0; u < 16; ++u) valid += (uint32_t)__builtin_popcountll(__ballot(key[u] != 0u));
    if (valid < j) return;
    uint32_t prefix = 0;
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = prefix | (1u << bit);
        uint32_t cnt = 0;
#pragma unroll
        for (int u = 0; u < 16; ++u) cnt += (uint32_t)__builtin_popcountll(__ballot(key[u] >= cand));
        if (cnt >= j) prefix = cand;     // at least j keys are >= cand: the j-th largest has this bit set
    }
    if (lane == 0) thr[q] = key_to_score_rt(prefix, METRIC);
}

void launch_sample_select(const float* d_scores, uint64_t score_ld, uint32_t n_sample, int nq,
                          int metric, uint32_t j, float* d_thr, hipStream_t s) {
    if (!nq) return;
    if (n_sample <= kSampleWaveMax && j >= 1) {
        const int grid = (nq + 3) / 4;
        if (metric == M_COSINE) sample_select_wave_kernel<M_COSINE><<<grid, 256, 0, s>>>(d_scores, score_ld, n_sample, j, (uint32_t)nq, d_thr);
        else sample_select_wave_kernel<M_L2><<<grid, 256, 0, s>>>(d_scores, score_ld, n_sample, j, (uint32_t)nq, d_thr);
        return;
    }
    if (metric == M_COSINE) sample_select_kernel<M_COSINE><<<nq, 256, 0, s>>>(d_scores, score_ld, n_sample, j, d_thr);
    else sample_select_kernel<M_L2><<<nq, 256, 0, s>>>(d_scores, score_ld, n_sample, j, d_thr);
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
    int metric, IdMap idmap, int eps_mode, float eps_c, const uint32_t* __restrict__ max_qn2_bits,
    const uint32_t* __restrict__ max_xn2_bits, uint64_t* __restrict__ out_ids,
    float* __restrict__ out_scores, uint32_t* __restrict__ status, float* __restrict__ max_err) {
    extern __shared__ __attribute__((aligned(16))) uint64_t skeys[];
    const uint32_t q = blockIdx.x;
    const bool small = kp <= kRankSortMax;
    const uint32_t np2 = small ? kp : pow2_ceil(kp);
    float err = 0.0f;
    for (uint32_t i = threadIdx.x; i < np2; i += kSortThreads) {
        uint64_t key = 0;
        if (i < kp) {
            const uint32_t row = cand_rows[(uint64_t)q * kp + i];
            if (row != 0xFFFFFFFFu) {
                const float c = cand_canon[(uint64_t)q * kp + i];
                key = make_key(score_key_rt(c, metric), row);
                float e = __builtin_fabsf(cand_fast[(uint64_t)q * kp + i] - c);
                if (eps_mode == 1) e = e / __builtin_fmaxf(__builtin_fabsf(c), 1e-30f);  // relative bound: relative error
                if (e == e) err = __builtin_fmaxf(err, e);
            }
        }
        skeys[i] = key;
    }
    for (int o = 32; o > 0; o >>= 1) err = __builtin_fmaxf(err, __shfl_xor(err, o));
    if ((threadIdx.x & 63) == 0 && err > 0.0f) atomicMax((uint32_t*)max_err, __float_as_uint(err));
    __syncthreads();
    const uint64_t* sorted;
    if (small) {
        rank_sort_desc(skeys, skeys + kRankSortMax, np2);
        sorted = skeys + kRankSortMax;
    } else {
        bitonic_sort_desc(skeys, np2);
        sorted = skeys;
    }
    for (uint32_t i = threadIdx.x; i < k; i += kSortThreads) {
        const uint64_t key = i < np2 ? sorted[i] : 0ull;
        out_ids[(uint64_t)q * k + i] = key ? idmap(key_row(key)) : UINT64_MAX;
        out_scores[(uint64_t)q * k + i] = key ? key_to_score_rt(key_skey(key), metric) : __uint_as_float(kScoreNoneBits);
    }
    if (threadIdx.x == 0) {
        const float t = T[q];
        bool ok;
        // norms so large that the bound itself overflows (squared L2 distances beyond FLT_MAX):
        // fast scores are inf / NaN there and T = +inf no longer means "nothing was left out"
        const float qn_ = __builtin_sqrtf(__uint_as_float(*max_qn2_bits)), xn_ = __builtin_sqrtf(__uint_as_float(*max_xn2_bits));
        const float span = metric == M_COSINE ? qn_ * xn_ : (qn_ + xn_) * (qn_ + xn_);
        if (!(span < 3.0e38f)) {
            ok = false;
        } else if (t == worst_score(metric)) {
            ok = true;  // every row of the shard was a candidate
        } else {
            const uint64_t kk = (k >= 1 && k <= np2) ? sorted[k - 1] : 0ull;
            if (!kk) {
                ok = false;  // fewer than k candidates although rows were left out
            } else {
                const float sk = key_to_score_rt(key_skey(kk), metric);
                const float qn = __builtin_sqrtf(__uint_as_float(*max_qn2_bits));
                const float xn = __builtin_sqrtf(__uint_as_float(*max_xn2_bits));
                // + eps_c * 2^-125 = ~4d roundings of half a denormal ulp (2^-150) each: in the
                // denormal range (scores below 1e-38) a rounding error is absolute, not relative
                const float eps = (eps_mode == 0 ? eps_c * qn * xn
                                 : eps_mode == 1 ? eps_c * __builtin_fabsf(t) + 1e-30f
                                                 : eps_c * (qn + xn) * (qn + xn)) + eps_c * 2.3509887e-38f;
                ok = metric == M_COSINE ? (sk > t + eps) : (sk < t - eps);
            }
        }
        if (!ok) atomicOr(&status[q], 1u);
    }
}

void launch_final_topk(const uint32_t* d_cand_rows, const float* d_cand_fast,
                       const float* d_cand_canon, const float* d_T, int nq, uint32_t kp, uint32_t k,
                       int metric, const IdMap& idmap, int eps_mode, float eps_c,
                       const uint32_t* d_max_qn2_bits, const uint32_t* d_max_xn2_bits,
                       uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_status,
                       float* d_max_err, hipStream_t s) {
    if (!nq) return;
    size_t lds = sort_lds_bytes(kp);
    if (lds < 2 * kRankSortMax * sizeof(uint64_t)) lds = 2 * kRankSortMax * sizeof(uint64_t);
    final_topk_kernel<<<nq, kSortThreads, lds, s>>>(d_cand_rows, d_cand_fast, d_cand_canon, d_T, kp, k, metric, idmap, eps_mode, eps_c, d_max_qn2_bits, d_max_xn2_bits, d_out_ids, d_out_scores, d_status, d_max_err);
}

// ------------------------------------------------------------------ readback block
__global__ __launch_bounds__(256) void gather_readback_kernel(const uint32_t* __restrict__ status, uint32_t nq,
                                                              uint32_t* __restrict__ flags3,
                                                              const uint32_t* __restrict__ max_xn2,
                                                              uint32_t* __restrict__ out) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nq; i += gridDim.x * 256) out[i] = status[i];
    if (blockIdx.x == 0 && threadIdx.x < 4) {
        out[nq + threadIdx.x] = threadIdx.x < 3 ? flags3[threadIdx.x] : *max_xn2;
        // the per-search scalars (bad-value flag, max |q|^2, max error) are consumed here, so that the
        // next search in the stream (which may be enqueued before the host has looked at this one)
        // starts from clean words.  They must NOT be cleared by the launch that accumulates into them
        // (prep_queries_kernel's blocks atomicMax into max |q|^2 with no grid-wide order against a clear).
        if (threadIdx.x < 3) flags3[threadIdx.x] = 0;
    }
}

void launch_gather_readback(const uint32_t* d_status, uint32_t nq, uint32_t* d_flags3, const uint32_t* d_max_xn2,
                            uint32_t* d_out, hipStream_t s) {
    gather_readback_kernel<<<(nq + 255) / 256 > 0 ? (nq + 255) / 256 : 1, 256, 0, s>>>(d_status, nq, d_flags3, d_max_xn2, d_out);
}

// ------------------------------------------------------------------ band pass (second chance)
// A query whose certificate failed still has k candidates with canonical scores; c_k, the k-th of them, is a
// lower bound of the true k-th best canonical score (the best k of a subset cannot beat the best k of
// everything).  Every row of the true top-k -- ties at the boundary included -- therefore has
//     canonical >= c_k   =>   fast >= c_k - eps            (COSINE; mirrored for L2)
// so ONE more filtered scan with the per-query threshold c_k - eps collects a superset of the answer (the
// "band"); canonical re-score of the band + exact select by (score, id) is then exact with no certificate.
// band_prepare_kernel: per failed query f (original index qidx[f]): thr[f] = c_k -/+ 1.01 eps (the scan appends
// rows STRICTLY better than thr: the slack keeps equality in), counts[f] = 0, ok[f] = 1; a query with fewer than
// k candidates (c_k undefined) or norms that overflow the bound gets ok[f] = 0 and a threshold nothing passes.
// Padding queries f >= nf: a threshold nothing passes.
__global__ __launch_bounds__(256) void band_prepare_kernel(const uint32_t* __restrict__ qidx, uint32_t nf, uint32_t nf_pad,
                                                           const float* __restrict__ out_scores, uint32_t k, int metric,
                                                           int eps_mode, float eps_c, const uint32_t* __restrict__ max_qn2_bits,
                                                           const uint32_t* __restrict__ max_xn2_bits, const float* __restrict__ qn2_all,
                                                           float* __restrict__ thr, float* __restrict__ qn2, uint32_t* __restrict__ counts,
                                                           uint32_t* __restrict__ ok) {
    const uint32_t f = blockIdx.x * 256 + threadIdx.x;
    if (f >= nf_pad) return;
    const float never = metric == M_COSINE ? __builtin_huge_valf() : -__builtin_huge_valf();   // nothing is strictly better
    counts[f] = 0u;
    if (f >= nf) { thr[f] = never; qn2[f] = 0.0f; ok[f] = 0u; return; }
    const uint32_t q = qidx[f];
    qn2[f] = qn2_all[q];
    const float ck = out_scores[(uint64_t)q * k + (k - 1)];
    const float qn = __builtin_sqrtf(__uint_as_float(*max_qn2_bits)), xn = __builtin_sqrtf(__uint_as_float(*max_xn2_bits));
    const float span = metric == M_COSINE ? qn * xn : (qn + xn) * (qn + xn);
    if (!(ck == ck) || !(span < 3.0e38f) || eps_mode == 1) { thr[f] = never; ok[f] = 0u; return; }
    const float eps = (eps_mode == 0 ? eps_c * qn * xn : eps_c * (qn + xn) * (qn + xn)) + eps_c * 2.3509887e-38f;
    const float slack = 1.01f * eps + 1e-37f;
    thr[f] = metric == M_COSINE ? ck - slack : ck + slack;
    ok[f] = 1u;
}

void launch_band_prepare(const uint32_t* d_qidx, uint32_t nf, uint32_t nf_pad, const float* d_out_scores, uint32_t k, int metric,
                         int eps_mode, float eps_c, const uint32_t* d_max_qn2_bits, const uint32_t* d_max_xn2_bits,
                         const float* d_qn2_all, float* d_thr, float* d_qn2, uint32_t* d_counts, uint32_t* d_ok, hipStream_t s) {
    if (!nf_pad) return;
    band_prepare_kernel<<<(nf_pad + 255) / 256, 256, 0, s>>>(d_qidx, nf, nf_pad, d_out_scores, k, metric, eps_mode, eps_c, d_max_qn2_bits,
                                                            d_max_xn2_bits, d_qn2_all, d_thr, d_qn2, d_counts, d_ok);
}

// prepared query rows of the failed queries, gathered into a dense block [nf_pad][ld] (zero rows for padding); the bf16
// copy is exact (the prepared rows of a BF16 handle hold bf16 values already)
__global__ __launch_bounds__(256) void gather_query_rows_kernel(const float* __restrict__ src, const uint32_t* __restrict__ qidx, uint32_t nf,
                                                                uint32_t ld, float* __restrict__ dst_f32, bf16_t* __restrict__ dst_bf16) {
    const uint32_t f = blockIdx.x;
    const bool live = f < nf;
    const float* row = live ? src + (uint64_t)qidx[f] * ld : nullptr;
    for (uint32_t j = threadIdx.x; j < ld; j += 256) {
        const float v = live ? row[j] : 0.0f;
        dst_f32[(uint64_t)f * ld + j] = v;
        if (dst_bf16) dst_bf16[(uint64_t)f * ld + j] = f32_to_bf16_rne(v);
    }
}

void launch_gather_query_rows(const float* d_src, const uint32_t* d_qidx, uint32_t nf, uint32_t nf_pad, uint32_t ld, float* d_dst_f32,
                              void* d_dst_bf16, hipStream_t s) {
    if (!nf_pad) return;
    gather_query_rows_kernel<<<nf_pad, 256, 0, s>>>(d_src, d_qidx, nf, ld, d_dst_f32, (bf16_t*)d_dst_bf16);
}

// rows [f][k] of the band pass's results -> rows qidx[f] of the caller's outputs, for the resolved queries only
__global__ __launch_bounds__(256) void scatter_results_kernel(const uint64_t* __restrict__ ids, const float* __restrict__ scores,
                                                              const uint32_t* __restrict__ qidx, const uint32_t* __restrict__ resolved,
                                                              uint32_t k, uint64_t* __restrict__ out_ids, float* __restrict__ out_scores) {
    const uint32_t f = blockIdx.x;
    if (!resolved[f]) return;
    const uint32_t q = qidx[f];
    for (uint32_t i = threadIdx.x; i < k; i += 256) {
        out_ids[(uint64_t)q * k + i] = ids[(uint64_t)f * k + i];
        out_scores[(uint64_t)q * k + i] = scores[(uint64_t)f * k + i];
    }
}

void launch_scatter_results(const uint64_t* d_ids, const float* d_scores, const uint32_t* d_qidx, const uint32_t* d_resolved, uint32_t nf,
                            uint32_t k, uint64_t* d_out_ids, float* d_out_scores, hipStream_t s) {
    if (!nf) return;
    scatter_results_kernel<<<nf, 256, 0, s>>>(d_ids, d_scores, d_qidx, d_resolved, k, d_out_ids, d_out_scores);
}

// ------------------------------------------------------------------ exact path output
// keys: n <= kSelectChunk composite keys built from CANONICAL scores of one query.
__global__ __launch_bounds__(kSortThreads) void keys_to_output_kernel(
    const uint64_t* __restrict__ keys, uint32_t n, int metric, uint32_t k, IdMap idmap,
    uint64_t* __restrict__ out_ids, float* __restrict__ out_scores) {
    extern __shared__ __attribute__((aligned(16))) uint64_t skeys[];
    const uint32_t np2 = pow2_ceil(n < 2 ? 2 : n);
    for (uint32_t i = threadIdx.x; i < np2; i += kSortThreads) skeys[i] = i < n ? keys[i] : 0ull;
    __syncthreads();
    bitonic_sort_desc(skeys, np2);
    for (uint32_t i = threadIdx.x; i < k; i += kSortThreads) {
        const uint64_t key = i < np2 ? skeys[i] : 0ull;
        out_ids[i] = key ? idmap(key_row(key)) : UINT64_MAX;
        out_scores[i] = key ? key_to_score_rt(key_skey(key), metric) : __uint_as_float(kScoreNoneBits);
    }
}

void launch_keys_to_output(const uint64_t* d_keys, uint64_t n, int metric, uint32_t k,
                           const IdMap& idmap, uint64_t* d_out_ids, float* d_out_scores,
                           hipStream_t s) {
    keys_to_output_kernel<<<1, kSortThreads, sort_lds_bytes(n), s>>>(d_keys, (uint32_t)n, metric, k, idmap, d_out_ids, d_out_scores);
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
                                                         uint64_t list_stride_ids, uint64_t list_stride_scores,
                                                         uint32_t n_lists, uint32_t k,
                                                         uint64_t* __restrict__ out_ids,
                                                         float* __restrict__ out_scores) {
    const uint32_t q = blockIdx.x;  // gridDim.x == nq
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
        out_ids[(uint64_t)q * k + i] = UINT64_MAX;
        out_scores[(uint64_t)q * k + i] = __uint_as_float(kScoreNoneBits);
    }
    __syncthreads();
    const uint32_t total = n_lists * k;
    for (uint32_t e = threadIdx.x; e < total; e += blockDim.x) {
        const uint32_t li = e / k, pi = e - li * k;
        const uint64_t base_i = (uint64_t)li * list_stride_ids + (uint64_t)q * k;
        const uint64_t base_s = (uint64_t)li * list_stride_scores + (uint64_t)q * k;
        const uint64_t id = ids[base_i + pi];
        if (id == UINT64_MAX) continue;
        const float sc = scores[base_s + pi];
        const uint32_t key = score_key_rt(sc, metric);
        uint32_t rank = pi;
        for (uint32_t m = 0; m < n_lists; ++m) {
            if (m == li) continue;
            const uint64_t mb_i = (uint64_t)m * list_stride_ids + (uint64_t)q * k;
            const uint64_t mb_s = (uint64_t)m * list_stride_scores + (uint64_t)q * k;
            uint32_t lo = 0, hi = k;  // first index whose element does NOT rank before (key, id)
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                const uint64_t mid_id = ids[mb_i + mid];
                const bool before = mid_id != UINT64_MAX &&
                                    ranks_before(score_key_rt(scores[mb_s + mid], metric), mid_id, key, id);
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

__global__ __launch_bounds__(256) void fill_none_kernel(uint64_t* __restrict__ ids, float* __restrict__ scores, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        ids[i] = UINT64_MAX;
        scores[i] = __uint_as_float(kScoreNoneBits);
    }
}

void launch_fill_none(uint64_t* d_ids, float* d_scores, uint64_t n, hipStream_t s) {
    if (!n) return;
    fill_none_kernel<<<(unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), 256, 0, s>>>(d_ids, d_scores, n);
}

void launch_merge_topk(int metric, const uint64_t* d_ids, const float* d_scores, uint64_t list_stride_ids,
                       uint64_t list_stride_scores, uint32_t n_lists, uint32_t nq, uint32_t k,
                       uint64_t* d_out_ids, float* d_out_scores, hipStream_t s) {
    if (!nq || !k) return;
    merge_topk_kernel<<<nq, 256, 0, s>>>(metric, d_ids, d_scores, list_stride_ids, list_stride_scores, n_lists, k, d_out_ids, d_out_scores);
}

}  // namespace vrod
