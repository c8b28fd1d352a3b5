// kernels_stream.hip -- the HBM-bound small-batch scan (1..8 queries) for gfx950.
//
// Fills the scan slot under SearchSimilarCommand::execute (reference
// src/command/types.rs:121-132, empty) for small batches: every corpus byte is read once
// with 16-B-per-lane coalesced loads straight to VGPRs (no LDS round trip: the stream is
// used once per wave -- cdna_hip_programming.md "GEMV / M <= 16" row), the queries sit in
// LDS, and each wave reduces its rows with cross-lane shuffles.  Output: the fp32 FAST
// score of every row (N*4 bytes per query, <0.2% of the corpus bytes); selection and the
// canonical re-score happen downstream (kernels_select.hip, kernels_rescore.hip).
//
// Roofline: HBM.  Algorithmic bytes per launch = nrows * ld * sizeof(T)  (SURVEY.md 8d).
//
// Row mapping: a row of `ld` elements is `upr` 16-B units.  LPR = the largest power of
// two <= 64 dividing upr lanes share a row (IT = upr/LPR loads per lane per row), so one
// wave-load covers R = 64/LPR rows with whole 128-B lines.  A wave owns 64 consecutive
// rows per outer step and emits their 64 scores with one coalesced store.
#include "vrod_common.h"
#include "vrod_kernels.h"

namespace vrod {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Unit;
template <> struct Unit<float> { typedef f32x4 vec; static constexpr int EPU = 4; };
template <> struct Unit<bf16_t> { typedef u32x4 vec; static constexpr int EPU = 8; };

template <int METRIC>
__device__ __forceinline__ float acc1(float acc, float q, float x) {
    if constexpr (METRIC == M_COSINE) return __builtin_fmaf(q, x, acc);
    else { const float d = q - x; return __builtin_fmaf(d, d, acc); }
}

template <typename T, int METRIC>
__device__ __forceinline__ float unit_accumulate(float acc, const typename Unit<T>::vec v,
                                                 const float* __restrict__ q) {
    if constexpr (sizeof(T) == 4) {
        const f32x4 qv = *reinterpret_cast<const f32x4*>(q);
        acc = acc1<METRIC>(acc, qv.x, v.x);
        acc = acc1<METRIC>(acc, qv.y, v.y);
        acc = acc1<METRIC>(acc, qv.z, v.z);
        acc = acc1<METRIC>(acc, qv.w, v.w);
    } else {
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(q);
        const f32x4 q1 = *reinterpret_cast<const f32x4*>(q + 4);
        acc = acc1<METRIC>(acc, q0.x, __uint_as_float(v.x << 16));
        acc = acc1<METRIC>(acc, q0.y, __uint_as_float(v.x & 0xFFFF0000u));
        acc = acc1<METRIC>(acc, q0.z, __uint_as_float(v.y << 16));
        acc = acc1<METRIC>(acc, q0.w, __uint_as_float(v.y & 0xFFFF0000u));
        acc = acc1<METRIC>(acc, q1.x, __uint_as_float(v.z << 16));
        acc = acc1<METRIC>(acc, q1.y, __uint_as_float(v.z & 0xFFFF0000u));
        acc = acc1<METRIC>(acc, q1.z, __uint_as_float(v.w << 16));
        acc = acc1<METRIC>(acc, q1.w, __uint_as_float(v.w & 0xFFFF0000u));
    }
    return acc;
}

// Histogram resolution of the fused first radix pass: the top HB bits of the order-preserving
// score key (sign, exponent, leading mantissa bits).  Smaller for more queries (LDS).
template <int NQ> struct StreamHist { static constexpr int HB = NQ <= 2 ? 12 : (NQ == 4 ? 11 : 10); };
int stream_hist_bits(int nq_pad) { return nq_pad <= 2 ? 12 : (nq_pad == 4 ? 11 : 10); }
// Queries a scan pass can hold in LDS next to its histograms (160 KB per work-group): 8 up to
// d = 4096, fewer for longer rows, 0 when not even one fits (d > ~36k floats).
int stream_max_queries_per_pass(uint32_t ld) {
    for (int nq = 8; nq >= 1; nq >>= 1) {
        const size_t lds = (size_t)nq * ld * sizeof(float) + (size_t)nq * (sizeof(uint32_t) << stream_hist_bits(nq)) + 32;
        if (lds <= 160 * 1024) return nq;
    }
    return 0;
}

// IT > 0: loads per lane per row known at compile time, UNROLL row-steps in flight.
// IT == 0: runtime `it` (any ld), one row-step at a time.
template <typename T, int METRIC, int NQ, int IT, int UNROLL>
__global__ __launch_bounds__(256) void scan_stream_kernel(const T* __restrict__ corpus,
                                                          uint32_t ld, uint64_t nrows,
                                                          const float* __restrict__ q,
                                                          float* __restrict__ scores,
                                                          uint64_t score_ld, int lpr_log2,
                                                          int it_rt, uint32_t* __restrict__ ghist,
                                                          uint32_t kp) {
    typedef typename Unit<T>::vec vec_t;
    constexpr int EPU = Unit<T>::EPU;
    constexpr int HB = StreamHist<NQ>::HB;          // histogram bits of the score key
    constexpr int NBINS = 1 << HB;
    extern __shared__ __attribute__((aligned(16))) float q_lds[];  // [NQ][ld] then hist [NQ][NBINS]
    uint32_t* hist = reinterpret_cast<uint32_t*>(q_lds + (size_t)NQ * ld);

    for (uint32_t i = threadIdx.x; i < (uint32_t)NQ * ld; i += blockDim.x) q_lds[i] = q[i];
    for (uint32_t i = threadIdx.x; i < (uint32_t)NQ * NBINS; i += blockDim.x) hist[i] = 0u;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int LPR = 1 << lpr_log2;
    const int R = 64 >> lpr_log2;          // rows per wave-load
    const int pos0 = lane & (LPR - 1);     // 16-B unit within the row (first iteration)
    const int grp = lane >> lpr_log2;      // which of the R rows
    const int it = IT > 0 ? IT : it_rt;

    const uint64_t nblk = (nrows + 63) / 64;  // 64-row wave blocks
    const uint64_t wave0 = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x >> 6);

    for (uint64_t wb = wave0; wb < nblk; wb += nwaves) {
        const uint64_t base = wb * 64;
        float keep[NQ];
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) keep[qi] = 0.0f;

        if constexpr (IT > 0) {
            for (int s0 = 0; s0 < LPR; s0 += UNROLL) {
                vec_t v[UNROLL][IT];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    // capacity is padded to 256 rows, so rows past nrows are readable zeros
                    const uint64_t row = base + (uint64_t)(s0 + u) * R + grp;
                    const vec_t* rp = reinterpret_cast<const vec_t*>(corpus + row * ld) + pos0;
#pragma unroll
                    for (int i = 0; i < IT; ++i) v[u][i] = __builtin_nontemporal_load(rp + i * LPR);
                }
                float acc[UNROLL][NQ];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi) acc[u][qi] = 0.0f;
#pragma unroll
                    for (int i = 0; i < IT; ++i) {
                        const float* qp = q_lds + (pos0 + i * LPR) * EPU;
#pragma unroll
                        for (int qi = 0; qi < NQ; ++qi)
                            acc[u][qi] = unit_accumulate<T, METRIC>(acc[u][qi], v[u][i], qp + qi * ld);
                    }
                }
                // cross-lane sums: all UNROLL*NQ chains advance together through each shuffle step
                // (one chain at a time is UNROLL*NQ*log2(LPR) dependent LDS round trips)
                for (int o = LPR >> 1; o > 0; o >>= 1) {
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                        for (int qi = 0; qi < NQ; ++qi) acc[u][qi] += __shfl_xor(acc[u][qi], o);
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi)
                        if (pos0 == s0 + u) keep[qi] = acc[u][qi];
            }
        } else {
            for (int s = 0; s < LPR; ++s) {
                const uint64_t row = base + (uint64_t)s * R + grp;
                const vec_t* rp = reinterpret_cast<const vec_t*>(corpus + row * ld) + pos0;
                float acc[NQ];
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) acc[qi] = 0.0f;
#pragma unroll 4
                for (int i = 0; i < it; ++i) {
                    const vec_t v = __builtin_nontemporal_load(rp + i * LPR);
                    const float* qp = q_lds + (pos0 + i * LPR) * EPU;
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi)
                        acc[qi] = unit_accumulate<T, METRIC>(acc[qi], v, qp + qi * ld);
                }
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
                    float a = acc[qi];
                    for (int o = LPR >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o);
                    if (pos0 == s) keep[qi] = a;
                }
            }
        }
        // lane (grp, pos0) holds row base + pos0*R + grp: one 256-B segment per query
        const uint64_t orow = base + (uint64_t)pos0 * R + grp;
        if (orow < nrows) {
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                scores[(uint64_t)qi * score_ld + orow] = keep[qi];
                atomicAdd(&hist[qi * NBINS + (score_key<METRIC>(keep[qi]) >> (32 - HB))], 1u);
            }
        }
    }

    // ---- publish this block's histogram, top bins only.  The global kp-th best score is at
    // least as good as this block's own kp-th best, so bins below the block's kp-th bin can
    // never hold the global threshold: only bins >= that bin are added to the global
    // histogram (a few bins instead of every non-empty one -> few contended atomics).
    __syncthreads();
    uint32_t* ctl = hist + NQ * NBINS;  // 8 words
    constexpr int BPT = NBINS / 256;
#pragma unroll 1
    for (int qi = 0; qi < NQ; ++qi) {
        const uint32_t* h = hist + qi * NBINS;
        block_find_cut_bin(h, NBINS, kp, ctl);
        const int cb = (int)ctl[4];
        const int top = NBINS - 1 - (int)threadIdx.x * BPT;
#pragma unroll
        for (int b = 0; b < BPT; ++b) {
            const int bin = top - b;
            const uint32_t c = h[bin];
            if (bin >= cb && c) atomicAdd(&ghist[qi * NBINS + bin], c);
        }
        __syncthreads();
    }
}

template <typename T, int METRIC, int NQ>
static void dispatch_it(const T* corpus, uint32_t ld, uint64_t nrows, const float* q, float* scores,
                        uint64_t score_ld, int num_blocks, uint32_t* ghist, uint32_t kp, hipStream_t s) {
    constexpr int EPU = Unit<T>::EPU;
    const uint32_t upr = ld / EPU;  // ld is a multiple of 32 (f32) / 64 (bf16): upr % 8 == 0
    int lpr_log2 = 6;
    while ((upr & ((1u << lpr_log2) - 1)) != 0) --lpr_log2;
    const int it = (int)(upr >> lpr_log2);
    const size_t lds = (size_t)NQ * ld * sizeof(float) + (size_t)NQ * (sizeof(uint32_t) << StreamHist<NQ>::HB) + 32;
    const LaunchEvents lev = g_launch_events;
    g_launch_events = LaunchEvents{};
#define VROD_LAUNCH(ITV, UNR)                                                                   \
    hipExtLaunchKernelGGL((scan_stream_kernel<T, METRIC, NQ, ITV, UNR>), dim3(num_blocks), dim3(256), lds, s, \
                          lev.start, lev.stop, 0, corpus, ld, nrows, q, scores, score_ld, lpr_log2, it, ghist, kp)
    switch (it) {
        case 1: VROD_LAUNCH(1, 8); break;
        case 2: VROD_LAUNCH(2, 4); break;
        case 3: VROD_LAUNCH(3, 4); break;
        case 4: VROD_LAUNCH(4, 2); break;
        case 6: VROD_LAUNCH(6, 2); break;
        default: VROD_LAUNCH(0, 1); break;
    }
#undef VROD_LAUNCH
}

template <typename T, int METRIC>
static void dispatch_nq(const T* corpus, uint32_t ld, uint64_t nrows, const float* q, int nq_pad,
                        float* scores, uint64_t score_ld, int num_blocks, uint32_t* ghist, uint32_t kp,
                        hipStream_t s) {
    switch (nq_pad) {
        case 1: dispatch_it<T, METRIC, 1>(corpus, ld, nrows, q, scores, score_ld, num_blocks, ghist, kp, s); break;
        case 2: dispatch_it<T, METRIC, 2>(corpus, ld, nrows, q, scores, score_ld, num_blocks, ghist, kp, s); break;
        case 4: dispatch_it<T, METRIC, 4>(corpus, ld, nrows, q, scores, score_ld, num_blocks, ghist, kp, s); break;
        default: dispatch_it<T, METRIC, 8>(corpus, ld, nrows, q, scores, score_ld, num_blocks, ghist, kp, s); break;
    }
}

void launch_scan_stream(const void* d_corpus, int dtype, int metric, uint32_t ld, uint64_t nrows,
                        const float* d_q, int nq_pad, float* d_scores, uint64_t score_ld,
                        uint32_t* d_hist, uint32_t kp, hipStream_t s) {
    if (!nrows) { g_launch_events = LaunchEvents{}; return; }
    // 64 rows per wave step, 4 waves per block.  2 blocks per CU (8 waves x 12 KB of loads in
    // flight) measured best on MI355X: 6.46 TB/s at 1M x 768 fp32 vs 5.9 TB/s with 8 blocks per
    // CU, whose per-block histogram set-up/flush then costs 10 % (profiles/r01).
    uint64_t blocks = (nrows + 255) / 256;
    if (blocks > 256 * 2) blocks = 256 * 2;
    const int nb = (int)blocks;
    if (dtype == DT_BF16) {
        if (metric == M_COSINE) dispatch_nq<bf16_t, M_COSINE>((const bf16_t*)d_corpus, ld, nrows, d_q, nq_pad, d_scores, score_ld, nb, d_hist, kp, s);
        else dispatch_nq<bf16_t, M_L2>((const bf16_t*)d_corpus, ld, nrows, d_q, nq_pad, d_scores, score_ld, nb, d_hist, kp, s);
    } else {
        if (metric == M_COSINE) dispatch_nq<float, M_COSINE>((const float*)d_corpus, ld, nrows, d_q, nq_pad, d_scores, score_ld, nb, d_hist, kp, s);
        else dispatch_nq<float, M_L2>((const float*)d_corpus, ld, nrows, d_q, nq_pad, d_scores, score_ld, nb, d_hist, kp, s);
    }
}

}  // namespace vrod
