// kernels_rescore.hip -- canonical (oracle-order) re-score on gfx950.
//
// The fast passes (kernels_stream.hip, kernels_mfma.hip) accumulate in whatever order is
// fastest.  vRod's data model is a plain Vec<Vec<f32>> walked by one thread
// (reference src/utils/embeddings.rs:29, src/command/types.rs:10), so the result contract
// (DESIGN.md "Scan spec", oracle/vrod_oracle.c orc_dot_canonical / orc_l2_canonical) is the
// strictly sequential fp32 chain   acc = acc + (q[j] * x[j])   with the product and the
// sum each rounded to fp32 -- no FMA, no reassociation.  This kernel recomputes exactly
// that chain for the few candidates the fast pass selected, so ids AND score bits equal
// the oracle's.  v_mul_f32 / v_add_f32 / v_sub_f32 are issued through inline asm so that
// no compiler contraction (hipcc defaults to -ffp-contract=fast) can fuse them.
//
// Layout: one lane owns one (query, candidate row) chain.  A wave stages 64 candidate
// rows x 64 elements through an LDS tile (coalesced 256-B / 128-B row segments in,
// conflict-free column walk out: row stride 65 dwords).
#include "vrod_common.h"
#include "vrod_kernels.h"

namespace vrod {

__device__ __forceinline__ float mul_rn(float a, float b) {
    float r;
    asm("v_mul_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float add_rn(float a, float b) {
    float r;
    asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
    float r;
    asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

constexpr int kTileStride = 65;

// ALL == false: candidates d_cand_rows[q][kp] (~0u = empty).  grid = (ceil(kp/256), nq).
// ALL == true : rows blockIdx.x*256 + ... of the shard for the single query d_q. grid = (ceil(n/256), 1).
template <typename T, int METRIC, bool ALL>
__global__ __launch_bounds__(256) void rescore_kernel(const T* __restrict__ corpus, uint32_t dim,
                                                      uint32_t ld, const float* __restrict__ q,
                                                      const uint32_t* __restrict__ cand_rows,
                                                      uint32_t kp, uint64_t nrows,
                                                      float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* q_lds = smem;                                   // [ld]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* tile = smem + ld + wave * (64 * kTileStride);   // [64][65]

    const uint32_t qi = ALL ? 0u : blockIdx.y;
    const float* qrow = q + (uint64_t)qi * ld;
    for (uint32_t i = threadIdx.x; i < ld; i += blockDim.x) q_lds[i] = qrow[i];
    __syncthreads();

    const uint64_t slot = (uint64_t)blockIdx.x * 256 + threadIdx.x;  // candidate slot / row
    uint32_t my_row;
    bool valid;
    if constexpr (ALL) {
        valid = slot < nrows;
        my_row = valid ? (uint32_t)slot : 0u;
    } else {
        my_row = slot < kp ? cand_rows[(uint64_t)qi * kp + slot] : 0xFFFFFFFFu;
        valid = my_row != 0xFFFFFFFFu;
        if (!valid) my_row = 0u;
    }
    // whole wave idle? (uniform) -- still must not skip the barrier above, which is done
    if (__ballot(valid) != 0ull) {
        float acc = 0.0f;
        for (uint32_t j0 = 0; j0 < dim; j0 += 64) {
            const uint32_t j = j0 + lane;
#pragma unroll 8
            for (int c = 0; c < 64; ++c) {
                const uint32_t row_c = __shfl(my_row, c);
                float v = 0.0f;
                if (j < dim) {
                    if constexpr (sizeof(T) == 2) v = bf16_to_f32(corpus[(uint64_t)row_c * ld + j]);
                    else v = corpus[(uint64_t)row_c * ld + j];
                }
                tile[c * kTileStride + lane] = v;
            }
            // tile is private to the wave: wave-level ordering of LDS ops suffices
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t jn = dim - j0 < 64 ? dim - j0 : 64;
            for (uint32_t l = 0; l < jn; ++l) {
                const float x = tile[lane * kTileStride + l];
                const float qq = q_lds[j0 + l];
                if constexpr (METRIC == M_COSINE) {
                    acc = add_rn(acc, mul_rn(qq, x));
                } else {
                    const float d = sub_rn(qq, x);
                    acc = add_rn(acc, mul_rn(d, d));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (ALL) {
            if (valid) out[slot] = acc;
        } else {
            if (slot < kp) out[(uint64_t)qi * kp + slot] = valid ? acc : __uint_as_float(kScoreNoneBits);
        }
    }
}

template <bool ALL>
static void launch_rescore(const void* d_corpus, int dtype, int metric, uint32_t dim, uint32_t ld,
                           const float* d_q, int nq, const uint32_t* d_cand_rows, uint32_t kp,
                           uint64_t nrows, float* d_out, hipStream_t s) {
    const size_t lds = ((size_t)ld + 4 * 64 * kTileStride) * sizeof(float);
    dim3 grid(ALL ? (unsigned)((nrows + 255) / 256) : (kp + 255) / 256, ALL ? 1 : nq);
#define VROD_RS(TT, MM) rescore_kernel<TT, MM, ALL><<<grid, 256, lds, s>>>((const TT*)d_corpus, dim, ld, d_q, d_cand_rows, kp, nrows, d_out)
    if (dtype == DT_BF16) { if (metric == M_COSINE) VROD_RS(bf16_t, M_COSINE); else VROD_RS(bf16_t, M_L2); }
    else { if (metric == M_COSINE) VROD_RS(float, M_COSINE); else VROD_RS(float, M_L2); }
#undef VROD_RS
}

void launch_rescore_candidates(const void* d_corpus, int dtype, int metric, uint32_t dim,
                               uint32_t ld, const float* d_q, int nq, const uint32_t* d_cand_rows,
                               uint32_t kp, float* d_out, hipStream_t s) {
    if (!nq || !kp) return;
    launch_rescore<false>(d_corpus, dtype, metric, dim, ld, d_q, nq, d_cand_rows, kp, 0, d_out, s);
}

void launch_rescore_all(const void* d_corpus, int dtype, int metric, uint32_t dim, uint32_t ld,
                        const float* d_q1, uint64_t nrows, float* d_out, hipStream_t s) {
    if (!nrows) return;
    launch_rescore<true>(d_corpus, dtype, metric, dim, ld, d_q1, 1, nullptr, 0, nrows, d_out, s);
}

}  // namespace vrod
