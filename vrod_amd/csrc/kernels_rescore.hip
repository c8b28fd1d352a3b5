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
// 16-B column walk out: row stride 68 dwords).
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

constexpr int kTileStride = 68;  // dwords: 16-B aligned rows, conflict-free b128 column walk

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// One wave per block.  ALL == false: the block owns candidate slots [64*blockIdx.x, +64) of
// query blockIdx.y (rows from d_cand_rows, ~0u = empty; empties sit at the end of a list).
// ALL == true: rows [64*blockIdx.x, +64) of the shard for the single query d_q.
// Each 64-element chunk of the 64 rows is fetched with 16-B loads, all issued before the
// first LDS write (the chain itself is sequential, the loads must not be), and then every
// lane walks its own row of the tile in order.
template <typename T, int METRIC, bool ALL>
__global__ __launch_bounds__(64) void rescore_kernel(const T* __restrict__ corpus, uint32_t dim,
                                                     uint32_t ld, const float* __restrict__ q,
                                                     const uint32_t* __restrict__ cand_rows,
                                                     uint32_t kp, uint64_t nrows,
                                                     float* __restrict__ out) {
    constexpr int EPU = 16 / (int)sizeof(T);   // elements per 16-B load: 4 fp32 / 8 bf16
    constexpr int LPC = 64 / EPU;              // lanes covering one row's 64-element chunk
    constexpr int RPI = 64 / LPC;              // rows per load instruction
    constexpr int NI = 64 / RPI;               // load instructions per chunk
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* q_lds = smem;                       // [ld]
    float* tile = smem + ld;                   // [64][68]
    const int lane = threadIdx.x;

    const uint32_t qi = ALL ? 0u : blockIdx.y;
    const float* qrow = q + (uint64_t)qi * ld;
    for (uint32_t i = lane; i < ld; i += 64) q_lds[i] = qrow[i];

    const uint64_t slot = (uint64_t)blockIdx.x * 64 + lane;
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
    const unsigned long long vmask = __ballot(valid);
    if (vmask == 0ull) return;  // whole wave idle (single-wave block: no barrier is skipped)
    // valid slots are a prefix of the wave; rows beyond it are not fetched
    const int nvalid = 64 - __builtin_clzll(vmask);
    const int ni_used = (nvalid + RPI - 1) / RPI;
    const int sub = lane / LPC, part = lane % LPC;

    float acc = 0.0f;
    // The chain is sequential and latency-bound; the row gathers must not be: the loads of chunks
    // c+1 .. c+D-1 are in flight while the chain of chunk c runs (ring of D register sets; with
    // one chunk ahead a 768-d re-score waited a full gather latency twelve times: 38-44 us).
    constexpr int D = 3;
    u32x4 v[D][NI];
    auto fetch = [&](u32x4 (&vs)[NI], uint32_t j0) {
        const uint32_t e0 = j0 + part * EPU;  // first element this lane fetches
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            vs[it] = u32x4{0u, 0u, 0u, 0u};
            if (it < ni_used) {
                const uint32_t row = __shfl(my_row, it * RPI + sub);
                if (e0 < ld) vs[it] = *reinterpret_cast<const u32x4*>(corpus + (uint64_t)row * ld + e0);
            }
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d)
        if ((uint32_t)d * 64 < dim) fetch(v[d], d * 64);
    for (uint32_t jb = 0; jb < dim; jb += 64 * D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const uint32_t j0 = jb + d * 64;
            if (j0 >= dim) break;
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                float* t = tile + (it * RPI + sub) * kTileStride + part * EPU;
                if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<u32x4*>(t) = v[d][it];
                } else {
                    *reinterpret_cast<u32x4*>(t) = u32x4{v[d][it].x << 16, v[d][it].x & 0xFFFF0000u, v[d][it].y << 16, v[d][it].y & 0xFFFF0000u};
                    *reinterpret_cast<u32x4*>(t + 4) = u32x4{v[d][it].z << 16, v[d][it].z & 0xFFFF0000u, v[d][it].w << 16, v[d][it].w & 0xFFFF0000u};
                }
            }
            if (j0 + 64 * D < dim) fetch(v[d], j0 + 64 * D);
            // the tile (and q_lds on the first pass) is wave-private: in-order LDS + a compiler fence
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t jn = dim - j0 < 64 ? dim - j0 : 64;
            const float* trow = tile + lane * kTileStride;
            if (jn == 64) {
                // whole chunk: fully unrolled, so the LDS reads and the products (independent)
                // run ahead of the one thing that is serial, the 64 dependent adds
#pragma unroll
                for (uint32_t l = 0; l < 64; l += 4) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(trow + l);
                    const f32x4 qq = *reinterpret_cast<const f32x4*>(q_lds + j0 + l);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (METRIC == M_COSINE) {
                            acc = add_rn(acc, mul_rn(qq[e], x[e]));
                        } else {
                            const float dd = sub_rn(qq[e], x[e]);
                            acc = add_rn(acc, mul_rn(dd, dd));
                        }
                    }
                }
            } else {
                for (uint32_t l = 0; l < jn; l += 4) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(trow + l);
                    const f32x4 qq = *reinterpret_cast<const f32x4*>(q_lds + j0 + l);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (l + e < jn) {
                            if constexpr (METRIC == M_COSINE) {
                                acc = add_rn(acc, mul_rn(qq[e], x[e]));
                            } else {
                                const float dd = sub_rn(qq[e], x[e]);
                                acc = add_rn(acc, mul_rn(dd, dd));
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if constexpr (ALL) {
        if (valid) out[slot] = acc;
    } else {
        if (slot < kp) out[(uint64_t)qi * kp + slot] = valid ? acc : __uint_as_float(kScoreNoneBits);
    }
}

template <bool ALL>
static void launch_rescore(const void* d_corpus, int dtype, int metric, uint32_t dim, uint32_t ld,
                           const float* d_q, int nq, const uint32_t* d_cand_rows, uint32_t kp,
                           uint64_t nrows, float* d_out, hipStream_t s) {
    const size_t lds = ((size_t)ld + 64 * kTileStride) * sizeof(float);
    dim3 grid(ALL ? (unsigned)((nrows + 63) / 64) : (kp + 63) / 64, ALL ? 1 : nq);
#define VROD_RS(TT, MM) rescore_kernel<TT, MM, ALL><<<grid, 64, lds, s>>>((const TT*)d_corpus, dim, ld, d_q, d_cand_rows, kp, nrows, d_out)
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
