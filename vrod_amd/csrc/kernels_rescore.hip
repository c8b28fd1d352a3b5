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

// One wave per block: the block owns candidate slots [64*blockIdx.x, +64) of query blockIdx.y
// (rows from d_cand_rows, ~0u = empty; empties sit at the end of a list).
// Each 64-element chunk of the 64 rows is fetched with 16-B loads, all issued before the
// first LDS write (the chain itself is sequential, the loads must not be), and then every
// lane walks its own row of the tile in order.
template <typename T, int METRIC>
__global__ __launch_bounds__(64) void rescore_kernel(const T* __restrict__ corpus, uint32_t dim,
                                                     uint32_t ld, const float* __restrict__ q,
                                                     const uint32_t* __restrict__ cand_rows,
                                                     uint32_t kp, float* __restrict__ out) {
    constexpr int EPU = 16 / (int)sizeof(T);   // elements per 16-B load: 4 fp32 / 8 bf16
    constexpr int LPC = 64 / EPU;              // lanes covering one row's 64-element chunk
    constexpr int RPI = 64 / LPC;              // rows per load instruction
    constexpr int NI = 64 / RPI;               // load instructions per chunk
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* q_lds = smem;                       // [ld]
    float* tile = smem + ld;                   // [64][68]
    const int lane = threadIdx.x;

    const uint32_t qi = blockIdx.y;
    const float* qrow = q + (uint64_t)qi * ld;
    for (uint32_t i = lane; i < ld; i += 64) q_lds[i] = qrow[i];

    const uint64_t slot = (uint64_t)blockIdx.x * 64 + lane;
    uint32_t my_row = slot < kp ? cand_rows[(uint64_t)qi * kp + slot] : 0xFFFFFFFFu;
    const bool valid = my_row != 0xFFFFFFFFu;
    if (!valid) my_row = 0u;
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
    if (slot < kp) out[(uint64_t)qi * kp + slot] = valid ? acc : __uint_as_float(kScoreNoneBits);
}

// ---------------------------------------------------------------------------------------
// Exact path: the canonical score of EVERY row for NQ queries at once (the queries whose
// certificate failed).  One pass over the shard serves NQ chains per lane: the corpus chunk is
// staged once, and the NQ dependent add chains are independent of each other, so they fill the
// latency a single chain leaves idle (one query: ~64 serial adds per chunk per lane).
// A query element is the same for all 64 rows of the wave: lane l of a VGPR holds q[j0 + l], and
// v_readlane moves it to an SGPR operand of the product -- NQ query rows in LDS would cost a
// 16-B broadcast read per 4 elements per query (LDS-bound at 8 queries) and 3 waves per CU.
// out[n * out_ld + row], n = position of the query in `qs`.
struct RescoreQuerySet { uint32_t qi[8]; };
template <int I> struct RingSlot { static constexpr int value = I; };

__device__ __forceinline__ float mul_rn_s(float s, float b) {
    float r;
    asm("v_mul_f32_e32 %0, %1, %2" : "=v"(r) : "s"(s), "v"(b));
    return r;
}
__device__ __forceinline__ float sub_rn_s(float s, float b) {
    float r;
    asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "s"(s), "v"(b));
    return r;
}

template <typename T, int METRIC, int NQ>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void rescore_all_kernel(const T* __restrict__ corpus, uint32_t dim, uint32_t ld,
                                                         const float* __restrict__ q, RescoreQuerySet qs,
                                                         uint64_t nrows, float* __restrict__ out, uint64_t out_ld) {
    constexpr int EPU = 16 / (int)sizeof(T);
    constexpr int LPC = 64 / EPU;
    constexpr int RPI = 64 / LPC;
    constexpr int NI = 64 / RPI;
    __shared__ __attribute__((aligned(16))) float tile[64 * kTileStride];
    const int lane = threadIdx.x;
    const uint64_t row0 = (uint64_t)blockIdx.x * 64;
    const uint64_t slot = row0 + lane;
    const bool valid = slot < nrows;
    const int nvalid = nrows - row0 < 64 ? (int)(nrows - row0) : 64;   // rows are a prefix of the wave
    const int ni_used = (nvalid + RPI - 1) / RPI;
    const int sub = lane / LPC, part = lane % LPC;

    float acc[NQ];
#pragma unroll
    for (int n = 0; n < NQ; ++n) acc[n] = 0.0f;
    // ring depth: a bf16 chunk of 64 rows is 8 KB (8 x 16 B per lane), an fp32 chunk 16 KB
    constexpr int D = sizeof(T) == 4 ? 2 : 3;
    u32x4 v[D][NI];
    float vq[D][NQ];
    auto fetch = [&](u32x4 (&vs)[NI], float (&qv)[NQ], uint32_t j0) {
        const uint32_t e0 = j0 + part * EPU;
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            vs[it] = u32x4{0u, 0u, 0u, 0u};
            if (it < ni_used) {
                uint64_t row = row0 + it * RPI + sub;
                if (row >= nrows) row = row0;
                if (e0 < ld) vs[it] = *reinterpret_cast<const u32x4*>(corpus + row * ld + e0);
            }
        }
#pragma unroll
        for (int n = 0; n < NQ; ++n) qv[n] = j0 + lane < ld ? q[(uint64_t)qs.qi[n] * ld + j0 + lane] : 0.0f;
    };
#pragma unroll
    for (int d = 0; d < D; ++d)
        if ((uint32_t)d * 64 < dim) fetch(v[d], vq[d], d * 64);
    // one ring slot per call, the slot a compile-time constant (a `#pragma unroll` over d is
    // refused for the largest variants, and a runtime d would put the ring into scratch)
    auto chunk = [&](auto dc, uint32_t jb) __attribute__((always_inline)) -> bool {
            constexpr int d = decltype(dc)::value;
            const uint32_t j0 = jb + d * 64;
            if (j0 >= dim) return false;
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
            float qc[NQ];
#pragma unroll
            for (int n = 0; n < NQ; ++n) qc[n] = vq[d][n];
            if (j0 + 64 * D < dim) fetch(v[d], vq[d], j0 + 64 * D);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t jn = dim - j0 < 64 ? dim - j0 : 64;
            const float* trow = tile + lane * kTileStride;
            // elements [0, jn) of the chunk; the tail chunk's padding (x = q = 0 beyond dim) is
            // NOT walked: acc + (+0 * +0) would turn an accumulated -0.0 into +0.0
#pragma unroll
            for (uint32_t l = 0; l < 64; l += 4) {
                if (l < jn) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(trow + l);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (l + e < jn) {
#pragma unroll
                            for (int n = 0; n < NQ; ++n) {
                                const float qs_ = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qc[n]), (int)(l + e)));
                                if constexpr (METRIC == M_COSINE) {
                                    acc[n] = add_rn(acc[n], mul_rn_s(qs_, x[e]));
                                } else {
                                    const float dd = sub_rn_s(qs_, x[e]);
                                    acc[n] = add_rn(acc[n], mul_rn(dd, dd));
                                }
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            return true;
    };
    for (uint32_t jb = 0; jb < dim; jb += 64 * D) {
        if (!chunk(RingSlot<0>{}, jb)) break;
        if (!chunk(RingSlot<1>{}, jb)) break;
        if constexpr (D == 3)
            if (!chunk(RingSlot<2>{}, jb)) break;
    }
    if (valid) {
#pragma unroll
        for (int n = 0; n < NQ; ++n) out[(uint64_t)n * out_ld + slot] = acc[n];
    }
}

static void launch_rescore(const void* d_corpus, int dtype, int metric, uint32_t dim, uint32_t ld,
                           const float* d_q, int nq, const uint32_t* d_cand_rows, uint32_t kp,
                           float* d_out, hipStream_t s) {
    const size_t lds = ((size_t)ld + 64 * kTileStride) * sizeof(float);
    dim3 grid((kp + 63) / 64, nq);
#define VROD_RS(TT, MM) rescore_kernel<TT, MM><<<grid, 64, lds, s>>>((const TT*)d_corpus, dim, ld, d_q, d_cand_rows, kp, d_out)
    if (dtype == DT_BF16) { if (metric == M_COSINE) VROD_RS(bf16_t, M_COSINE); else VROD_RS(bf16_t, M_L2); }
    else { if (metric == M_COSINE) VROD_RS(float, M_COSINE); else VROD_RS(float, M_L2); }
#undef VROD_RS
}

void launch_rescore_candidates(const void* d_corpus, int dtype, int metric, uint32_t dim,
                               uint32_t ld, const float* d_q, int nq, const uint32_t* d_cand_rows,
                               uint32_t kp, float* d_out, hipStream_t s) {
    if (!nq || !kp) return;
    launch_rescore(d_corpus, dtype, metric, dim, ld, d_q, nq, d_cand_rows, kp, d_out, s);
}

int rescore_all_max_queries(uint32_t) { return 8; }

void launch_rescore_all(const void* d_corpus, int dtype, int metric, uint32_t dim, uint32_t ld,
                        const float* d_q, const uint32_t* query_index, int nq, uint64_t nrows,
                        float* d_out, uint64_t out_ld, hipStream_t s) {
    if (!nrows || nq <= 0) return;
    RescoreQuerySet qs;
    for (int i = 0; i < 8; ++i) qs.qi[i] = query_index[i < nq ? i : nq - 1];
    const unsigned grid = (unsigned)((nrows + 63) / 64);
#define VROD_RA(TT, MM, NN) rescore_all_kernel<TT, MM, NN><<<grid, 64, 0, s>>>((const TT*)d_corpus, dim, ld, d_q, qs, nrows, d_out, out_ld)
#define VROD_RA_N(TT, MM)                                                                           \
    switch (nq) {                                                                                   \
        case 1: VROD_RA(TT, MM, 1); break;                                                          \
        case 2: VROD_RA(TT, MM, 2); break;                                                          \
        case 4: VROD_RA(TT, MM, 4); break;                                                          \
        default: VROD_RA(TT, MM, 8); break;                                                         \
    }
    if (dtype == DT_BF16) { if (metric == M_COSINE) { VROD_RA_N(bf16_t, M_COSINE) } else { VROD_RA_N(bf16_t, M_L2) } }
    else { if (metric == M_COSINE) { VROD_RA_N(float, M_COSINE) } else { VROD_RA_N(float, M_L2) } }
#undef VROD_RA_N
#undef VROD_RA
}

}  // namespace vrod
