// kernels_prep.hip -- insert/query preparation for libvrod_hip (gfx950).
//
// Fills the slot vRod leaves empty under BulkInsertCommand::execute / InsertCommand::execute
// (reference src/command/types.rs:56-80): rows arrive as the reference's Vec<Vec<f32>>
// (src/utils/embeddings.rs:29) and are stored row-major, padded to `ld` elements
// (128-B multiples) in HBM, either fp32 or bf16.
//
// Spec (DESIGN.md "Scan spec", mirrored by oracle/vrod_oracle.c orc_prepare_row):
//   COSINE: x / sqrt(sum x^2), the sum accumulated left to right in fp64, the divide in
//   fp64, then rounded to fp32; zero rows stay zero.  BF16: round to nearest even.
#include "vrod_common.h"
#include "vrod_kernels.h"

namespace vrod {

// ------------------------------------------------------------------ synthetic stream
// One wave per row; the sum of squares is an exact integer (< 2^53), so the lane order
// does not matter and the result equals orc_synth_row_f32 bit for bit.
__global__ __launch_bounds__(256) void synth_rows_kernel(uint64_t key, uint64_t first_row,
                                                         uint64_t n, uint32_t dim,
                                                         float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t r = wave; r < n; r += nwaves) {
        const uint64_t base = (first_row + r) * (uint64_t)dim;
        unsigned long long ss = 0;
        for (uint32_t j = lane; j < dim; j += 64) {
            long long v = synth_int(key, base + j);
            ss += (unsigned long long)(v * v);
        }
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        float* o_row = out + r * (uint64_t)dim;
        if (ss == 0) {
            for (uint32_t j = lane; j < dim; j += 64) o_row[j] = 0.0f;
        } else {
            const double nrm = __builtin_sqrt((double)ss);
            for (uint32_t j = lane; j < dim; j += 64)
                o_row[j] = (float)((double)synth_int(key, base + j) / nrm);
        }
    }
}

// ------------------------------------------------------------------ row norms (fp64, sequential)
// One thread per row walks its row left to right: the accumulation order is the spec's.
// v*v is exact in fp64 for fp32 v, so fma(v, v, ss) == ss + v*v rounded once.
__global__ __launch_bounds__(256) void row_norm_kernel(const float* __restrict__ in, uint64_t n,
                                                       uint32_t dim, double* __restrict__ nrm,
                                                       uint32_t* __restrict__ bad_flag) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float* x = in + r * (uint64_t)dim;
    double ss = 0.0;
    bool bad = false;
    for (uint32_t j = 0; j < dim; ++j) {
        const float f = x[j];
        bad |= !(__builtin_fabsf(f) <= 3.4028234663852886e38f);  // NaN or Inf
        const double v = (double)f;
        ss = __builtin_fma(v, v, ss);
    }
    nrm[r] = __builtin_sqrt(ss);
    if (bad) atomicOr(bad_flag, 1u);
}

// ------------------------------------------------------------------ write prepared rows
// Elementwise, coalesced: out[r][j] for j < ld (zero padding beyond dim).
// NORMALISE: divide by the fp64 norm (0 -> zero row).  Emits fp32 and/or bf16 copies.
template <bool NORMALISE>
__global__ __launch_bounds__(256) void row_write_kernel(const float* __restrict__ in, uint64_t n,
                                                        uint32_t dim, uint32_t ld,
                                                        const double* __restrict__ nrm,
                                                        int round_bf16,
                                                        float* __restrict__ out_f32,
                                                        bf16_t* __restrict__ out_bf16) {
    const uint64_t total = n * (uint64_t)ld;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / ld;
        const uint32_t j = (uint32_t)(i - r * ld);
        float v = 0.0f;
        if (j < dim) {
            v = in[r * (uint64_t)dim + j];
            if (NORMALISE) {
                const double d = nrm[r];
                v = d == 0.0 ? 0.0f : (float)((double)v / d);
            }
        }
        if (round_bf16) {
            const bf16_t h = f32_to_bf16_rne(v);
            if (out_bf16) out_bf16[i] = h;
            v = bf16_to_f32(h);
        }
        if (out_f32) out_f32[i] = v;
    }
}

// ------------------------------------------------------------------ fast squared norms of stored rows
// fp32, one wave per row, 16-B loads (rows are whole 128-B lines: ld*sizeof(T) % 128 == 0).
// Used by the L2 fast pass and for the certificate's bound.
template <typename T>
__global__ __launch_bounds__(256) void row_fastnorm_kernel(const T* __restrict__ rows, uint64_t n,
                                                           uint32_t ld, float* __restrict__ xn2,
                                                           uint32_t* __restrict__ max_bits) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint32_t units = ld * (uint32_t)sizeof(T) / 16;
    float wave_max = 0.0f;   // one atomic per wave, not per row (every row hits the same address)
    for (uint64_t r = wave; r < n; r += nwaves) {
        const u32x4* x = reinterpret_cast<const u32x4*>(rows + r * (uint64_t)ld);
        float s = 0.0f;
        for (uint32_t j = lane; j < units; j += 64) {
            const u32x4 v = x[j];
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = __uint_as_float(v[e] << 16), hi = __uint_as_float(v[e] & 0xFFFF0000u);
                    s = __builtin_fmaf(lo, lo, s);
                    s = __builtin_fmaf(hi, hi, s);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float f = __uint_as_float(v[e]); s = __builtin_fmaf(f, f, s); }
            }
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) xn2[r] = s;
        wave_max = __builtin_fmaxf(wave_max, s);
    }
    if (lane == 0 && wave_max > 0.0f) atomicMax(max_bits, __float_as_uint(wave_max));  // >= 0: uint order == float order
}

// ------------------------------------------------------------------ widen stored rows back to fp32
template <typename T>
__global__ __launch_bounds__(256) void rows_get_kernel(const T* __restrict__ rows, uint64_t n,
                                                       uint32_t dim, uint32_t ld,
                                                       float* __restrict__ out) {
    const uint64_t total = n * (uint64_t)dim;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / dim;
        const uint32_t j = (uint32_t)(i - r * dim);
        if constexpr (sizeof(T) == 2) out[i] = bf16_to_f32(rows[r * (uint64_t)ld + j]);
        else out[i] = rows[r * (uint64_t)ld + j];
    }
}

// ------------------------------------------------------------------ query preparation, one launch
// One block per (padded) query: stage the row in LDS, thread 0 walks it left to right in fp64
// (the spec's order), every thread then writes the normalised / bf16-rounded row (zero padding
// to ld, zero rows for q >= nq) and the block reduces the fast squared norm.
template <bool NORMALISE>
__global__ __launch_bounds__(256) void prep_queries_kernel(const float* __restrict__ in, uint32_t nq, uint32_t dim,
                                                           uint32_t ld, int round_bf16, float* __restrict__ out_f32,
                                                           bf16_t* __restrict__ out_bf16, float* __restrict__ qn2,
                                                           uint32_t* __restrict__ bad_flag,
                                                           uint32_t* __restrict__ max_bits, QueryInit init) {
    extern __shared__ __attribute__((aligned(16))) float row[];  // [dim]
    // per-search state, reset here instead of by separate memset launches
    if (threadIdx.x == 0) {
        init.status[blockIdx.x] = 0u;
        if (init.counts) init.counts[blockIdx.x] = 0u;
        if (init.thr) init.thr[blockIdx.x] = __uint_as_float(blockIdx.x < nq ? init.thr_live_bits : init.thr_pad_bits);
    }
    if (blockIdx.x == 0 && init.zero_words)
        for (uint32_t i = threadIdx.x; i < init.n_zero_words; i += 256) init.zero_words[i] = 0u;
    if (blockIdx.x == 0 && init.zero_words2)
        for (uint32_t i = threadIdx.x; i < init.n_zero_words2; i += 256) init.zero_words2[i] = 0u;
    __shared__ double s_nrm;
    __shared__ float s_part[4];
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const bool live = q < nq;
    bool bad = false;
    if (live) {
        for (uint32_t j = tid; j < dim; j += 256) {
            const float f = in[(uint64_t)q * dim + j];
            bad |= !(__builtin_fabsf(f) <= 3.4028234663852886e38f);
            row[j] = f;
        }
    }
    __syncthreads();
    if (tid == 0) {
        double ss = 0.0;
        if (live && NORMALISE) {
            // same left-to-right fp64 chain, fed by 16-B LDS reads issued ahead of it (one scalar
            // LDS read per step left the chain waiting ~100 cycles per element: 23 us at d = 768)
            typedef float f32x4_t __attribute__((ext_vector_type(4)));
            uint32_t j = 0;
#pragma unroll 4
            for (; j + 8 <= dim; j += 8) {
                const f32x4_t a = *reinterpret_cast<const f32x4_t*>(row + j);
                const f32x4_t b = *reinterpret_cast<const f32x4_t*>(row + j + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { const double v = (double)a[e]; ss = __builtin_fma(v, v, ss); }
#pragma unroll
                for (int e = 0; e < 4; ++e) { const double v = (double)b[e]; ss = __builtin_fma(v, v, ss); }
            }
            for (; j < dim; ++j) { const double v = (double)row[j]; ss = __builtin_fma(v, v, ss); }
        }
        s_nrm = __builtin_sqrt(ss);
    }
    __syncthreads();
    const double nrm = s_nrm;
    float fs = 0.0f;
    for (uint32_t j = tid; j < ld; j += 256) {
        float v = 0.0f;
        if (live && j < dim) {
            v = row[j];
            if (NORMALISE) v = nrm == 0.0 ? 0.0f : (float)((double)v / nrm);
        }
        if (round_bf16) {
            const bf16_t h = f32_to_bf16_rne(v);
            if (out_bf16) out_bf16[(uint64_t)q * ld + j] = h;
            v = bf16_to_f32(h);
        }
        out_f32[(uint64_t)q * ld + j] = v;
        fs = __builtin_fmaf(v, v, fs);
    }
    for (int o = 32; o > 0; o >>= 1) fs += __shfl_xor(fs, o);
    if ((tid & 63) == 0) s_part[tid >> 6] = fs;
    __syncthreads();
    if (tid == 0) {
        const float t = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        qn2[q] = t;
        if (live) atomicMax(max_bits, __float_as_uint(t));
    }
    if (bad) atomicOr(bad_flag, 1u);
}

// ------------------------------------------------------------------ bf16 split of fp32 rows
// x = hi + lo + r with hi = bf16(x), lo = bf16(x - hi) (the subtraction is exact in fp32),
// |r| <= 2^-16 |x|.  The batched scan of an fp32 corpus can then run on the bf16 matrix cores as
// q.x ~ hi_q.hi_x + hi_q.lo_x + lo_q.hi_x (16x the fp32 MFMA rate for 3x the products); the
// certificate's bound covers the representation error (vrod_index.hip).
// Both layouts are interleaved per K-tile j (64 elements = one 128-B line), so that the scan walks
// both rows forward and reads the corpus planes from HBM once:
// MODE 0 (corpus):  out row = [hi_0 | lo_0 | hi_1 | lo_1 | ...]            (2*ldp elements)
// MODE 1 (queries): out row = [hi_0 | lo_0 | hi_0 | hi_1 | lo_1 | hi_1 ...] (3*ldp elements)
// and K-tile 3j, 3j+1, 3j+2 of the product pairs corpus (hi_j, hi_j, lo_j) with queries (hi_j, lo_j, hi_j).
template <int MODE>
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ in, uint64_t n, uint32_t ld,
                                                         uint32_t ldp, bf16_t* __restrict__ out) {
    constexpr uint32_t SEG = MODE == 0 ? 2u : 3u;
    const uint64_t total = n * (uint64_t)ldp;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / ldp;
        const uint32_t j = (uint32_t)(i - r * ldp);
        const float x = j < ld ? in[r * ld + j] : 0.0f;
        const bf16_t hi = f32_to_bf16_rne(x);
        const bf16_t lo = f32_to_bf16_rne(x - bf16_to_f32(hi));
        bf16_t* o = out + r * (uint64_t)(SEG * ldp) + (uint64_t)(j >> 6) * (SEG * 64) + (j & 63);
        o[0] = hi;
        o[64] = lo;
        if (MODE == 1) o[128] = hi;
    }
}

void launch_split_rows(const float* d_in, uint64_t n, uint32_t ld, uint32_t ldp, void* d_out, bool queries, hipStream_t s) {
    if (!n) return;
    const uint64_t work = n * (uint64_t)ldp;
    uint64_t g = (work + 255) / 256;
    if (g > 256ull * 16) g = 256ull * 16;
    if (queries) split_rows_kernel<1><<<(unsigned)g, 256, 0, s>>>(d_in, n, ld, ldp, (bf16_t*)d_out);
    else split_rows_kernel<0><<<(unsigned)g, 256, 0, s>>>(d_in, n, ld, ldp, (bf16_t*)d_out);
}

// ------------------------------------------------------------------ launchers
static inline int grid_for(uint64_t work, int block, int cap = 256 * 8) {
    uint64_t g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > (uint64_t)cap) g = cap;
    return (int)g;
}

void launch_synth_rows(uint64_t seed, uint64_t first_row, uint64_t n, uint32_t dim, float* d_out,
                       hipStream_t s) {
    if (!n) return;
    const uint64_t key = splitmix64(seed);
    synth_rows_kernel<<<grid_for(n, 4), 256, 0, s>>>(key, first_row, n, dim, d_out);
}

void launch_prepare_rows(const float* d_in, uint64_t n, uint32_t dim, uint32_t ld, int metric,
                         int dtype, double* d_nrm_ws, uint32_t* d_bad_flag, float* d_out_f32,
                         void* d_out_bf16, hipStream_t s) {
    if (!n) return;
    // the norm pass also performs the NaN/Inf check, so it always runs
    row_norm_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(d_in, n, dim, d_nrm_ws, d_bad_flag);
    const int g = grid_for(n * (uint64_t)ld, 256);
    const int rb = dtype == DT_BF16;
    if (metric == M_COSINE)
        row_write_kernel<true><<<g, 256, 0, s>>>(d_in, n, dim, ld, d_nrm_ws, rb, d_out_f32,
                                                 (bf16_t*)d_out_bf16);
    else
        row_write_kernel<false><<<g, 256, 0, s>>>(d_in, n, dim, ld, d_nrm_ws, rb, d_out_f32,
                                                  (bf16_t*)d_out_bf16);
}

void launch_prep_queries(const float* d_in, uint32_t nq, uint32_t nq_pad, uint32_t dim, uint32_t ld, int metric,
                         int dtype, float* d_out_f32, void* d_out_bf16, float* d_qn2, uint32_t* d_bad_flag,
                         uint32_t* d_max_bits, const QueryInit& init, hipStream_t s) {
    if (!nq_pad) return;
    const size_t lds = (size_t)dim * sizeof(float);
    const int rb = dtype == DT_BF16;
    if (metric == M_COSINE)
        prep_queries_kernel<true><<<nq_pad, 256, lds, s>>>(d_in, nq, dim, ld, rb, d_out_f32, (bf16_t*)d_out_bf16, d_qn2, d_bad_flag, d_max_bits, init);
    else
        prep_queries_kernel<false><<<nq_pad, 256, lds, s>>>(d_in, nq, dim, ld, rb, d_out_f32, (bf16_t*)d_out_bf16, d_qn2, d_bad_flag, d_max_bits, init);
}

void launch_row_fastnorm(const void* d_rows, int dtype, uint64_t n, uint32_t ld, float* d_xn2,
                         uint32_t* d_max_bits, hipStream_t s) {
    if (!n) return;
    if (dtype == DT_BF16)
        row_fastnorm_kernel<bf16_t><<<grid_for(n, 4), 256, 0, s>>>((const bf16_t*)d_rows, n, ld, d_xn2, d_max_bits);
    else
        row_fastnorm_kernel<float><<<grid_for(n, 4), 256, 0, s>>>((const float*)d_rows, n, ld, d_xn2, d_max_bits);
}

void launch_rows_get(const void* d_rows, int dtype, uint64_t n, uint32_t dim, uint32_t ld,
                     float* d_out, hipStream_t s) {
    if (!n) return;
    const int g = grid_for(n * (uint64_t)dim, 256);
    if (dtype == DT_BF16)
        rows_get_kernel<bf16_t><<<g, 256, 0, s>>>((const bf16_t*)d_rows, n, dim, ld, d_out);
    else
        rows_get_kernel<float><<<g, 256, 0, s>>>((const float*)d_rows, n, dim, ld, d_out);
}

}  // namespace vrod
