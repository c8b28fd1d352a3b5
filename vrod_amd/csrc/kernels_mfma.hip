// kernels_mfma.hip -- batched Q.K^T distance scan on the gfx950 matrix cores with a fused
// per-query threshold filter (the score matrix is never written to memory).
//
// Fills the batched half of the scan slot under SearchSimilarCommand::execute (reference
// src/command/types.rs:121-132, empty).  GEMM view: M = corpus rows, N = queries,
// K = vector dimension; both operands are stored [row][k] with k contiguous, so both MFMA
// fragments are 16-B LDS reads.
//
// Work-group = 512 threads = 8 waves (2 along M x 4 along N), output tile 256 rows x 256
// queries, wave tile 128 x 64 = 8 x 4 MFMA 16x16 tiles, 128 fp32 accumulators per lane.
// The corpus rows are the MFMA A operand and the queries the B operand, so an accumulator
// lane holds ONE query column per N-tile: its threshold is a register and the filter is a
// max/compare over the lane's own registers -- no cross-lane traffic in the common case.
//
// Staging: global -> LDS with global_load_lds_dwordx4 (LDS-DMA, 1 KB per wave instruction
// = 8 rows x one 128-B line), two LDS stages (K-step = 128 B per row: 64 bf16 / 32 fp32).
// The LDS image is lane-linear per piece, so the bank-conflict swizzle (16-B chunk index
// XOR row&7) is applied to the per-lane SOURCE address and again on the fragment read
// (cdna_hip_programming.md rule 21).
//
// Filter: a score that beats its query's read-only threshold is appended to an LDS log
// (one ds_add_rtn per hit); the log is flushed to per-query lists in HBM with global
// atomics (rare).  Thresholds are refreshed between launches (levels) by
// list_compact_kernel, so every launch of this kernel is a pure function of its inputs.
//
// Roofline: MFMA.  Algorithmic flops per launch = 2 * nq * rows * dim  (SURVEY.md 8d).
#include "vrod_common.h"
#include "vrod_kernels.h"

namespace vrod {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBM = 256, kBN = 256;
constexpr int kStageBytes = 64 * 1024;            // A (32 KB) + B (32 KB)
constexpr int kLogCap = 2048;                     // LDS log entries (8 B each)
constexpr int kLdsLog = 2 * kStageBytes;          // byte offset of the log
constexpr int kLdsCtl = kLdsLog + kLogCap * 8;    // [0] log count, [1..2] flush flags
constexpr int kLdsTotal = kLdsCtl + 64;

#define VROD_GLDS16(gptr, lptr)                                                               \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),   \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

struct MfmaKernelArgs {
    const char* corpus;
    const char* queries;
    const float* xnorm2;
    const float* qnorm2;
    const float* thr;
    uint2* lists;
    uint32_t* counts;
    uint32_t cap;
    uint32_t ld_bytes;      // bytes per row (multiple of 128)
    uint32_t nqb;           // query blocks of 256
    uint32_t tile_first;    // first 256-row tile of the launch
    uint32_t ntiles;        // tiles in the launch
    uint32_t row_lo;        // appends are limited to rows [row_lo, row_end)
    uint32_t row_end;
    uint32_t nstrips;       // corpus strips (8 * strips_per_xcd)
    uint32_t strips_per_xcd;
    uint32_t slots;         // work-groups per XCD label (gridDim.x / 8)
};

template <int METRIC>
__device__ __forceinline__ bool better(float a, float b) {
    return METRIC == M_COSINE ? a > b : a < b;
}

__device__ __forceinline__ void global_append(const MfmaKernelArgs& a, uint32_t gq, uint32_t bits, uint32_t row) {
    const uint32_t pos = atomicAdd(&a.counts[gq], 1u);
    if (pos < a.cap) a.lists[(uint64_t)gq * a.cap + pos] = make_uint2(bits, row);
}

// T = bf16_t: v_mfma_f32_16x16x32_bf16.  T = float: v_mfma_f32_16x16x4_f32 (exact fp32 fma chain).
template <typename T, int METRIC>
__global__ __launch_bounds__(512) void scan_mfma_kernel(const MfmaKernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    uint32_t* log_cnt = reinterpret_cast<uint32_t*>(lds + kLdsCtl);
    uint32_t* flush_flag = log_cnt + 1;  // [2]
    uint2* log = reinterpret_cast<uint2*>(lds + kLdsLog);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;

    // ---- which (strip, query block) this work-group owns.  blockIdx % 8 labels the XCD the
    // dispatcher tends to use, so the work-groups that share corpus tiles share an L2
    // (speed only; nothing depends on it).
    const uint32_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    uint32_t strip, qb0, qb_step;
    if (a.nqb <= a.slots) {
        if (slot >= a.strips_per_xcd * a.nqb) return;
        qb0 = slot % a.nqb;
        qb_step = a.nqb;  // single pass
        strip = xcd * a.strips_per_xcd + slot / a.nqb;
    } else {
        qb0 = slot;
        qb_step = a.slots;
        strip = xcd;
    }
    const uint32_t t0 = a.tile_first + (uint32_t)((uint64_t)a.ntiles * strip / a.nstrips);
    const uint32_t t1 = a.tile_first + (uint32_t)((uint64_t)a.ntiles * (strip + 1) / a.nstrips);
    if (t0 >= t1) return;

    if (tid == 0) { log_cnt[0] = 0; flush_flag[0] = 0; flush_flag[1] = 0; }
    __syncthreads();

    const uint32_t KT = a.ld_bytes >> 7;  // K-steps (128 B of every row) per tile
    const uint32_t rel_base = a.tile_first * kBM;

    // ---- per-lane staging offsets: piece = 8 rows x 128 B; lane -> (row lane>>3, chunk lane&7),
    // the source chunk is XOR-swizzled so that the linear LDS image is conflict-free to read
    const uint32_t st_row = lane >> 3;
    const uint32_t st_lane_off = st_row * a.ld_bytes + (((lane & 7) ^ st_row) << 4);
    // ---- per-lane fragment read offsets within a stage (A: +0, B: +32 KB)
    const uint32_t fr = lane & 15, fg = lane >> 4, r7 = fr & 7;
    // piece of row (base + m*16 + fr) = (base>>3) + m*2 + (fr>>3); byte = piece*1024 + r7*128 + ((c ^ r7) << 4)
    const uint32_t a_frag0 = ((wr * 16 + (fr >> 3)) << 10) + (r7 << 7);
    const uint32_t b_frag0 = 32768u + ((wc * 8 + (fr >> 3)) << 10) + (r7 << 7);
    const uint32_t c_off0 = ((0 * 4 + fg) ^ r7) << 4, c_off1 = ((1 * 4 + fg) ^ r7) << 4;

    for (uint32_t qb = qb0; qb < a.nqb; qb += qb_step) {
        // thresholds / norms of this lane's 4 query columns
        float thr[4], qn2[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const uint32_t gq = qb * kBN + wc * 64 + n * 16 + fr;
            thr[n] = a.thr[gq];
            qn2[n] = METRIC == M_L2 ? a.qnorm2[gq] : 0.0f;
        }
        const char* q_base = a.queries + (uint64_t)qb * kBN * a.ld_bytes;

        f32x4 acc[8][4];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

        const uint32_t total_it = (t1 - t0) * KT;

        auto stage = [&](uint32_t it, uint32_t buf) {
            const uint32_t tile = t0 + it / KT, kt = it % KT;
            const char* a_src = a.corpus + (uint64_t)tile * kBM * a.ld_bytes + (uint64_t)kt * 128 + st_lane_off;
            const char* b_src = q_base + (uint64_t)kt * 128 + st_lane_off;
            char* l = lds + buf * kStageBytes;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t p = wave * 4 + i;
                VROD_GLDS16(a_src + (uint64_t)p * 8 * a.ld_bytes, l + p * 1024);
                VROD_GLDS16(b_src + (uint64_t)p * 8 * a.ld_bytes, l + 32768 + p * 1024);
            }
        };

        stage(0, 0);
        __syncthreads();

        for (uint32_t it = 0; it < total_it; ++it) {
            const uint32_t buf = it & 1;
            const uint32_t kt = it % KT;
            const uint32_t tile = t0 + it / KT;
            // ---- log flush protocol (see header): decision by thread 0 at a tile's first
            // K-step, published by the barrier that ends that K-step, acted on here.
            if (it > 0 && ((it - 1) % KT) == 0) {
                const uint32_t prev_tile_parity = ((it - 1) / KT) & 1;
                if (flush_flag[prev_tile_parity]) {
                    const uint32_t n = log_cnt[0] < (uint32_t)kLogCap ? log_cnt[0] : (uint32_t)kLogCap;
                    for (uint32_t i = tid; i < n; i += 512) {
                        const uint2 e = log[i];
                        global_append(a, qb * kBN + (e.y >> 24), e.x, rel_base + (e.y & 0xFFFFFFu));
                    }
                    __syncthreads();
                    if (tid == 0) log_cnt[0] = 0;
                    __syncthreads();
                }
            }
            if (kt == 0 && tid == 0) flush_flag[(it / KT) & 1] = log_cnt[0] >= (uint32_t)(kLogCap / 2);

            if (it + 1 < total_it) stage(it + 1, buf ^ 1);

            // ---- fragments + MFMA for this K-step (two 64-B halves)
            const char* l = lds + buf * kStageBytes;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const uint32_t co = kk == 0 ? c_off0 : c_off1;
                if constexpr (sizeof(T) == 2) {
                    bf16x8 af[8], bfr[4];
#pragma unroll
                    for (int m = 0; m < 8; ++m) af[m] = *reinterpret_cast<const bf16x8*>(l + a_frag0 + m * 2048 + co);
#pragma unroll
                    for (int n = 0; n < 4; ++n) bfr[n] = *reinterpret_cast<const bf16x8*>(l + b_frag0 + n * 2048 + co);
#pragma unroll
                    for (int m = 0; m < 8; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bfr[n], acc[m][n], 0, 0, 0);
                } else {
                    f32x4 af[8], bfr[4];
#pragma unroll
                    for (int m = 0; m < 8; ++m) af[m] = *reinterpret_cast<const f32x4*>(l + a_frag0 + m * 2048 + co);
#pragma unroll
                    for (int n = 0; n < 4; ++n) bfr[n] = *reinterpret_cast<const f32x4*>(l + b_frag0 + n * 2048 + co);
                    // the 4 k-slots of one instruction are the 4 lane groups; element i of every
                    // lane's chunk is one instruction: any k permutation sums the same products
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int m = 0; m < 8; ++m)
#pragma unroll
                            for (int n = 0; n < 4; ++n)
                                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][i], bfr[n][i], acc[m][n], 0, 0, 0);
                }
            }

            // ---- tile finished: filter the 128 x 64 scores of this wave
            if (kt == KT - 1) {
                const uint32_t row_w = tile * kBM + wr * 128 + fg * 4;  // + m*16 + r
                bool hit[4];
                bool any = false;
                // L2: fast distance = |q|^2 + |x|^2 - 2 q.x ; the row norms are re-read per
                // 16-row block (L1/L2 hits) instead of held in 32 registers
                auto xnorm_of = [&](int m) -> f32x4 {
                    if constexpr (METRIC == M_L2) return *reinterpret_cast<const f32x4*>(a.xnorm2 + row_w + m * 16);
                    else return f32x4{0.f, 0.f, 0.f, 0.f};
                };
                auto score = [&](const f32x4& xv, int m, int n, int r) -> float {
                    if constexpr (METRIC == M_COSINE) return acc[m][n][r];
                    else return __builtin_fmaf(-2.0f, acc[m][n][r], xv[r] + qn2[n]);
                };
                float best[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) best[n] = worst_score(METRIC);
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const f32x4 xv = xnorm_of(m);
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float s = score(xv, m, n, r);
                            best[n] = METRIC == M_COSINE ? __builtin_fmaxf(best[n], s) : __builtin_fminf(best[n], s);
                        }
                }
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    hit[n] = better<METRIC>(best[n], thr[n]);
                    any |= hit[n];
                }
                if (__any(any)) {
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const f32x4 xv = xnorm_of(m);
#pragma unroll
                        for (int n = 0; n < 4; ++n) {
                            if (!__any(hit[n])) continue;
                            const uint32_t ql = wc * 64 + n * 16 + fr;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float s = score(xv, m, n, r);
                                const uint32_t row = row_w + m * 16 + r;
                                if (better<METRIC>(s, thr[n]) && row >= a.row_lo && row < a.row_end) {
                                    const uint32_t pos = atomicAdd(&log_cnt[0], 1u);
                                    if (pos < (uint32_t)kLogCap)
                                        log[pos] = make_uint2(__float_as_uint(s), (ql << 24) | (row - rel_base));
                                    else
                                        global_append(a, qb * kBN + ql, __float_as_uint(s), row);
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            __syncthreads();  // stage it+1 landed (vmcnt(0) + barrier); buf may be restaged
        }

        // ---- end of this query block: flush what is left in the log
        {
            const uint32_t n = log_cnt[0] < (uint32_t)kLogCap ? log_cnt[0] : (uint32_t)kLogCap;
            for (uint32_t i = tid; i < n; i += 512) {
                const uint2 e = log[i];
                global_append(a, qb * kBN + (e.y >> 24), e.x, rel_base + (e.y & 0xFFFFFFu));
            }
            __syncthreads();
            if (tid == 0) { log_cnt[0] = 0; flush_flag[0] = 0; flush_flag[1] = 0; }
            __syncthreads();
        }
    }
}

void launch_scan_mfma(const MfmaScanArgs& h, int dtype, int num_cus, hipStream_t s) {
    if (h.row_end <= h.row_begin) return;
    MfmaKernelArgs a{};
    a.corpus = (const char*)h.corpus;
    a.queries = (const char*)h.queries;
    a.xnorm2 = h.xnorm2;
    a.qnorm2 = h.qnorm2;
    a.thr = h.thr;
    a.lists = h.lists;
    a.counts = h.counts;
    a.cap = h.cap;
    a.ld_bytes = h.ld * (dtype == DT_BF16 ? 2u : 4u);
    a.nqb = h.nq_pad / kBN;
    a.tile_first = h.row_begin / kBM;
    a.ntiles = (h.row_end + kBM - 1) / kBM - a.tile_first;
    a.row_lo = h.row_begin;
    a.row_end = h.row_end;
    int grid = num_cus / 8 * 8;
    if (grid < 8) grid = 8;
    a.slots = grid / 8;
    a.strips_per_xcd = a.nqb <= a.slots ? a.slots / a.nqb : 1;
    a.nstrips = 8 * a.strips_per_xcd;
    // fewer tiles than strips: shrink the strip count so no strip is empty more than needed
#define VROD_MFMA(TT, MM)                                                                                   \
    do {                                                                                                    \
        static bool attr_set = false;                                                                       \
        if (!attr_set) {                                                                                    \
            (void)hipFuncSetAttribute((const void*)scan_mfma_kernel<TT, MM>,                                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);               \
            attr_set = true;                                                                                \
        }                                                                                                   \
        scan_mfma_kernel<TT, MM><<<grid, 512, kLdsTotal, s>>>(a);                                           \
    } while (0)
    if (dtype == DT_BF16) { if (h.metric == M_COSINE) VROD_MFMA(bf16_t, M_COSINE); else VROD_MFMA(bf16_t, M_L2); }
    else { if (h.metric == M_COSINE) VROD_MFMA(float, M_COSINE); else VROD_MFMA(float, M_L2); }
#undef VROD_MFMA
}

}  // namespace vrod
