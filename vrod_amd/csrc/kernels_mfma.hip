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
// = 8 rows x one 128-B line), two LDS stages (K-tile = 128 B per row: 64 bf16 / 32 fp32).
// The LDS image is lane-linear per piece, so the bank-conflict swizzle (16-B chunk index
// XOR row&7) is applied to the per-lane SOURCE address and again on the fragment read
// (cdna_hip_programming.md rule 21).
//
// Two schedules of the same tile:
//   scan_mfma_phased_kernel (default): 4 phases per K-tile (one 64x32 accumulator quadrant
//     = 16 MFMAs each).  The two wave groups (rows 0-127 / 128-255; one wave of each per
//     SIMD) run half a phase apart -- group 1 executes one extra s_barrier up front -- so
//     on every SIMD one wave issues its LDS fragment reads and LDS-DMA staging while its
//     partner runs MFMAs.  Raw s_barrier + counted s_waitcnt vmcnt(4): the staging of the
//     next K-tile stays in flight across barriers and is never drained in the loop.
//   scan_mfma_kernel: the plain double-buffered form (one __syncthreads per K-tile), kept
//     as the A/B reference (VROD_MFMA_SIMPLE=1).
//
// Filter: a score that beats its query's read-only threshold is appended to an LDS log
// (one ds_add_rtn per hit); the log is flushed to per-query lists in HBM with global
// atomics (rare).  Thresholds are refreshed between launches (levels) by
// list_compact_kernel, so every launch of this kernel is a pure function of its inputs.
//
// Roofline: MFMA.  Algorithmic flops per launch = 2 * nq * rows * dim  (SURVEY.md 8d).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>

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
// 4-wave kernel only: the work-group's 256 thresholds / query norms and two 256-row slots of row
// norms live in LDS, so that the tile epilogue issues no global load (a compiler-counted load
// there would drain the LDS-DMA pieces in flight)
constexpr int kLdsThr = kLdsTotal;                // [256] f32
constexpr int kLdsQn2 = kLdsThr + 1024;           // [256] f32
constexpr int kLdsXn2 = kLdsQn2 + 1024;           // [2][256] f32, slot = tile parity
constexpr int kLdsTotalW4 = kLdsXn2 + 2048;
constexpr int kLogCapW4 = kLogCap / 4;            // log entries per wave (4-wave kernel: wave-private segments)

// raw s_barrier (no vmcnt drain) fenced for the compiler only: memory operations may not be
// moved across it, nothing is emitted for the fences
#define VROD_BARRIER()                          \
    do {                                        \
        asm volatile("" ::: "memory");          \
        __builtin_amdgcn_s_barrier();           \
        asm volatile("" ::: "memory");          \
    } while (0)

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
    uint32_t ld_bytes;      // bytes per row (multiple of 128); in the 4-wave kernel: per QUERY row = the K extent
    uint32_t lda_bytes;     // 4-wave kernel, SPLIT form: bytes per CORPUS row (the K extent is 3/2 of it: split-bf16
    uint32_t a_wrap;        // pass over fp32 rows, kernels_prep.hip split_rows_kernel); a_wrap != 0 selects SPLIT
    uint32_t nqb;           // query blocks of 256
    uint32_t tile_first;    // first 256-row tile of the launch
    uint32_t ntiles;        // tiles in the launch
    uint32_t row_lo;        // appends are limited to rows [row_lo, row_end)
    uint32_t row_end;
    uint32_t nstrips;       // corpus strips (8 * strips_per_xcd)
    uint32_t strips_per_xcd;
    uint32_t slots;         // work-groups per XCD label (gridDim.x / 8)
    float* dense_out;       // DENSE launches: fast scores [nq_pad][dense_ld], column = row - row_lo
    uint32_t dense_ld;
    uint32_t* pace;         // [nstrips] arrival counters of the sibling work-groups (zeroed per launch)
    uint32_t pace_every;    // re-align the siblings of a strip every this many tiles (0 = never)
    uint32_t qb_base;       // 4-wave kernel: first query block of this launch (nqb <= slots per launch)
    uint32_t dense_group;   // DENSE launch of the 2 x 2 4-wave kernel: write one score per (query, group of 32 rows) -- the
                            // best of the group -- to dense_out[query][group], dense_ld groups per query
};

template <int METRIC>
__device__ __forceinline__ bool better(float a, float b) {
    return METRIC == M_COSINE ? a > b : a < b;
}

__device__ __forceinline__ void global_append(const MfmaKernelArgs& a, uint32_t gq, uint32_t bits, uint32_t row) {
    const uint32_t pos = atomicAdd(&a.counts[gq], 1u);
    if (pos < a.cap) a.lists[(uint64_t)gq * a.cap + pos] = make_uint2(bits, row);
}

// (strip, query block) of a work-group.  blockIdx % 8 labels the XCD the dispatcher tends to
// use, so the work-groups that share corpus tiles share an L2 (speed only).
__device__ __forceinline__ bool wg_assignment(const MfmaKernelArgs& a, uint32_t& strip, uint32_t& qb0,
                                              uint32_t& qb_step) {
    const uint32_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (a.nqb <= a.slots) {
        if (slot >= a.strips_per_xcd * a.nqb) return false;
        qb0 = slot % a.nqb;
        qb_step = a.nqb;  // single pass
        strip = xcd * a.strips_per_xcd + slot / a.nqb;
    } else {
        qb0 = slot;
        qb_step = a.slots;
        strip = xcd;
    }
    return true;
}

// log_cnt[0..2] = 0 with the zero made in a VGPR on the spot.  (As a plain store hipcc keeps a
// zero vector alive across the whole kernel -- in AGPRs where it may, which the 4-wave kernel
// owns: scripts/audit_w4.py.)
__device__ __forceinline__ void lds_zero3(uint32_t* p) {
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)p;
    asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %1 offset:4\n\tds_write_b32 %0, %1 offset:8\n\ts_waitcnt lgkmcnt(0)"
                 :: "v"(addr), "v"(0u) : "memory");
}

// Drain the LDS log into the per-query lists.  Called by ALL threads at the same program
// point; the leading barrier makes sure every wave's appends (a wave group may still be in
// its tile filter) are in the log.  Uses three block barriers.
__device__ __forceinline__ void flush_log(const MfmaKernelArgs& a, uint2* log, uint32_t* log_cnt, uint32_t qb,
                                          uint32_t rel_base, int tid) {
    __syncthreads();
    const uint32_t n = log_cnt[0] < (uint32_t)kLogCap ? log_cnt[0] : (uint32_t)kLogCap;
    for (uint32_t i = tid; i < n; i += blockDim.x) {
        const uint2 e = log[i];
        global_append(a, qb * kBN + (e.y >> 24), e.x, rel_base + (e.y & 0xFFFFFFu));
    }
    __syncthreads();
    if (tid == 0) lds_zero3(log_cnt);
    __syncthreads();
}

// The fused filter: the wave's 128 x 64 scores against the 4 per-lane thresholds.
// Returns true when the LDS log passed half of its capacity (a flush is due).
template <int METRIC>
__device__ __forceinline__ void filter_tile(const MfmaKernelArgs& a, f32x4 (&acc)[8][4], const float (&thr)[4],
                                            const float (&qn2)[4], uint32_t row_w, uint32_t ql0, uint32_t qb,
                                            uint32_t rel_base, uint2* log, uint32_t* log_cnt) {
    // LDS byte addresses of the log and its counter (for the inline-asm appends)
    const uint32_t lds_log_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)log;
    const uint32_t lds_cnt_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)log_cnt;
    bool hit[4];
    bool any = false;
    // L2: fast distance = |q|^2 + |x|^2 - 2 q.x ; the row norms are re-read per 16-row block
    // (L1/L2 hits) instead of held in 32 registers
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
                // one test per 16x16 tile (4 scores per lane) before the 4 predicated append sites
                const float s0 = score(xv, m, n, 0), s1 = score(xv, m, n, 1), s2 = score(xv, m, n, 2), s3 = score(xv, m, n, 3);
                const float tb = METRIC == M_COSINE ? __builtin_fmaxf(__builtin_fmaxf(s0, s1), __builtin_fmaxf(s2, s3))
                                                    : __builtin_fminf(__builtin_fminf(s0, s1), __builtin_fminf(s2, s3));
                if (!__any(better<METRIC>(tb, thr[n]))) continue;
                const uint32_t ql = ql0 + n * 16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s = r == 0 ? s0 : r == 1 ? s1 : r == 2 ? s2 : s3;
                    const uint32_t row = row_w + m * 16 + r;
                    if (better<METRIC>(s, thr[n]) && row >= a.row_lo && row < a.row_end) {
                        // LDS log append in inline asm: as compiler-visible LDS accesses these would
                        // each be preceded by s_waitcnt vmcnt(0) (they may alias the LDS-DMA
                        // destination), draining the staging pipeline on every hit.
                        uint32_t pos;
                        asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                                     : "=v"(pos) : "v"(lds_cnt_addr), "v"(1u) : "memory");
                        if (pos < (uint32_t)kLogCap) {
                            const uint64_t e = ((uint64_t)((ql << 24) | (row - rel_base)) << 32) | __float_as_uint(s);
                            asm volatile("ds_write_b64 %0, %1" :: "v"(lds_log_addr + pos * 8u), "v"(e) : "memory");
                            if (pos >= (uint32_t)(kLogCap / 2))  // sticky "flush due"
                                asm volatile("ds_write_b32 %0, %1" :: "v"(lds_cnt_addr + 12u), "v"(1u) : "memory");
                        } else {
                            global_append(a, qb * kBN + ql, __float_as_uint(s), row);
                        }
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the asm LDS writes above are not tracked by the compiler
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// Dense epilogue (sample pass): write the wave's 128 x 64 fast scores instead of filtering.
// A lane holds 4 consecutive rows of one query column per 16x16 tile -> one 16-B store each.
template <int METRIC>
__device__ __forceinline__ void dense_store_tile(const MfmaKernelArgs& a, f32x4 (&acc)[8][4], const float (&qn2)[4],
                                                 uint32_t row_w, uint32_t gq0) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const uint32_t row = row_w + m * 16;
        f32x4 xv = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (METRIC == M_L2) xv = *reinterpret_cast<const f32x4*>(a.xnorm2 + row);
        if (row - a.row_lo < a.dense_ld) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                f32x4 sc;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    sc[r] = METRIC == M_COSINE ? acc[m][n][r] : __builtin_fmaf(-2.0f, acc[m][n][r], xv[r] + qn2[n]);
                *reinterpret_cast<f32x4*>(a.dense_out + (uint64_t)(gq0 + n * 16) * a.dense_ld + (row - a.row_lo)) = sc;
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// ---------------------------------------------------------------------------------------------
// Schedule 1 (reference): double-buffered, one __syncthreads per K-tile.
// T = bf16_t: v_mfma_f32_16x16x32_bf16.  T = float: v_mfma_f32_16x16x4_f32 (exact fp32 fma chain).
// ---------------------------------------------------------------------------------------------
template <typename T, int METRIC>
__global__ __launch_bounds__(512) void scan_mfma_kernel(const MfmaKernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    uint32_t* log_cnt = reinterpret_cast<uint32_t*>(lds + kLdsCtl);  // [0] count [1..2] flags [3] due
    uint32_t* flush_flag = log_cnt + 1;
    uint2* log = reinterpret_cast<uint2*>(lds + kLdsLog);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    uint32_t strip, qb0, qb_step;
    if (!wg_assignment(a, strip, qb0, qb_step)) return;
    const uint32_t t0 = a.tile_first + (uint32_t)((uint64_t)a.ntiles * strip / a.nstrips);
    const uint32_t t1 = a.tile_first + (uint32_t)((uint64_t)a.ntiles * (strip + 1) / a.nstrips);
    if (t0 >= t1) return;

    if (tid == 0) { log_cnt[0] = 0; log_cnt[1] = 0; log_cnt[2] = 0; log_cnt[3] = 0; }
    __syncthreads();

    const uint32_t KT = a.ld_bytes >> 7;  // K-tiles (128 B of every row) per corpus tile
    const uint32_t rel_base = a.tile_first * kBM;
    const uint32_t st_row = lane >> 3;
    const uint32_t st_lane_off = st_row * a.ld_bytes + (((lane & 7) ^ st_row) << 4);
    const uint32_t fr = lane & 15, fg = lane >> 4, r7 = fr & 7;
    const uint32_t a_frag0 = ((wr * 16 + (fr >> 3)) << 10) + (r7 << 7);
    const uint32_t b_frag0 = 32768u + ((wc * 8 + (fr >> 3)) << 10) + (r7 << 7);
    const uint32_t c_off0 = ((0 * 4 + fg) ^ r7) << 4, c_off1 = ((1 * 4 + fg) ^ r7) << 4;

    for (uint32_t qb = qb0; qb < a.nqb; qb += qb_step) {
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
            // log flush: decided by thread 0 at a tile's first K-tile, published by the barrier
            // that ends it, acted on at the top of the next iteration (flag slot by tile parity)
            if (it > 0 && ((it - 1) % KT) == 0) {
                if (flush_flag[((it - 1) / KT) & 1]) flush_log(a, log, log_cnt, qb, rel_base, tid);
            }
            if (kt == 0 && tid == 0) flush_flag[(it / KT) & 1] = log_cnt[0] >= (uint32_t)(kLogCap / 2);

            if (it + 1 < total_it) stage(it + 1, buf ^ 1);

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
            if (kt == KT - 1)
                filter_tile<METRIC>(a, acc, thr, qn2, tile * kBM + wr * 128 + fg * 4, wc * 64 + fr, qb, rel_base, log, log_cnt);
            __syncthreads();  // stage it+1 landed (vmcnt(0) + barrier); buf may be restaged
        }
        flush_log(a, log, log_cnt, qb, rel_base, tid);
    }
}

// ---------------------------------------------------------------------------------------------
// Schedule 2 (default): staggered wave groups, 4 phases per K-tile, counted vmcnt.
//
// Phase p of a K-tile works on accumulator quadrant (mh, nh) = (0,0) (0,1) (1,1) (1,0):
//   LOAD    : ds_read the fragments the quadrant needs that are not in registers yet
//             (p0: A rows mh=0 + B cols nh=0, p1: B nh=1, p2: A mh=1, p3: B nh=0), and issue
//             2 LDS-DMA pieces of one staging unit of the NEXT K-tile
//             (p0: A_m0, p1: B_n0, p2: B_n1, p3: A_m1 -- the order of first use);
//             s_waitcnt vmcnt(4): all but the 2 youngest units this wave issued have landed
//   s_barrier
//   COMPUTE : 16 MFMAs (4 x 2 tiles x 2 k-halves), s_setprio 1 around them
//   s_barrier
// Group 1 (waves 4-7) runs one barrier behind group 0, so LOAD of one group overlaps COMPUTE
// of the other on every SIMD.  Hazards (cdna_hip_programming.md "Read a staged buffer one
// phase AFTER the wait that retires it"): a unit is read >= 1 phase after every wave's
// vmcnt wait for it plus a barrier, and restaged >= 2 phases after its last read.
// ---------------------------------------------------------------------------------------------
// GP = how many of a phase's 2 LDS-DMA pieces are issued inside the MFMA cluster instead of next
// to the ds_reads (an LDS-DMA issue is ~2-3x cheaper among MFMAs than beside LDS reads, and
// the load segment is the one that must not outlast the partner's 256-cycle MFMA segment).
// The unit issue order per wave is unchanged, so the counted wait is vmcnt(4 - GP).
// DENSE: the epilogue stores every score (sample pass) instead of filtering against thresholds.
template <typename T, int METRIC, int GP, bool DENSE>
__global__ __launch_bounds__(512) void scan_mfma_phased_kernel(const MfmaKernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    uint32_t* log_cnt = reinterpret_cast<uint32_t*>(lds + kLdsCtl);  // [0] count [3] flush due
    uint2* log = reinterpret_cast<uint2*>(lds + kLdsLog);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    uint32_t strip, qb0, qb_step;
    if (!wg_assignment(a, strip, qb0, qb_step)) return;
    const uint32_t t0 = a.tile_first + (uint32_t)((uint64_t)a.ntiles * strip / a.nstrips);
    const uint32_t t1 = a.tile_first + (uint32_t)((uint64_t)a.ntiles * (strip + 1) / a.nstrips);
    if (t0 >= t1) return;

    if (tid == 0) { log_cnt[0] = 0; log_cnt[1] = 0; log_cnt[2] = 0; log_cnt[3] = 0; }
    __syncthreads();

    const uint32_t KT = a.ld_bytes >> 7;
    const uint32_t rel_base = a.tile_first * kBM;
    const uint32_t st_row = lane >> 3;
    const uint32_t st_lane_off = st_row * a.ld_bytes + (((lane & 7) ^ st_row) << 4);
    const uint32_t fr = lane & 15, fg = lane >> 4, r7 = fr & 7;
    const uint32_t a_frag0 = ((wr * 16 + (fr >> 3)) << 10) + (r7 << 7);
    const uint32_t b_frag0 = 32768u + ((wc * 8 + (fr >> 3)) << 10) + (r7 << 7);
    const uint32_t c_off0 = ((0 * 4 + fg) ^ r7) << 4, c_off1 = ((1 * 4 + fg) ^ r7) << 4;

    // staging pieces of this wave inside each unit (16 pieces per unit, 2 per wave):
    //   A_m0 = pieces {0-7,16-23}, A_m1 = +8;  B_n0 = pieces {0-3,8-11,16-19,24-27}, B_n1 = +4
    uint32_t pa[2], pb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t idx = wave * 2 + i;
        pa[i] = (idx & 7) + (idx >> 3) * 16;
        pb[i] = (idx & 3) + (idx >> 2) * 8;
    }

    for (uint32_t qb = qb0; qb < a.nqb; qb += qb_step) {
        float thr[4], qn2[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const uint32_t gq = qb * kBN + wc * 64 + n * 16 + fr;
            thr[n] = a.thr[gq];
            qn2[n] = METRIC == M_L2 ? a.qnorm2[gq] : 0.0f;
        }
        const char* q_base = a.queries + (uint64_t)qb * kBN * a.ld_bytes + st_lane_off;
        const char* c_base = a.corpus + st_lane_off;
        f32x4 acc[8][4];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        const uint32_t total_it = (t1 - t0) * KT;

        // ---- prologue: the whole K-tile 0 into buffer 0, fully landed, groups not yet staggered
        {
            const char* a_src = c_base + (uint64_t)t0 * kBM * a.ld_bytes;
            const char* b_src = q_base;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                VROD_GLDS16(a_src + (uint64_t)pa[i] * 8 * a.ld_bytes, lds + pa[i] * 1024);
                VROD_GLDS16(a_src + (uint64_t)(pa[i] + 8) * 8 * a.ld_bytes, lds + (pa[i] + 8) * 1024);
                VROD_GLDS16(b_src + (uint64_t)pb[i] * 8 * a.ld_bytes, lds + 32768 + pb[i] * 1024);
                VROD_GLDS16(b_src + (uint64_t)(pb[i] + 4) * 8 * a.ld_bytes, lds + 32768 + (pb[i] + 4) * 1024);
            }
        }
        __syncthreads();                       // vmcnt(0) + barrier
        if (wr == 1) VROD_BARRIER();   // group 1 now runs one barrier behind

        // fragment registers: A half (4 m-tiles x 2 k-halves), B half (2 n-tiles x 2 k-halves)
        typedef typename std::conditional<sizeof(T) == 2, bf16x8, f32x4>::type frag_t;
        frag_t af[4][2], bf[2][2];
        frag_t bf0[2][2];   // the nh = 0 query fragments stay live from phase 0 to phase 3 (no LDS re-read)
        bool pace_on = true;   // (thread 0) false after one pacing timeout

        for (uint32_t it = 0; it < total_it; ++it) {
            const uint32_t buf = it & 1;
            const char* l = lds + buf * kStageBytes;
            char* lnext = lds + (buf ^ 1) * kStageBytes;
            // source of the NEXT K-tile (clamped at the end: the redundant loads keep the
            // vmcnt bookkeeping uniform and are never read)
            const uint32_t nx = it + 1 < total_it ? it + 1 : it;
            const uint32_t ntile = t0 + nx / KT, nkt = nx % KT;
            const char* a_src = c_base + (uint64_t)ntile * kBM * a.ld_bytes + (uint64_t)nkt * 128;
            const char* b_src = q_base + (uint64_t)nkt * 128;
            const uint32_t kt = it % KT;

            // the log is flushed with both groups re-aligned, one phase after a tile's first
            // (see below); `due` is read where no append can be in flight in either group
            bool flush_now = false;

#define VROD_LOAD_A(MH)                                                                                   \
    _Pragma("unroll") for (int mm = 0; mm < 4; ++mm) {                                                   \
        af[mm][0] = *reinterpret_cast<const frag_t*>(l + a_frag0 + ((MH) * 4 + mm) * 2048 + c_off0);    \
        af[mm][1] = *reinterpret_cast<const frag_t*>(l + a_frag0 + ((MH) * 4 + mm) * 2048 + c_off1);    \
    }
#define VROD_LOAD_B(BF, NH)                                                                               \
    _Pragma("unroll") for (int nn = 0; nn < 2; ++nn) {                                                   \
        BF[nn][0] = *reinterpret_cast<const frag_t*>(l + b_frag0 + ((NH) * 2 + nn) * 2048 + c_off0);    \
        BF[nn][1] = *reinterpret_cast<const frag_t*>(l + b_frag0 + ((NH) * 2 + nn) * 2048 + c_off1);    \
    }
#define VROD_STAGE_A1(OFF, I)                                                                             \
    VROD_GLDS16(a_src + (uint64_t)(pa[I] + (OFF)) * 8 * a.ld_bytes, lnext + (pa[I] + (OFF)) * 1024);
#define VROD_STAGE_B1(OFF, I)                                                                             \
    VROD_GLDS16(b_src + (uint64_t)(pb[I] + (OFF)) * 8 * a.ld_bytes, lnext + 32768 + (pb[I] + (OFF)) * 1024);
// pieces issued in the load segment / inside the MFMA cluster
#define VROD_STAGE_L(KIND, OFF)                                                                           \
    if constexpr (GP <= 1) { VROD_STAGE_##KIND##1(OFF, 0) }                                               \
    if constexpr (GP == 0) { VROD_STAGE_##KIND##1(OFF, 1) }
#define VROD_MFMA_ONE(BF, MH, NH, KK, MM, NN)                                                             \
    if constexpr (sizeof(T) == 2) {                                                                       \
        acc[(MH) * 4 + MM][(NH) * 2 + NN] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                      \
            af[MM][KK], BF[NN][KK], acc[(MH) * 4 + MM][(NH) * 2 + NN], 0, 0, 0);                          \
    } else {                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                     \
            acc[(MH) * 4 + MM][(NH) * 2 + NN] = __builtin_amdgcn_mfma_f32_16x16x4f32(                     \
                af[MM][KK][i], BF[NN][KK][i], acc[(MH) * 4 + MM][(NH) * 2 + NN], 0, 0, 0);                \
    }
#define VROD_MFMA_ROW(BF, MH, NH, KK, MM) VROD_MFMA_ONE(BF, MH, NH, KK, MM, 0) VROD_MFMA_ONE(BF, MH, NH, KK, MM, 1)
// 16 MFMAs; DMA pieces dropped in after the 4th and the 10th when GP says so
#define VROD_COMPUTE(BF, MH, NH, KIND, OFF)                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                        \
    VROD_MFMA_ROW(BF, MH, NH, 0, 0) VROD_MFMA_ROW(BF, MH, NH, 0, 1)                                       \
    if constexpr (GP == 2) { VROD_STAGE_##KIND##1(OFF, 0) }                                               \
    VROD_MFMA_ROW(BF, MH, NH, 0, 2) VROD_MFMA_ROW(BF, MH, NH, 0, 3) VROD_MFMA_ROW(BF, MH, NH, 1, 0)       \
    if constexpr (GP >= 1) { VROD_STAGE_##KIND##1(OFF, 1) }                                               \
    VROD_MFMA_ROW(BF, MH, NH, 1, 1) VROD_MFMA_ROW(BF, MH, NH, 1, 2) VROD_MFMA_ROW(BF, MH, NH, 1, 3)       \
    __builtin_amdgcn_s_setprio(0);
#define VROD_PHASE_SYNC()                                                                                 \
    if constexpr (GP == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                               \
    else if constexpr (GP == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                          \
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                                 \
    VROD_BARRIER();

            // ---------------- pacing: the nqb work-groups that walk the same strip (one per query
            // block, same XCD) must stay within about one tile of each other or the corpus tile
            // they share falls out of the XCD's L2 and is fetched from HBM once per work-group.
            // Nothing but speed depends on it: relaxed agent-scope counter, bounded spin.
            // (<= ~40 us per wait; a sibling that never arrives -- not resident beside a co-tenant kernel, or
            // under a counter mode that serialises dispatch -- costs ONE timeout, then this work-group stops pacing)
            if (a.pace_every && pace_on && kt == 0 && it > 0 && tid == 0) {
                const uint32_t tix = it / KT;
                if (tix % a.pace_every == 0) {
                    uint32_t* ctr = a.pace + strip;
                    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t want = a.nqb * (tix / a.pace_every);
                    bool ok = false;
                    for (uint32_t spin = 0; spin < 64u; ++spin) {
                        if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) { ok = true; break; }
                        __builtin_amdgcn_s_sleep(8);
                    }
                    pace_on = ok;
                }
            }
            // ---------------- phase 0: quadrant (0,0), stages A_m0 of the next K-tile
            VROD_LOAD_A(0)
            VROD_LOAD_B(bf0, 0)
            VROD_STAGE_L(A, 0)
            VROD_PHASE_SYNC()
            VROD_COMPUTE(bf0, 0, 0, A, 0)
            VROD_BARRIER();

            // ---------------- phase 1: quadrant (0,1), stages B_n0
            if (kt == 0 && it > 0) flush_now = log_cnt[3] != 0u;  // previous tile's appends are all done
            VROD_LOAD_B(bf, 1)
            VROD_STAGE_L(B, 0)
            VROD_PHASE_SYNC()
            VROD_COMPUTE(bf, 0, 1, B, 0)
            VROD_BARRIER();

            // ---------------- phase 2: quadrant (1,1), stages B_n1
            VROD_LOAD_A(1)
            VROD_STAGE_L(B, 4)
            VROD_PHASE_SYNC()
            VROD_COMPUTE(bf, 1, 1, B, 4)
            VROD_BARRIER();

            // ---------------- phase 3: quadrant (1,0), stages A_m1 (B_n0 fragments still in registers)
            VROD_STAGE_L(A, 8)
            VROD_PHASE_SYNC()
            VROD_COMPUTE(bf0, 1, 0, A, 8)
            VROD_BARRIER();
            // A finished corpus tile is filtered AFTER this barrier, i.e. in this group's load slot,
            // so the other group's MFMA segment runs meanwhile (inside the compute segment it would
            // stall both groups).  The accumulators are not touched again before the next compute.
            if (kt == KT - 1) {
                const uint32_t tile = t0 + it / KT;
                if constexpr (DENSE)
                    dense_store_tile<METRIC>(a, acc, qn2, tile * kBM + wr * 128 + fg * 4, qb * kBN + wc * 64 + fr);
                else
                    filter_tile<METRIC>(a, acc, thr, qn2, tile * kBM + wr * 128 + fg * 4, wc * 64 + fr, qb, rel_base, log, log_cnt);
            }

            if (flush_now) {
                // re-align the groups (group 0 waits one barrier), flush, stagger again
                if (wr == 0) VROD_BARRIER();
                flush_log(a, log, log_cnt, qb, rel_base, tid);
                if (tid == 0) log_cnt[3] = 0u;
                __syncthreads();
                if (wr == 1) VROD_BARRIER();
            }
        }
#undef VROD_LOAD_A
#undef VROD_LOAD_B
#undef VROD_STAGE_A1
#undef VROD_STAGE_B1
#undef VROD_STAGE_L
#undef VROD_MFMA_ONE
#undef VROD_MFMA_ROW
#undef VROD_COMPUTE
#undef VROD_PHASE_SYNC
        if (wr == 0) VROD_BARRIER();   // group 0 waits for group 1's last phase
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        flush_log(a, log, log_cnt, qb, rel_base, tid);
        if (tid == 0) log_cnt[3] = 0u;
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------
// Schedule 3 (bf16 default): 4 waves, one per SIMD, 128 rows x 128 queries per wave.
//
// Why: with 8 waves every K-tile moves 192 KB of fragment reads through the LDS and needs 8
// barriers; with 4 waves it is 128 KB and 2 barriers, and the per-lane filter state is 8
// thresholds instead of 4 per wave but half as many waves.  One wave per SIMD owns the whole
// 512-entry register file: the 256 accumulator registers are a[0:255], named literally by the
// inline-asm MFMAs (hipcc's own allocation of a 256-register accumulator array spills; the
// clobber list below makes the kernel descriptor allocate the AGPRs and keeps the compiler out
// of them -- audited in the build: no compiler v_accvgpr_* and no scratch, scripts/audit_w4.py).
// Tile (m, n) of the wave's 8 x 8 grid of 16x16 tiles is a[(m*8+n)*4 .. +3].
//
// Per K-tile, 4 phases over the accumulator quadrants (mh, nh) = (0,0) (0,1) (1,1) (1,0), 32 MFMAs
// each, with between them the 8 fragment reads the NEXT phase needs and 4 LDS-DMA pieces:
//   q0: MFMA(A0,B0) | read B1 of this K-tile        | stage A_m0 of K-tile it+2
//   q1: MFMA(A0,B1) | read A1                       | stage B_n0 of K-tile it+2
//   -- wait vmcnt(16) (A_m0, B_n0 of K-tile it+1 have landed), barrier M
//   q2: MFMA(A1,B1) | read A0 of K-tile it+1        | stage B_n1 of K-tile it+2
//   q3: MFMA(A1,B0) | read B0 of K-tile it+1        | stage A_m1 of K-tile it+2
//   -- tile epilogue (filter / dense store) when the K-tile was the tile's last
//   -- wait vmcnt(16) (B_n1, A_m1 of K-tile it+1 have landed), barrier E
// K-tile it+2 goes into the buffer of K-tile it: a unit is restaged only after the barrier that
// follows its last read (A_m0, B_n0: read in q2, q3 of iteration it-1, barrier E(it-1); B_n1,
// A_m1: read in q0, q1, barrier M(it)), every ds_read is retired (lgkmcnt(0)) before the barrier,
// and a unit is read only after every wave's counted wait for it plus a barrier.  16 to 32 DMA
// pieces per wave stay in flight: a piece has more than one K-tile of MFMAs to land.
// The first K-tile of a corpus tile uses the C = 0 form of the MFMA, so the accumulators are
// never cleared; the epilogue reads them with v_accvgpr_read.
// ---------------------------------------------------------------------------------------------
#ifdef VROD_W4_PROF
// diagnostic build only (scripts/build_variant.sh prof -DVROD_W4_PROF): shader-clock totals of the filtered 4-wave kernel
// [0] wave cycles in the kernel  [1] tile epilogues  [2] of which the walk of columns with a hit  [3] wait at the barrier
// that follows an epilogue  [4] epilogues  [5] epilogues that walked  [6] columns walked  [7] appends  [8] flushes [9] flush cycles
// [10] counted wait + barrier M  [11] counted wait + barrier E (K-tiles without an epilogue)  [12] K-tiles  [13] phases q0 q1  [14] phases q2 q3
// (kept in wave-uniform registers while the kernel runs, added to the totals once at its end)
__device__ unsigned long long g_w4_prof[16];
__device__ __forceinline__ uint32_t w4_clock() { return (uint32_t)__builtin_readcyclecounter(); }
#define W4_PROF(...) __VA_ARGS__
#else
#define W4_PROF(...)
#endif
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// The accumulator file is owned by these statements: tile (m, n) is a[(m*8+n)*4 .. +3], named
// literally.  (Letting hipcc allocate the 256 accumulator registers -- a plain f32x4 array with
// the MFMA builtin, or "a"-constrained asm operands -- ends in hundreds of spills.)  This is sound
// only while the compiler keeps out of the AGPRs wherever the accumulators are live, which
// scripts/audit_w4.py checks on the emitted assembly (tests/test_build_audit.py runs it).
template <int BASE, bool ZERO>
__device__ __forceinline__ void w4_mfma1(const bf16x8& x, const bf16x8& y) {
    if constexpr (ZERO)
        asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, 0" ::"v"(x), "v"(y), "i"(BASE), "i"(BASE + 3) : "memory");
    else
        asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(x), "v"(y), "i"(BASE), "i"(BASE + 3) : "memory");
}
// group G (0..7) of a quadrant's 32 MFMAs: index I = kk*16 + mm*4 + nn
template <int MH, int NH, bool ZERO, int G>
__device__ __forceinline__ void w4_mfma_group(const bf16x8 (&FA)[4][2], const bf16x8 (&FB)[4][2]) {
    static_for<0, 4>([&](auto ic) {
        constexpr int I = G * 4 + decltype(ic)::value;
        constexpr int kk = I / 16, mm = (I / 4) % 4, nn = I % 4;
        w4_mfma1<((MH * 4 + mm) * 8 + NH * 4 + nn) * 4, ZERO && kk == 0>(FA[mm][kk], FB[nn][kk]);
    });
}
template <int BASE>
__device__ __forceinline__ f32x4 w4_read_acc() {
    f32x4 v;
    asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c5]\n\tv_accvgpr_read_b32 %2, a[%c6]\n\tv_accvgpr_read_b32 %3, a[%c7]"
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]) : "i"(BASE), "i"(BASE + 1), "i"(BASE + 2), "i"(BASE + 3));
    return v;
}

// the same with a wave-uniform run-time tile index t = m * 8 + n (0..63): one computed jump into a table of
// 36-byte cases (4 reads of 8 B + s_branch).  (As a C++ switch hipcc emits a chain of ~25 scalar branches.)
#ifndef W4_RD_STRIDE_S
#define W4_RD_STRIDE_S "36"   // bytes per case; tests/test_gpu_walk.py must fail on a build with any other value
#endif
__device__ __forceinline__ f32x4 w4_read_acc_dyn(uint32_t t) {
    f32x4 v;
    uint32_t tmp;
#define W4_RD(T)                                                                                    \
    "v_accvgpr_read_b32 %0, a[4*" #T "]\n\tv_accvgpr_read_b32 %1, a[4*" #T "+1]\n\t"                \
    "v_accvgpr_read_b32 %2, a[4*" #T "+2]\n\tv_accvgpr_read_b32 %3, a[4*" #T "+3]\n\ts_branch .Lw4rd_e_%=\n\t"
#define W4_RD8(A, B, C, D, E, F, G, H) W4_RD(A) W4_RD(B) W4_RD(C) W4_RD(D) W4_RD(E) W4_RD(F) W4_RD(G) W4_RD(H)
    asm volatile("s_getpc_b64 vcc\n"
                 ".Lw4rd_a_%=:\n\t"
                 "s_mul_i32 %4, %5, " W4_RD_STRIDE_S "\n\t"
                 "s_add_u32 vcc_lo, vcc_lo, %4\n\t"
                 "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
                 "s_add_u32 vcc_lo, vcc_lo, .Lw4rd_t_%=-.Lw4rd_a_%=\n\t"
                 "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
                 "s_setpc_b64 vcc\n"
                 ".Lw4rd_t_%=:\n\t"
                 W4_RD8(0, 1, 2, 3, 4, 5, 6, 7) W4_RD8(8, 9, 10, 11, 12, 13, 14, 15)
                 W4_RD8(16, 17, 18, 19, 20, 21, 22, 23) W4_RD8(24, 25, 26, 27, 28, 29, 30, 31)
                 W4_RD8(32, 33, 34, 35, 36, 37, 38, 39) W4_RD8(40, 41, 42, 43, 44, 45, 46, 47)
                 W4_RD8(48, 49, 50, 51, 52, 53, 54, 55) W4_RD8(56, 57, 58, 59, 60, 61, 62, 63)
                 ".Lw4rd_e_%=:"
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=&s"(tmp)
                 : "s"(t)
                 : "vcc", "scc");
#undef W4_RD8
#undef W4_RD
    return v;
}

// Drain the four wave-private log segments (counts in log_cnt[4..7]) into the per-query lists.
// Called by ALL threads at the same program point, after a barrier that follows every wave's
// appends; the caller's waves reset their register counters.
__device__ __forceinline__ void flush_log_w4(const MfmaKernelArgs& a, const uint2* log, uint32_t* log_cnt, uint32_t qb,
                                             uint32_t rel_base, int tid) {
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t n = log_cnt[4 + w] < (uint32_t)kLogCapW4 ? log_cnt[4 + w] : (uint32_t)kLogCapW4;
        for (uint32_t i = tid; i < n; i += 256) {
            const uint2 e = log[w * kLogCapW4 + i];
            global_append(a, qb * kBN + (e.y >> 24), e.x, rel_base + (e.y & 0xFFFFFFu));
        }
    }
    __syncthreads();
    if (tid == 0) { lds_zero3(log_cnt + 2); lds_zero3(log_cnt + 5); }   // [2..3] flush-due flags, [4..7] wave counts
    __syncthreads();
}

// The fused filter of the 4-wave kernel: the wave's 128 x 128 scores (in a[0:255]) against the 8
// per-lane thresholds.  row_w = first row of the lane's 4-row group in tile m = 0.
template <int METRIC>
__device__ __forceinline__ void w4_filter_tile(const MfmaKernelArgs& a, const float* thr_l, const float* qn2_l, const float* xn_l,
                                               uint32_t row_w, uint32_t ql0, uint32_t qb, uint32_t rel_base,
                                               uint2* log /* this wave's segment */, uint32_t* log_cnt, int wave, uint32_t& wlog
                                               W4_PROF(, uint32_t (&pc)[16])) {
    const uint32_t wlog_in = wlog;
    // thr_l / qn2_l: the work-group's per-query values in LDS; xn_l: this lane's 4-row group of the
    // tile's row norms in LDS (m = 0), 16 floats apart per m
    float thr[8], qn2[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        thr[n] = thr_l[ql0 + n * 16];
        qn2[n] = METRIC == M_L2 ? qn2_l[ql0 + n * 16] : 0.0f;
    }
    const uint32_t lds_log_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)log;
    // the last MFMAs are still in the pipe: an accumulator may be read 4 passes + 2 states later
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    float best[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) best[n] = worst_score(METRIC);
    static_for<0, 8>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        f32x4 xv = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (METRIC == M_L2) xv = *reinterpret_cast<const f32x4*>(xn_l + m * 16);
        static_for<0, 8>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
            const f32x4 v = w4_read_acc<(m * 8 + n) * 4>();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sc = METRIC == M_COSINE ? v[r] : __builtin_fmaf(-2.0f, v[r], xv[r] + qn2[n]);
                best[n] = METRIC == M_COSINE ? __builtin_fmaxf(best[n], sc) : __builtin_fminf(best[n], sc);
            }
        });
    });
    // columns (n) that hold a hit, as a wave-uniform bit mask
    uint32_t colmask = 0u;
#pragma unroll
    for (int n = 0; n < 8; ++n) colmask |= __any(better<METRIC>(best[n], thr[n])) ? 1u << n : 0u;
    W4_PROF(const uint32_t pt0 = w4_clock(); pc[4] += 1; if (colmask) pc[5] += 1;)
    if (colmask) {
        // Which 16 x 16 tiles (m, n) of such a column hold one: straight-line code, 9 instructions per
        // tile, bit n * 8 + m of a wave-uniform mask.  The hits themselves are then appended by ONE copy
        // of code in a run-time loop over the marked tiles, the accumulator registers picked by a
        // computed jump.  (Unrolled over the 64 tiles the appends were ~100 KB of code per kernel, a
        // column's 10 KB executed once in a while and fetched from L2 every time: 2000 cycles per
        // column with the other three waves waiting at the barrier, profiles/r02/q_w4_cycle_profile.txt.)
        uint64_t tmask = 0ull;
        static_for<0, 8>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
            if (!(colmask & (1u << n))) return;
            W4_PROF(pc[6] += 1;)
            static_for<0, 8>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                f32x4 xv = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (METRIC == M_L2) xv = *reinterpret_cast<const f32x4*>(xn_l + m * 16);
                const f32x4 v = w4_read_acc<(m * 8 + n) * 4>();
                float sc[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[r] = METRIC == M_COSINE ? v[r] : __builtin_fmaf(-2.0f, v[r], xv[r] + qn2[n]);
                const float tb = METRIC == M_COSINE ? __builtin_fmaxf(__builtin_fmaxf(sc[0], sc[1]), __builtin_fmaxf(sc[2], sc[3]))
                                                    : __builtin_fminf(__builtin_fminf(sc[0], sc[1]), __builtin_fminf(sc[2], sc[3]));
                tmask |= __any(better<METRIC>(tb, thr[n])) ? 1ull << (n * 8 + m) : 0ull;
            });
        });
#pragma unroll 1
        while (tmask) {
            const uint32_t t = (uint32_t)__builtin_ctzll(tmask);
            tmask &= tmask - 1ull;
            const uint32_t n = t >> 3, m = t & 7u;
            const uint32_t ql = ql0 + n * 16;
            const float thr_n = thr_l[ql];
            f32x4 xv = f32x4{0.f, 0.f, 0.f, 0.f};
            float qn2_n = 0.0f;
            if constexpr (METRIC == M_L2) { xv = *reinterpret_cast<const f32x4*>(xn_l + m * 16); qn2_n = qn2_l[ql]; }
            const f32x4 v = w4_read_acc_dyn(m * 8u + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sc = METRIC == M_COSINE ? v[r] : __builtin_fmaf(-2.0f, v[r], xv[r] + qn2_n);
                const uint32_t row = row_w + m * 16 + r;
                const bool hitr = better<METRIC>(sc, thr_n) && row >= a.row_lo && row < a.row_end;
                // the wave owns a quarter of the log and counts its entries in a register: a
                // ballot and a lane prefix give every hit its slot -- no LDS atomic, no wait.
                // (The write stays in asm: as a compiler-visible LDS store it would be preceded
                // by s_waitcnt vmcnt(0), see filter_tile.)
                const unsigned long long hm = __ballot(hitr);
                if (hm == 0ull) continue;
                const uint32_t pos = wlog + __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                if (hitr) {
                    if (pos < (uint32_t)kLogCapW4) {
                        const uint64_t e = ((uint64_t)((ql << 24) | (row - rel_base)) << 32) | __float_as_uint(sc);
                        asm volatile("ds_write_b64 %0, %1" :: "v"(lds_log_addr + pos * 8u), "v"(e) : "memory");
                    } else {
                        global_append(a, qb * kBN + ql, __float_as_uint(sc), row);
                    }
                }
                wlog += (uint32_t)__builtin_popcountll(hm);
            }
        }
    }
    W4_PROF(pc[2] += w4_clock() - pt0; pc[7] += wlog - wlog_in;)
    // publish the wave's count for the flush; past half of the segment: ask for one
    if (wlog != wlog_in) {
        const uint32_t cnt_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)log_cnt;
        asm volatile("ds_write_b32 %0, %1" :: "v"(cnt_addr + 16u + 4u * (uint32_t)wave), "v"(wlog) : "memory");
        if (wlog >= (uint32_t)(kLogCapW4 / 2)) asm volatile("ds_write_b32 %0, %1" :: "v"(cnt_addr + 12u), "v"(1u) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int METRIC>
__device__ __forceinline__ void w4_dense_store_tile(const MfmaKernelArgs& a, const float* qn2_l, uint32_t ql0, uint32_t row_w, uint32_t gq0) {
    float qn2[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) qn2[n] = METRIC == M_L2 ? qn2_l[ql0 + n * 16] : 0.0f;
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    static_for<0, 8>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const uint32_t row = row_w + m * 16;
        f32x4 xv = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (METRIC == M_L2) xv = *reinterpret_cast<const f32x4*>(a.xnorm2 + row);
        const bool in = row - a.row_lo < a.dense_ld;
        static_for<0, 8>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
            const f32x4 v = w4_read_acc<(m * 8 + n) * 4>();
            f32x4 sc;
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[r] = METRIC == M_COSINE ? v[r] : __builtin_fmaf(-2.0f, v[r], xv[r] + qn2[n]);
            if (in) *reinterpret_cast<f32x4*>(a.dense_out + (uint64_t)(gq0 + n * 16) * a.dense_ld + (row - a.row_lo)) = sc;
        });
    });
}

// The sample pass only feeds a threshold (the j-th best score of the sample rows, vrod_index.hip): the j-th best of the
// per-group BESTS is a valid stand-in (at least j rows are that good; it is the exact value unless two of the j best rows
// share a group) and costs 1/32 of the writes and of the select's reads.  A lane's 32 scores per query column are one
// group: rows row_w + 16 m + r of the tile, m < 8, r < 4; eight groups per 256-row tile ((wave row, lane >> 4)).
// The launch covers whole tiles of real rows only (row_end a multiple of 256: the caller's condition for this form) --
// a per-row mask here costs 32 lane masks in SGPRs, which no longer fit beside the main loop's.
template <int METRIC>
__device__ __forceinline__ void w4_groupmax_store_tile(const MfmaKernelArgs& a, const float* qn2_l, const float* xn_l, uint32_t ql0,
                                                       uint32_t gq0, uint32_t group) {
    // (opaque per tile: hipcc otherwise hoists the store addresses out of the scan loop, which does not fit beside the
    //  fragments -- it then parks values in the accumulator file, scripts/audit_w4.py)
    asm volatile("" : "+v"(gq0), "+v"(group));
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    const uint32_t off0 = gq0 * a.dense_ld + group, step = 16u * a.dense_ld;   // nq_pad * dense_ld < 2^32 (launcher)
    // one query column at a time, its best stored at once: one running best and four scores live (walking the row tiles
    // outermost, as the filter does, hipcc kept all 256 scores in flight here and spilled the fragments)
    static_for<0, 8>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        const float qn2 = METRIC == M_L2 ? qn2_l[ql0 + n * 16] : 0.0f;
        float best = worst_score(METRIC);
        static_for<0, 8>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            f32x4 xv = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (METRIC == M_L2) xv = *reinterpret_cast<const f32x4*>(xn_l + m * 16);
            const f32x4 v = w4_read_acc<(m * 8 + n) * 4>();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sc = METRIC == M_COSINE ? v[r] : __builtin_fmaf(-2.0f, v[r], xv[r] + qn2);
                best = METRIC == M_COSINE ? __builtin_fmaxf(best, sc) : __builtin_fminf(best, sc);
            }
        });
        if (group < a.dense_ld) a.dense_out[off0 + (uint32_t)n * step] = best;
        __builtin_amdgcn_sched_barrier(0);
    });
}

// DENSE: 0 = filtered launch, 1 = sample pass writing every score, 2 = sample pass writing group bests (one kernel per
// form: with both sample epilogues in one function hipcc ran out of VGPRs and went into the accumulator file).
template <int METRIC, int DENSE, bool SPLIT>
__global__ __launch_bounds__(256) void scan_mfma_w4_kernel(const MfmaKernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // makes the kernel descriptor allocate a[0:255]
    asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");
    uint32_t* log_cnt = reinterpret_cast<uint32_t*>(lds + kLdsCtl);  // [0] count [3] flush due
    uint2* log = reinterpret_cast<uint2*>(lds + kLdsLog);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    uint32_t strip, qb0, qb_step;
    if (!wg_assignment(a, strip, qb0, qb_step)) return;
    const uint32_t t0 = a.tile_first + (uint32_t)((uint64_t)a.ntiles * strip / a.nstrips);
    const uint32_t t1 = a.tile_first + (uint32_t)((uint64_t)a.ntiles * (strip + 1) / a.nstrips);
    if (t0 >= t1) return;

    if (tid == 0) { lds_zero3(log_cnt); lds_zero3(log_cnt + 3); lds_zero3(log_cnt + 5); }
    __syncthreads();

    const uint32_t KT = a.ld_bytes >> 7;
    const uint32_t rel_base = a.tile_first * kBM;
    const uint32_t st_row = lane >> 3;
    const uint32_t lda_bytes = SPLIT ? a.lda_bytes : a.ld_bytes;
    const uint32_t st_lane_off_a = st_row * lda_bytes + (((lane & 7) ^ st_row) << 4);
    const uint32_t st_lane_off_b = st_row * a.ld_bytes + (((lane & 7) ^ st_row) << 4);
    const uint32_t fr = lane & 15, fg = lane >> 4, r7 = fr & 7;
    const uint32_t a_frag0 = ((wr * 16 + (fr >> 3)) << 10) + (r7 << 7);
    const uint32_t b_frag0 = 32768u + ((wc * 16 + (fr >> 3)) << 10) + (r7 << 7);
    const uint32_t c_off0 = ((0 * 4 + fg) ^ r7) << 4, c_off1 = ((1 * 4 + fg) ^ r7) << 4;
    const uint32_t piece_stride_a = 8u * lda_bytes, piece_stride_b = 8u * a.ld_bytes;   // 8 rows; 31 pieces fit 32 bits
    const uint32_t total_it = (t1 - t0) * KT;

    // ONE query block per work-group: a loop over query blocks here would re-enter the prologue with the
    // accumulator file live in hipcc's eyes (it parks kernel-entry values in AGPRs up to the first MFMA);
    // the launcher splits batches of more than `slots` query blocks into several launches instead
    const uint32_t qb = a.qb_base + qb0;
    {
        float* thr_l = reinterpret_cast<float*>(lds + kLdsThr);
        float* qn2_l = reinterpret_cast<float*>(lds + kLdsQn2);
        const float* xn_l = reinterpret_cast<const float*>(lds + kLdsXn2);
        thr_l[tid] = DENSE ? 0.0f : a.thr[qb * kBN + tid];
        qn2_l[tid] = METRIC == M_L2 ? a.qnorm2[qb * kBN + tid] : 0.0f;
        // (published by the prologue's __syncthreads)
        // per-lane source pointers of the K-tile being staged (two K-tiles ahead of the MFMAs)
        const char* ua_src = a.corpus + st_lane_off_a + (uint64_t)t0 * kBM * lda_bytes;
        const char* ub_src = a.queries + st_lane_off_b + (uint64_t)qb * kBN * a.ld_bytes;
        uint32_t st_kt = 0, st_tile = t0, a_kt = 0;   // a_kt: K-tile of the corpus row ua_src points at
        // unit A_mh / B_nh = the 16 pieces (8 rows x 128 B each) of rows [h*64, h*64+64) of both
        // 128-row halves; this wave moves 4 of them: idx = wave*4 + i -> piece (idx>>3)*16 + (idx&7) + h*8
        auto stage_a = [&](uint32_t buf, int h, int i) {
            const uint32_t idx = wave * 4 + i;
            const uint32_t p = (idx >> 3) * 16 + (idx & 7) + h * 8;
            VROD_GLDS16(ua_src + p * piece_stride_a, lds + (buf & 1) * kStageBytes + p * 1024);
        };
        auto stage_b = [&](uint32_t buf, int h, int i) {
            const uint32_t idx = wave * 4 + i;
            const uint32_t p = (idx >> 3) * 16 + (idx & 7) + h * 8;
            VROD_GLDS16(ub_src + p * piece_stride_b, lds + (buf & 1) * kStageBytes + 32768 + p * 1024);
        };
        // next K-tile of the strip (clamped at its end: the last K-tile is re-staged, never read)
        auto stage_advance = [&]() {
            // one update site per pointer (uniform deltas picked by selects): written as branches
            // with in-place updates, hipcc moves the two pointers into a scratch array
            const bool in_tile = st_kt + 1 < KT;
            const bool next_tile = !in_tile && st_tile + 1 < t1;
            // SPLIT: K-tiles 3j, 3j+1, 3j+2 read corpus K-tile 2j, 2j, 2j+1 ([hi_j | lo_j] interleaved: the
            // hi plane is staged twice in a row, the second time from L2); a_kt = corpus K-tile
            const bool hold = SPLIT && (st_kt % 3u) == 0u;
            const int64_t da_in = hold ? 0 : 128;
            const int64_t a_back = SPLIT ? (int64_t)a_kt * 128 : (int64_t)(KT - 1) * 128;
            const int64_t da = in_tile ? da_in : next_tile ? (int64_t)kBM * lda_bytes - a_back : 0;
            const int64_t db = in_tile ? 128 : next_tile ? -(int64_t)(KT - 1) * 128 : 0;
            if constexpr (SPLIT) a_kt = in_tile ? (hold ? a_kt : a_kt + 1) : next_tile ? 0u : a_kt;
            st_kt = in_tile ? st_kt + 1 : next_tile ? 0u : st_kt;
            st_tile += next_tile ? 1u : 0u;
            ua_src += da;
            ub_src += db;
        };

        // ---- prologue: K-tiles 0 and 1 whole, landed; fragments A0, B0 of K-tile 0
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i) { stage_a(b, h, i); stage_b(b, h, i); }
            stage_advance();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        bf16x8 FA0[4][2], FA1[4][2], FBx[4][2], FBy[4][2];
#define W4_LOAD_A1(FA, MH, L, J) FA[(J) >> 1][(J) & 1] = *reinterpret_cast<const bf16x8*>((L) + a_frag0 + ((MH) * 4 + ((J) >> 1)) * 2048 + (((J) & 1) ? c_off1 : c_off0));
#define W4_LOAD_B1(FB, NH, L, J) FB[(J) >> 1][(J) & 1] = *reinterpret_cast<const bf16x8*>((L) + b_frag0 + ((NH) * 4 + ((J) >> 1)) * 2048 + (((J) & 1) ? c_off1 : c_off0));
// (Where in a phase the four DMA pieces go does not matter: all behind the fragment reads +0.2 %, all in front of
// them -1.1 %, alternating 0.0 % -- profiles/r02/mfma_experiments.md section 8.)
#define W4_PHASE_S(FA, FB, MH, NH, ZERO, LD, DM)                                                   \
    w4_mfma_group<MH, NH, ZERO, 0>(FA, FB); LD(0) LD(1) DM(0)                                      \
    w4_mfma_group<MH, NH, ZERO, 1>(FA, FB); LD(2) LD(3)                                            \
    w4_mfma_group<MH, NH, ZERO, 2>(FA, FB); LD(4) LD(5) DM(1)                                      \
    w4_mfma_group<MH, NH, ZERO, 3>(FA, FB); LD(6) LD(7)                                            \
    w4_mfma_group<MH, NH, ZERO, 4>(FA, FB); DM(2)                                                  \
    w4_mfma_group<MH, NH, ZERO, 5>(FA, FB);                                                        \
    w4_mfma_group<MH, NH, ZERO, 6>(FA, FB); DM(3)                                                  \
    w4_mfma_group<MH, NH, ZERO, 7>(FA, FB);
#define W4_PHASE(FA, FB, MH, NH, LD, DM)                                                           \
    if (first) { W4_PHASE_S(FA, FB, MH, NH, true, LD, DM) } else { W4_PHASE_S(FA, FB, MH, NH, false, LD, DM) }
#define W4_DMU0(j) stage_a(it & 1, 0, j);
#define W4_DMU1(j) stage_b(it & 1, 0, j);
#define W4_DMU2(j) stage_b(it & 1, 1, j);
#define W4_DMU3(j) stage_a(it & 1, 1, j);
#define W4_VMWAIT "s_waitcnt vmcnt(16) lgkmcnt(0)"
#define W4_ITER(BX, BY, LDQ0, LDQ3)                                                                \
    {                                                                                              \
        const char* l = lds + (it & 1) * kStageBytes;                                              \
        const char* ln = lds + ((it + 1) & 1) * kStageBytes;                                       \
        const bool first = kt == 0;                                                                \
        /* pacing of the sibling work-groups (see the phased kernel), here in units of K-tiles: with  \
           the staging two K-tiles ahead no work-group ever waits for HBM, so nothing else keeps   \
           the siblings of a strip together */                                                     \
        if (a.pace_every && pace_on && it > 0 && (it % a.pace_every) == 0 && tid == 0) {           \
            uint32_t* ctr = a.pace + strip;                                                        \
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);           \
            const uint32_t want = a.nqb * (it / a.pace_every);                                     \
            bool ok = false;                                                                       \
            /* <= ~20 us (128 cycles of sleep + an L2 round trip per spin): a sibling that is not resident -- a    \
               co-tenant kernel (RCCL) holding its CU, a counter mode that serialises dispatch -- costs ONE such  \
               timeout, after which this work-group stops pacing for the rest of the launch (it used to cost   \
               200000 spins at every pacing point: tens of ms each, indistinguishable from a hang) */              \
            for (uint32_t spin = 0; spin < 64u; ++spin) {                                          \
                if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) { ok = true; break; } \
                __builtin_amdgcn_s_sleep(2);                                                       \
            }                                                                                      \
            pace_on = ok;                                                                          \
        }                                                                                          \
        /* L2: the tile's 256 row norms -> LDS slot of its parity (one 1-KB piece, wave 0) */      \
        if (METRIC == M_L2 && DENSE != 1 && first && wave == 0)                                    \
            VROD_GLDS16(reinterpret_cast<const char*>(a.xnorm2 + (uint64_t)tile * kBM) + lane * 16, lds + kLdsXn2 + (tile & 1) * 1024); \
        W4_PROF(const uint32_t pm0 = w4_clock();)                                                  \
        W4_PHASE(FA0, BX, 0, 0, LDQ0, W4_DMU0)                                                     \
        W4_PHASE(FA0, BY, 0, 1, W4_LDQ1, W4_DMU1)                                                  \
        W4_PROF(const uint32_t pm1 = w4_clock();)                                                  \
        asm volatile(W4_VMWAIT ::: "memory");                                                      \
        VROD_BARRIER();                                                                            \
        W4_PROF(const uint32_t pm2 = w4_clock(); pc[10] += pm2 - pm1; pc[13] += pm1 - pm0; pc[12] += 1;)  \
        W4_PHASE(FA1, BY, 1, 1, W4_LDQ2, W4_DMU2)                                                  \
        W4_PHASE(FA1, BX, 1, 0, LDQ3, W4_DMU3)                                                     \
        stage_advance();                                                                           \
        W4_PROF(const uint32_t pm3 = w4_clock(); pc[14] += pm3 - pm2;)                             \
        const bool last = kt == KT - 1;                                                            \
        W4_PROF(uint32_t pe0 = 0, pe1 = 0;)                                                        \
        if (last) {                                                                                \
            W4_PROF(pe0 = w4_clock();)                                                             \
            if constexpr (DENSE == 2)                                                              \
                w4_groupmax_store_tile<METRIC>(a, qn2_l, xn_l + (tile & 1) * 256 + wr * 128 + fg * 4, wc * 128 + fr, \
                                               qb * kBN + wc * 128 + fr, ((tile - a.tile_first) * 2 + wr) * 4 + fg); \
            else if constexpr (DENSE == 1)                                                         \
                w4_dense_store_tile<METRIC>(a, qn2_l, wc * 128 + fr, tile * kBM + wr * 128 + fg * 4, qb * kBN + wc * 128 + fr); \
            else                                                                                   \
                w4_filter_tile<METRIC>(a, thr_l, qn2_l, xn_l + (tile & 1) * 256 + wr * 128 + fg * 4,        \
                                       tile * kBM + wr * 128 + fg * 4, wc * 128 + fr, qb, rel_base,          \
                                       log + wave * kLogCapW4, log_cnt, wave, wlog W4_PROF(, pc));          \
            W4_PROF(pe1 = w4_clock(); pc[1] += pe1 - pe0;)                                         \
            kt = 0; ++tile;                                                                        \
        } else ++kt;                                                                               \
        asm volatile(W4_VMWAIT ::: "memory");                                                      \
        VROD_BARRIER();                                                                            \
        W4_PROF(if (!DENSE && last) pc[3] += w4_clock() - pe1; else pc[11] += w4_clock() - pm3;)   \
        if (!DENSE && last && log_cnt[3] != 0u) {   /* every wave's appends are behind the barrier */ \
            W4_PROF(const uint32_t pf0 = w4_clock();)                                              \
            flush_log_w4(a, log, log_cnt, qb, rel_base, tid);                                      \
            wlog = 0u;                                                                             \
            W4_PROF(pc[8] += 1; pc[9] += w4_clock() - pf0;)                                        \
        }                                                                                          \
        ++it;                                                                                      \
    }
#define W4_LDQ1(j) W4_LOAD_A1(FA1, 1, l, j)
#define W4_LDQ2(j) W4_LOAD_A1(FA0, 0, ln, j)
#define W4_LDQ0x(j) W4_LOAD_B1(FBy, 1, l, j)
#define W4_LDQ3x(j) W4_LOAD_B1(FBy, 0, ln, j)
#define W4_LDQ0y(j) W4_LOAD_B1(FBx, 1, l, j)
#define W4_LDQ3y(j) W4_LOAD_B1(FBx, 0, ln, j)
#define W4_LDP_A(j) W4_LOAD_A1(FA0, 0, lds, j)
#define W4_LDP_B(j) W4_LOAD_B1(FBx, 0, lds, j)
        W4_LDP_A(0) W4_LDP_A(1) W4_LDP_A(2) W4_LDP_A(3) W4_LDP_A(4) W4_LDP_A(5) W4_LDP_A(6) W4_LDP_A(7)
        W4_LDP_B(0) W4_LDP_B(1) W4_LDP_B(2) W4_LDP_B(3) W4_LDP_B(4) W4_LDP_B(5) W4_LDP_B(6) W4_LDP_B(7)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        VROD_BARRIER();   // every wave holds its first fragments: buffer 0's A_m0 / B_n0 may be restaged

        uint32_t it = 0, kt = 0, tile = t0;
        uint32_t wlog = 0u;   // entries in this wave's log segment (wave-uniform)
        bool pace_on = true;  // (thread 0) false after one pacing timeout: no more pacing in this launch
        W4_PROF(uint32_t pc[16] = {}; const uint32_t pk0 = w4_clock();)
        while (it < total_it) {
            W4_ITER(FBx, FBy, W4_LDQ0x, W4_LDQ3x)
            if (it >= total_it) break;
            W4_ITER(FBy, FBx, W4_LDQ0y, W4_LDQ3y)
        }
        W4_PROF(pc[0] = w4_clock() - pk0;)
#undef W4_LOAD_A1
#undef W4_LOAD_B1
#undef W4_PHASE_S
#undef W4_PHASE
#undef W4_VMWAIT
#undef W4_DMU0
#undef W4_DMU1
#undef W4_DMU2
#undef W4_DMU3
#undef W4_ITER
#undef W4_LDQ1
#undef W4_LDQ2
#undef W4_LDQ0x
#undef W4_LDQ3x
#undef W4_LDQ0y
#undef W4_LDQ3y
#undef W4_LDP_A
#undef W4_LDP_B
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if constexpr (DENSE == 0) flush_log_w4(a, log, log_cnt, qb, rel_base, tid);
        W4_PROF(if (!DENSE && lane < 16) {
            uint32_t v = 0;
            static_for<0, 16>([&](auto ic) { v = lane == decltype(ic)::value ? pc[decltype(ic)::value] : v; });
            atomicAdd(&g_w4_prof[lane], (unsigned long long)v);
        })
    }
}

// ---------------------------------------------------------------------------------------------
// Schedule 4 (bf16 default, plain rows): 4 waves laid out 4 x 1, the corpus never touches LDS.
//
// Wave w owns rows [64 w, 64 w + 64) of the 256-row tile against ALL 256 queries of the block:
// 4 x 16 MFMA tiles, the same 256 accumulator registers a[0:255] (tile (m, n) = a[(m*16+n)*4 .. +3]).
//   * A (corpus rows): nothing is shared between waves any more, so the rows go global -> VGPR
//     as MFMA A fragments (16 rows x 64 B per instruction: lane l holds bytes [16 (l >> 4), +16) of
//     half line kk of row l & 15 -- the lane -> k mapping of the B fragments, so equal elements meet
//     whatever the instruction's internal k order), a ring of 4 K-tiles (128 VGPRs) requested three
//     K-tiles ahead of the MFMAs that use them.  The loads are inline asm: hipcc would wait vmcnt(0)
//     before every use of an ordinary load while LDS-DMA is in flight; their data is valid behind the
//     counted wait that names the registers ("+v").
//   * B (queries): LDS-DMA into FOUR 32-KB stages, three K-tiles ahead, read by all four waves
//     (asm ds_read_b128 + counted lgkmcnt(2): the compiler's own lgkmcnt(0) in front of every MFMA
//     group exposed the latency of the reads just issued for the next group).
//   * ONE barrier per K-tile, placed in front of the last n-tile's MFMAs: the wait in front of it
//     (vmcnt(32): everything requested two K-tiles ago has landed = this wave's A fragments and B
//     pieces of K-tile it+1; lgkmcnt(0): its fragment reads of stage it are back) makes the barrier
//     publish stage it+1 and free stage it for the requests of K-tile it+4; the first fragment reads
//     of K-tile it+1 then hide behind the last 8 MFMAs of K-tile it.
// Every wave issues exactly 16 memory requests per K-tile (8 A fragments, 8 DMA pieces; the L2
// form adds 4 row-norm loads in a tile's first K-tile, which only makes the counted wait stricter).
// Per K-tile the CU's LDS takes 32 KB of fills (4-wave kernel above: 64 KB) and 128 KB of fragment
// reads (the same).  Prototype scripts/ubench/gemm_w4a.hip, same box, 4M x 768: 1.29-1.30 PFLOP/s
// against 1.09 for the structure above (profiles/r02/mfma_experiments.md).
// ---------------------------------------------------------------------------------------------
constexpr int kW4aStageB = 32768;                          // one K-tile of the block's 256 queries
constexpr int kW4aLog = 4 * kW4aStageB;                    // the wave-private log segments behind the four stages
constexpr int kW4aCtl = kW4aLog + kLogCap * 8;
constexpr int kW4aThr = kW4aCtl + 64;                      // [256] f32 thresholds
constexpr int kW4aQn2 = kW4aThr + 1024;                    // [256] f32 query norms (L2)
constexpr int kW4aXn2 = kW4aQn2 + 1024;                    // [2][256] f32 row norms (L2), slot = tile parity
constexpr int kLdsTotalW4a = kW4aXn2 + 2048;

// a pointer hipcc keeps in an SGPR pair ("s" asm operands must be provably wave-uniform)
__device__ __forceinline__ const char* w4a_uniform_ptr(const char* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const char*)(((uint64_t)hi << 32) | lo);
}
// The A ring lives in v[128:255], named literally like the accumulators (slot s, fragment j = m*2+kk ->
// v[128 + (s*8+j)*4 .. +3]): an asm load's data is in flight until the counted vmcnt wait, and registers hipcc
// allocates may be copied, split or re-assigned in between (with the ring in "=v" operands it did exactly that
// under the epilogue's register pressure).  The kernel is compiled with amdgpu_num_vgpr(128): hipcc keeps to
// v0..v127, the clobber list of the kernel's first statement makes the descriptor allocate all 256 + 256
// registers, and scripts/audit_w4.py checks that no compiler instruction names v128 or above.
template <int DST, int OFF>
__device__ __forceinline__ void w4a_load_a(uint32_t voff, const char* sbase) {
    asm volatile("global_load_dwordx4 v[%c2:%c3], %0, %1 offset:%c4" :: "v"(voff), "s"(sbase), "i"(DST), "i"(DST + 3), "i"(OFF) : "memory");
}
template <int ACC, int RA, bool ZERO>
__device__ __forceinline__ void w4a_mfma1(const bf16x8& y) {
    if constexpr (ZERO)
        asm volatile("v_mfma_f32_16x16x32_bf16 a[%c1:%c2], v[%c3:%c4], %0, 0" ::"v"(y), "i"(ACC), "i"(ACC + 3), "i"(RA), "i"(RA + 3) : "memory");
    else
        asm volatile("v_mfma_f32_16x16x32_bf16 a[%c1:%c2], v[%c3:%c4], %0, a[%c1:%c2]" ::"v"(y), "i"(ACC), "i"(ACC + 3), "i"(RA), "i"(RA + 3) : "memory");
}
constexpr int kW4aRing = 128;   // first VGPR of the A ring
// one LDS-DMA piece (1 KB: 64 lanes x 16 B) from sbase + voff into the wave-uniform LDS byte address lds_dst.
// In asm, with M0 saved and restored inside the statement: the builtin form keeps a 64-bit per-lane pointer
// per piece (16 VGPRs the 4 x 1 kernel does not have), and hipcc's waits stay out of the way.
__device__ __forceinline__ void w4a_dma_piece(uint32_t lds_dst, uint32_t voff, const char* sbase) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}
// The fused filter of the 4 x 1 layout: the wave's 64 x 256 scores (a[0:255]) against the lane's 16
// thresholds.  row_w = first row of the lane's 4-row group in tile m = 0; xn_l = its row norms in LDS (L2), 16 floats apart per m.
template <int METRIC>
__device__ __forceinline__ void w4a_filter_tile(const MfmaKernelArgs& a, const float* thr_l, const float* qn2_l, const float* xn_l,
                                                uint32_t row_w, uint32_t fr, uint32_t qb, uint32_t rel_base,
                                                uint2* log /* this wave's segment */, uint32_t* log_cnt, int wave, uint32_t& wlog,
                                                uint32_t flag_word /* log_cnt word that asks for a flush: 2 + tile parity */) {
    const uint32_t wlog_in = wlog;
    const uint32_t lds_log_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)log;
    // Everything the epilogue derives from the lane's column / row is computed HERE, once per tile: left visible,
    // hipcc hoists it out of the scan loop (16 list pointers, 16 shifted column ids, 16 LDS addresses ... ~70
    // loop-invariant VGPRs) and then parks what no longer fits in the accumulator file (scripts/audit_w4.py).
    asm volatile("" : "+v"(fr), "+v"(row_w));
    // the last MFMAs are still in the pipe: an accumulator may be read 4 passes + 2 states later
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    // groups of NG n-tiles: NG thresholds (+ NG query norms) + NG running bests live at a time -- hipcc has
    // 128 VGPRs here (the A ring owns the other 128)
    constexpr int NG = METRIC == M_L2 ? 4 : 8;
    static_for<0, 16 / NG>([&](auto hc) {
        constexpr int h = decltype(hc)::value;
        float thr[NG], qn2[NG], best[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            thr[j] = thr_l[fr + (h * NG + j) * 16];
            qn2[j] = METRIC == M_L2 ? qn2_l[fr + (h * NG + j) * 16] : 0.0f;
            best[j] = worst_score(METRIC);
        }
        static_for<0, 4>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            f32x4 xv = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (METRIC == M_L2) xv = *reinterpret_cast<const f32x4*>(xn_l + m * 16);
            static_for<0, NG>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const f32x4 v = w4_read_acc<(m * 16 + h * NG + j) * 4>();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sc = METRIC == M_COSINE ? v[r] : __builtin_fmaf(-2.0f, v[r], xv[r] + qn2[j]);
                    best[j] = METRIC == M_COSINE ? __builtin_fmaxf(best[j], sc) : __builtin_fminf(best[j], sc);
                }
            });
        });
        uint32_t hitmask = 0u;
#pragma unroll
        for (int j = 0; j < NG; ++j) hitmask |= better<METRIC>(best[j], thr[j]) ? (1u << j) : 0u;
        if (__any(hitmask != 0u)) {
            static_for<0, NG>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if (!__any((hitmask >> j) & 1u)) return;
                const uint32_t ql = fr + (h * NG + j) * 16;
                static_for<0, 4>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    const f32x4 v = w4_read_acc<(m * 16 + h * NG + j) * 4>();
                    f32x4 xv = f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (METRIC == M_L2) xv = *reinterpret_cast<const f32x4*>(xn_l + m * 16);
                    float sc[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[r] = METRIC == M_COSINE ? v[r] : __builtin_fmaf(-2.0f, v[r], xv[r] + qn2[j]);
                    const float tb = METRIC == M_COSINE ? __builtin_fmaxf(__builtin_fmaxf(sc[0], sc[1]), __builtin_fmaxf(sc[2], sc[3]))
                                                        : __builtin_fminf(__builtin_fminf(sc[0], sc[1]), __builtin_fminf(sc[2], sc[3]));
                    if (!__any(better<METRIC>(tb, thr[j]))) return;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t row = row_w + m * 16 + r;
                        const bool hitr = better<METRIC>(sc[r], thr[j]) && row >= a.row_lo && row < a.row_end;
                        // the wave owns a quarter of the log and counts its entries in a register: a ballot and
                        // a lane prefix give every hit its slot (see w4_filter_tile)
                        const unsigned long long hm = __ballot(hitr);
                        if (hm == 0ull) continue;
                        const uint32_t pos = wlog + __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                        if (hitr) {
                            if (pos < (uint32_t)kLogCapW4) {
                                const uint64_t e = ((uint64_t)((ql << 24) | (row - rel_base)) << 32) | __float_as_uint(sc[r]);
                                asm volatile("ds_write_b64 %0, %1" :: "v"(lds_log_addr + pos * 8u), "v"(e) : "memory");
                            } else {
                                global_append(a, qb * kBN + ql, __float_as_uint(sc[r]), row);
                            }
                        }
                        wlog += (uint32_t)__builtin_popcountll(hm);
                    }
                });
            });
        }
    });
    if (wlog != wlog_in) {
        const uint32_t cnt_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)log_cnt;
        asm volatile("ds_write_b32 %0, %1" :: "v"(cnt_addr + 16u + 4u * (uint32_t)wave), "v"(wlog) : "memory");
        if (wlog >= (uint32_t)(kLogCapW4 / 2)) asm volatile("ds_write_b32 %0, %1" :: "v"(cnt_addr + 4u * flag_word), "v"(1u) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int METRIC>
__device__ __forceinline__ void w4a_dense_store_tile(const MfmaKernelArgs& a, const float* qn2_l, const float* xn_l, uint32_t fr,
                                                     uint32_t row_w, uint32_t gq0) {
    asm volatile("" : "+v"(fr), "+v"(row_w), "+v"(gq0));   // computed per tile, not hoisted (see w4a_filter_tile)
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    static_for<0, 16>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        const float qn2 = METRIC == M_L2 ? qn2_l[fr + n * 16] : 0.0f;
        float* out = a.dense_out + (uint64_t)(gq0 + n * 16) * a.dense_ld;
        static_for<0, 4>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const uint32_t row = row_w + m * 16;
            const f32x4 v = w4_read_acc<(m * 16 + n) * 4>();
            f32x4 xv = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (METRIC == M_L2) xv = *reinterpret_cast<const f32x4*>(xn_l + m * 16);
            f32x4 sc;
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[r] = METRIC == M_COSINE ? v[r] : __builtin_fmaf(-2.0f, v[r], xv[r] + qn2);
            if (row - a.row_lo < a.dense_ld) *reinterpret_cast<f32x4*>(out + (row - a.row_lo)) = sc;
        });
    });
}

template <int METRIC, bool DENSE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(128))) void scan_mfma_w4a_kernel(const MfmaKernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // makes the kernel descriptor allocate a[0:255] (accumulators) and v[128:255] (A ring)
    asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
    uint32_t* log_cnt = reinterpret_cast<uint32_t*>(lds + kW4aCtl);  // [3] flush due, [4..7] wave counts
    uint2* log = reinterpret_cast<uint2*>(lds + kW4aLog);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t strip, qb0, qb_step;
    if (!wg_assignment(a, strip, qb0, qb_step)) return;
    const uint32_t t0 = a.tile_first + (uint32_t)((uint64_t)a.ntiles * strip / a.nstrips);
    const uint32_t t1 = a.tile_first + (uint32_t)((uint64_t)a.ntiles * (strip + 1) / a.nstrips);
    if (t0 >= t1) return;

    if (tid == 0) { lds_zero3(log_cnt); lds_zero3(log_cnt + 3); lds_zero3(log_cnt + 5); }

    const uint32_t ld_bytes = a.ld_bytes;
    const uint32_t KT = ld_bytes >> 7;
    const uint32_t rel_base = a.tile_first * kBM;
    const uint32_t fr = lane & 15, fg = lane >> 4, r7 = fr & 7;
    const uint32_t total_it = (t1 - t0) * KT;
    const uint32_t qb = a.qb_base + qb0;   // ONE query block per work-group (see scan_mfma_w4_kernel)
    float* thr_l = reinterpret_cast<float*>(lds + kW4aThr);
    float* qn2_l = reinterpret_cast<float*>(lds + kW4aQn2);
    thr_l[tid] = DENSE ? 0.0f : a.thr[qb * kBN + tid];
    qn2_l[tid] = METRIC == M_L2 ? a.qnorm2[qb * kBN + tid] : 0.0f;
    // (published by the prologue's __syncthreads)

    // B: LDS-DMA piece p = query rows [8p, 8p+8) x one 128-B line; lane -> row lane >> 3, 16-B chunk (lane & 7) ^ row;
    // this wave moves pieces 8 wave .. 8 wave + 7 of every K-tile
    const uint32_t st_row = lane >> 3;
    const uint32_t st_lane_off = st_row * ld_bytes + (((lane & 7) ^ st_row) << 4);
    const uint32_t piece_stride = 8u * ld_bytes;   // piece i of this wave: 8 i rows further (added to the scalar base)
    const char* sb = w4a_uniform_ptr(a.queries + ((uint64_t)qb * kBN + (uint32_t)wave * 64) * ld_bytes);   // K-tile being requested
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const uint32_t dma_dst0 = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)wave * 8192u);   // + stage * 32768 + piece * 1024
    // B fragment of n-tile n, half kk: row n*16 + fr of the stage, chunk (kk*4 + fg) ^ (row & 7).  Two address pairs:
    // stages 0-1 through the 16-bit instruction offset of the first, stages 2-3 of the second
    uint32_t ba[2][2];
    {
        const uint32_t b_frag0 = lds_base + ((fr >> 3) << 10) + (r7 << 7);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            ba[h][0] = b_frag0 + h * 2 * kW4aStageB + (((0 * 4 + fg) ^ r7) << 4);
            ba[h][1] = b_frag0 + h * 2 * kW4aStageB + (((1 * 4 + fg) ^ r7) << 4);
        }
    }
    // A: fragment (m, kk) of the wave's 64 rows = rows [16m, 16m+16) x bytes [64 kk, +64) of the K-tile's line
    // (one lane offset; the 16 m rows go to the scalar base)
    const uint32_t voff = fr * ld_bytes + fg * 16;
    const uint32_t m_stride = 16u * ld_bytes;
    const char* sa = w4a_uniform_ptr(a.corpus + ((uint64_t)t0 * kBM + (uint32_t)wave * 64) * ld_bytes);   // K-tile being requested
    uint32_t st_kt = 0, st_tile = t0;
    auto advance = [&]() {   // next K-tile of the strip (clamped at its end: the last K-tile is requested again, never used)
        const bool in_tile = st_kt + 1 < KT;
        const bool next_tile = !in_tile && st_tile + 1 < t1;
        const int64_t da = in_tile ? 128 : next_tile ? (int64_t)kBM * ld_bytes - (int64_t)(KT - 1) * 128 : 0;
        const int64_t db = in_tile ? 128 : next_tile ? -(int64_t)(KT - 1) * 128 : 0;
        st_kt = in_tile ? st_kt + 1 : next_tile ? 0u : st_kt;
        st_tile += next_tile ? 1u : 0u;
        sa = w4a_uniform_ptr(sa + da);
        sb = w4a_uniform_ptr(sb + db);
    };
    // L2: the tile's 256 row norms -> LDS slot of its parity, one DMA piece by wave 0 in the tile's first K-tile
    const char* xn_base = w4a_uniform_ptr(reinterpret_cast<const char*>(a.xnorm2 + (uint64_t)t0 * kBM));
    const float* xn_lds = reinterpret_cast<const float*>(lds + kW4aXn2);

    bf16x8 FB[2][2];        // double buffer, kk

// A fragment J = m*2 + kk of ring slot S: a[256 + (S*8 + J)*4 .. +3]
#define W4A_LOADA(S, J) w4a_load_a<kW4aRing + ((S) * 8 + (J)) * 4, ((J) & 1) * 64>(voff, sa + (uint32_t)((J) >> 1) * m_stride);
#define W4A_DMAB(S, I) w4a_dma_piece(dma_dst0 + (uint32_t)((S) * kW4aStageB + (I) * 1024), st_lane_off, sb + (uint32_t)(I) * piece_stride);
#define W4A_READB(BUF, ST, N)                                                                       \
    asm volatile("ds_read_b128 %0, %2 offset:%c4\n\tds_read_b128 %1, %3 offset:%c4"                  \
                 : "=v"(FB[BUF][0]), "=v"(FB[BUF][1]) : "v"(ba[(ST) >> 1][0]), "v"(ba[(ST) >> 1][1]), "i"(((ST) & 1) * kW4aStageB + (N) * 2048) : "memory");
// the two reads issued last may stay in flight; everything older (fragments BUF) is back
#define W4A_WAITB(BUF) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(FB[BUF][0]), "+v"(FB[BUF][1]) :: "memory");
// the 8 MFMAs of n-tile N in (kk, m) order: an accumulator comes back after 4 instructions
#define W4A_MFMAS(S, BUF, N, ZERO)                                                                  \
    static_for<0, 8>([&](auto ic) {                                                                 \
        constexpr int kk = decltype(ic)::value / 4, m = decltype(ic)::value % 4;                    \
        w4a_mfma1<(m * 16 + (N)) * 4, kW4aRing + ((S) * 8 + m * 2 + kk) * 4, ZERO && kk == 0>(FB[BUF][kk]); \
    });

    // ---- prologue: K-tiles 0, 1, 2 requested and landed
#define W4A_PROLOGUE_STAGE(S)                                                                       \
    W4A_LOADA(S, 0) W4A_LOADA(S, 1) W4A_LOADA(S, 2) W4A_LOADA(S, 3) W4A_LOADA(S, 4) W4A_LOADA(S, 5) W4A_LOADA(S, 6) W4A_LOADA(S, 7) \
    W4A_DMAB(S, 0) W4A_DMAB(S, 1) W4A_DMAB(S, 2) W4A_DMAB(S, 3) W4A_DMAB(S, 4) W4A_DMAB(S, 5) W4A_DMAB(S, 6) W4A_DMAB(S, 7) \
    advance();
    W4A_PROLOGUE_STAGE(0) W4A_PROLOGUE_STAGE(1) W4A_PROLOGUE_STAGE(2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    W4A_READB(0, 0, 0)

    // Step n of a K-tile in ring slot / stage S (S3 = the slot being refilled): request the B fragments of n+1,
    // 8 MFMAs of n, one memory request (even n: A fragment n/2, odd n: DMA piece n/2)
#define W4A_STEP(S, S3, N, ZERO)                                                                    \
    W4A_READB(((N) + 1) & 1, S, (N) + 1)                                                            \
    W4A_WAITB((N) & 1)                                                                              \
    W4A_MFMAS(S, (N) & 1, N, ZERO)                                                                  \
    if constexpr (((N) & 1) == 0) { W4A_LOADA(S3, (N) >> 1) } else { W4A_DMAB(S3, (N) >> 1) }
#define W4A_KTILE_Z(S, S1, S3, ZERO)                                                                \
    W4A_STEP(S, S3, 0, ZERO) W4A_STEP(S, S3, 1, ZERO) W4A_STEP(S, S3, 2, ZERO) W4A_STEP(S, S3, 3, ZERO)          \
    W4A_STEP(S, S3, 4, ZERO) W4A_STEP(S, S3, 5, ZERO) W4A_STEP(S, S3, 6, ZERO) W4A_STEP(S, S3, 7, ZERO)          \
    W4A_STEP(S, S3, 8, ZERO) W4A_STEP(S, S3, 9, ZERO) W4A_STEP(S, S3, 10, ZERO) W4A_STEP(S, S3, 11, ZERO)        \
    W4A_STEP(S, S3, 12, ZERO) W4A_STEP(S, S3, 13, ZERO)                                             \
    W4A_READB(1, S, 15)                                                                             \
    W4A_WAITB(0)                                                                                    \
    W4A_MFMAS(S, 0, 14, ZERO)                                                                       \
    W4A_LOADA(S3, 7)                                                                                \
    W4A_DMAB(S3, 7)                                                                                 \
    asm volatile("s_waitcnt vmcnt(32) lgkmcnt(0)" : "+v"(FB[1][0]), "+v"(FB[1][1]) :: "memory");     \
    VROD_BARRIER();                                                                                 \
    /* a flush is decided here, behind the first barrier that follows a tile's epilogue: every wave's "flush due" \
       word of that tile (parity-indexed, so the next tile's epilogue cannot race this read) and its count are    \
       published, and no wave is past this point before all have read it (the flush itself synchronises)          \
    */                                                                                              \
    if (!DENSE && flush_word) {                                                                     \
        if (log_cnt[flush_word] != 0u) {                                                            \
            flush_log_w4(a, log, log_cnt, qb, rel_base, tid);                                       \
            wlog = 0u;                                                                              \
        }                                                                                           \
        flush_word = 0u;                                                                            \
    }                                                                                               \
    W4A_READB(0, S1, 0)                                                                             \
    W4A_MFMAS(S, 1, 15, ZERO)
#define W4A_KTILE(S, S1, S3)                                                                        \
    {                                                                                               \
        /* pacing of the sibling work-groups of a strip (they share every corpus tile through their XCD's L2): a \
           relaxed counter barrier every pace_every K-tiles, speed only.  Bounded: a sibling that is not       \
           resident (co-tenant kernels, counter modes that serialise dispatch) costs ONE short timeout, after  \
           which this work-group stops pacing for the rest of the launch */                                     \
        if (a.pace_every && pace_on && it > 0 && (it % a.pace_every) == 0 && tid == 0) {            \
            uint32_t* ctr = a.pace + strip;                                                         \
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);            \
            const uint32_t want = a.nqb * (it / a.pace_every);                                      \
            bool ok = false;                                                                        \
            for (uint32_t spin = 0; spin < 64u; ++spin) {   /* <= ~20 us: 128 cycles of sleep + an L2 round trip per spin */ \
                if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) { ok = true; break; } \
                __builtin_amdgcn_s_sleep(2);                                                        \
            }                                                                                       \
            pace_on = ok;                                                                           \
        }                                                                                           \
        if constexpr (METRIC == M_L2 && !DENSE) {                                                   \
            if (kt == 0) {                                                                          \
                if (wave == 0) w4a_dma_piece(__builtin_amdgcn_readfirstlane(lds_base + kW4aXn2 + (tile & 1u) * 1024u), (uint32_t)lane * 16u, xn_base); \
                xn_base = w4a_uniform_ptr(xn_base + (tile + 1 < t1 ? kBM * 4 : 0));                  \
            }                                                                                       \
        }                                                                                           \
        if (kt == 0) { W4A_KTILE_Z(S, S1, S3, true) } else { W4A_KTILE_Z(S, S1, S3, false) }         \
        advance();                                                                                  \
        if (++kt == KT) {                                                                           \
            kt = 0;                                                                                 \
            /* fewer than 3 K-tiles per tile: the norms were requested less than two K-tiles ago */ \
            if (METRIC == M_L2 && !DENSE && KT < 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); VROD_BARRIER(); } \
            if constexpr (DENSE)                                                                    \
                w4a_dense_store_tile<METRIC>(a, qn2_l, a.xnorm2 + tile * kBM + wave * 64 + fg * 4, fr, tile * kBM + wave * 64 + fg * 4, qb * kBN + fr); \
            else                                                                                    \
                w4a_filter_tile<METRIC>(a, thr_l, qn2_l, xn_lds + (tile & 1u) * 256u + wave * 64 + fg * 4, tile * kBM + wave * 64 + fg * 4, fr, qb, rel_base, \
                                        log + wave * kLogCapW4, log_cnt, wave, wlog, 2u + (tile & 1u)); \
            flush_word = 2u + (tile & 1u);                                                          \
            ++tile;                                                                                 \
        }                                                                                           \
        if (++it >= total_it) break;                                                                \
    }

    uint32_t it = 0, kt = 0, tile = t0;
    uint32_t wlog = 0u;   // entries in this wave's log segment (wave-uniform)
    bool pace_on = true;
    uint32_t flush_word = 0u;   // != 0: the log_cnt word to look at behind the next barrier (wave-uniform)
    for (;;) {
        W4A_KTILE(0, 1, 3)
        W4A_KTILE(1, 2, 0)
        W4A_KTILE(2, 3, 1)
        W4A_KTILE(3, 0, 2)
    }
#undef W4A_LOADA
#undef W4A_DMAB
#undef W4A_READB
#undef W4A_WAITB
#undef W4A_MFMAS
#undef W4A_PROLOGUE_STAGE
#undef W4A_STEP
#undef W4A_KTILE_Z
#undef W4A_KTILE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (!DENSE) flush_log_w4(a, log, log_cnt, qb, rel_base, tid);
}

// ---------------------------------------------------------------------------------------------
// Skinny form: batches of <= 64 (NT = 4) or <= 32 (NT = 2) queries over bf16 rows.  At this width
// the contraction needs 2 * nq flops per corpus byte -- far below the matrix cores' ridge -- so
// the kernel is built like the stream scan, around HBM: every wave streams its own 16-row blocks
// straight from memory into MFMA A fragments (no LDS staging of the corpus: nothing is shared),
// eight 128-B lines per row in flight, requested four at a time, eight waves per CU, while the
// whole query matrix sits in LDS as the B operand for the life of the work-group.  The ring of
// lines runs on across block ends (one stream of lines per wave).  The lane -> k mapping of a
// fragment is the SAME for both operands (lane group g = lane / 16 owns bytes [16 g, 16 g + 16) of
// a 64-B half line), so the contraction pairs equal element indices whatever the instruction's
// internal k order is.
// Same contract as the tiled kernels: DENSE writes every fast score, otherwise scores that beat
// the query's read-only threshold are appended to its list -- through a wave-private LDS segment
// flushed by the wave itself at the end of the block (no block barrier in the loop).
// Measured (2M x 768, 32 / 64 queries, same box): 0.69 / 0.72 ms per search against 0.78-0.82 with
// the 256-query tile; big stage 5.6 TB/s (the 1-query stream scan: 6.5).  Blocks of 64 or 32 rows
// (4 / 2 A fragments per B fragment) were slower (5.0 TB/s over all scans vs 5.3), a ring that
// drains at block ends much slower (4.7), deeper rings and 12-16 waves per CU no faster.
// Roofline: HBM.  Algorithmic bytes per launch = rows * ld_bytes.
// ---------------------------------------------------------------------------------------------
#ifndef VROD_SK_D
#define VROD_SK_D 8      // ring: lines (128 B of every row of the block) in flight per wave
#endif
#ifndef VROD_SK_G
#define VROD_SK_G 4      // lines requested together
#endif
#ifndef VROD_SK_LOAD
#define VROD_SK_LOAD(p) __builtin_nontemporal_load(p)   // (with the continuous ring: same at 32 queries, 2-4 % faster at 64 than plain loads)
#endif
constexpr int kSkLog = 256;   // log entries per wave
#ifndef VROD_SK_WAVES
#define VROD_SK_WAVES 8
#endif
constexpr int kSkWaves = VROD_SK_WAVES;   // waves per work-group, one block each at a time
#ifndef VROD_SK_MT
#define VROD_SK_MT 1
#endif
constexpr int kSkMT = VROD_SK_MT;      // 16-row A fragments per block

template <int NT> constexpr uint32_t skinny_lds_bytes(uint32_t ld_bytes) { return (uint32_t)NT * 16u * (ld_bytes + 32u) + (uint32_t)kSkWaves * kSkLog * 8u; }

template <int METRIC, bool DENSE, int NT, bool SPLIT>
__global__ __launch_bounds__(kSkWaves * 64) void scan_mfma_skinny_kernel(const MfmaKernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t fr = lane & 15, fg = lane >> 4;
    // SPLIT (bf16 planes of fp32 rows, kernels_prep.hip split_rows_kernel): a corpus row is
    // [hi_j | lo_j] per 64-element K-tile (row_bytes = lda_bytes), a query row [hi_j | lo_j | hi_j]
    // (ld_bytes); LDS keeps [hi_j | lo_j] of every query.  Line 2j of a row (hi_x) meets hi_q and
    // lo_q of K-tile j, line 2j+1 (lo_x) meets hi_q: q.x ~ hi.hi + hi.lo + lo.hi.
    const uint32_t row_bytes = SPLIT ? a.lda_bytes : a.ld_bytes;
    // +32 B per query row: a ds_read_b128 is served in four groups of 16 lanes, each mixing two lane
    // quarters (MI355X_MICROARCH.md, LDS); with a row stride of 32 (mod 64) bytes past a 256-B multiple
    // the 16-B slot of lane (row fr, chunk fg) is (2 fr + fg) mod 16 -- even slots for one quarter, odd
    // for the other, no two alike.  (+16 B leaves every group 2-way conflicted: measured 4 extra LDS
    // cycles per read.)
    const uint32_t qstride = row_bytes + 32;
    constexpr uint32_t NQ = NT * 16;
    {
        const uint32_t cpr = row_bytes >> 4;       // 16-B chunks per query row in LDS
        for (uint32_t c = tid; c < NQ * cpr; c += kSkWaves * 64) {
            const uint32_t r = c / cpr, o = c - r * cpr;
            const uint32_t so = SPLIT ? ((o >> 4) * 384u + ((o >> 3) & 1u) * 128u + (o & 7u) * 16u) : o * 16u;
            *reinterpret_cast<uint4*>(lds + r * qstride + o * 16) = *reinterpret_cast<const uint4*>(a.queries + (uint64_t)r * a.ld_bytes + so);
        }
    }
    __syncthreads();
    uint2* log = reinterpret_cast<uint2*>(lds + NQ * qstride) + wave * kSkLog;
    float thr[NT], qn2[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        thr[n] = a.thr[n * 16 + fr];
        qn2[n] = METRIC == M_L2 ? a.qnorm2[n * 16 + fr] : 0.0f;
    }
    const uint32_t KP = row_bytes >> 7;            // 128-B lines per row
    const uint32_t rel_base = a.tile_first * kBM;
    constexpr int MT = kSkMT;
    const uint32_t nblk = a.ntiles * (kBM / (16 * MT));
    const char* qfrag = lds + fr * qstride + fg * 16;

    // The wave's blocks are b0, b0 + bstep, ...; the lines of all of them form ONE stream of
    // nmine * KP lines, walked with a ring of D lines in flight that runs on across block ends
    // (a ring that drains at every block end leaves the memory pipe idle for a latency per block).
    const uint32_t b0 = blockIdx.x * kSkWaves + wave, bstep = gridDim.x * kSkWaves;
    uint32_t nmine = b0 < nblk ? (nblk - b0 + bstep - 1) / bstep : 0u;
    while (nmine && rel_base + (b0 + (nmine - 1) * bstep) * (16 * MT) >= a.row_end) --nmine;   // blocks of pure padding
    if (!nmine) return;
    const uint32_t total = nmine * KP;
    constexpr int D = VROD_SK_D;
    static_assert(!SPLIT || (D % 2 == 0 && VROD_SK_G % 2 == 0), "SPLIT pairs ring-slot parity with plane parity");
    bf16x8 ring[D][MT][2];
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fetch cursor (runs D lines ahead of the compute cursor)
    uint32_t f_kp = 0, f_blk = b0, f_it = 0;
    const char* f_src = a.corpus + (uint64_t)(rel_base + b0 * (16 * MT) + fr) * row_bytes + fg * 16;
    auto fetch1 = [&](bf16x8 (&rs)[2], int m) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
            rs[h] = VROD_SK_LOAD(reinterpret_cast<const bf16x8*>(f_src + (uint64_t)m * 16 * row_bytes + f_kp * 128 + h * 64));
    };
    auto fetch_advance = [&]() {
        ++f_it;
        if (++f_kp == KP) {
            f_kp = 0;
            f_blk += bstep;
            f_src = a.corpus + (uint64_t)(rel_base + f_blk * (16 * MT) + fr) * row_bytes + fg * 16;
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (f_it < total) {
#pragma unroll
            for (int m = 0; m < MT; ++m) fetch1(ring[d][m], m);
            fetch_advance();
        }
    uint32_t c_kp = 0, c_blk = b0;
    auto line = [&](auto dc, uint32_t it) __attribute__((always_inline)) -> bool {
        constexpr int d = decltype(dc)::value;
        if (it >= total) return false;
        // (SPLIT: KP and D are even and a block starts at a multiple of KP, so the slot's parity is the line's)
        constexpr bool lo_x = SPLIT && (d & 1);
        bf16x8 bq[2][NT];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int n = 0; n < NT; ++n) bq[h][n] = *reinterpret_cast<const bf16x8*>(qfrag + n * 16 * qstride + (c_kp - (lo_x ? 1u : 0u)) * 128 + h * 64);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[d][m][h], bq[h][n], acc[m][n], 0, 0, 0);
        if constexpr (SPLIT && !(d & 1)) {   // hi_x . lo_q
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int n = 0; n < NT; ++n) bq[h][n] = *reinterpret_cast<const bf16x8*>(qfrag + n * 16 * qstride + (c_kp + 1) * 128 + h * 64);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[d][m][h], bq[h][n], acc[m][n], 0, 0, 0);
        }
        // refill in bursts of G lines (G x 128 contiguous bytes of every row requested together:
        // DRAM page locality), as soon as the last line of a group of ring slots is consumed
        constexpr int G = VROD_SK_G;
        if constexpr (d % G == G - 1) {
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (f_it < total) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) fetch1(ring[d - (G - 1) + g][m], m);
                    fetch_advance();
                }
        }
        if (++c_kp < KP) return true;
        c_kp = 0;
        const uint32_t row0 = rel_base + c_blk * (16 * MT);
        c_blk += bstep;
        // ---- epilogue of the block: acc[m][n][r] = row row_w + 16 m + r, query 16 n + fr
        const uint32_t row_w = row0 + fg * 4;
        if constexpr (DENSE) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const uint32_t row = row_w + m * 16;
                f32x4 xv = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (METRIC == M_L2) xv = *reinterpret_cast<const f32x4*>(a.xnorm2 + row);
                if (row - a.row_lo < a.dense_ld) {
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        f32x4 sc;
#pragma unroll
                        for (int r = 0; r < 4; ++r) sc[r] = METRIC == M_COSINE ? acc[m][n][r] : __builtin_fmaf(-2.0f, acc[m][n][r], xv[r] + qn2[n]);
                        *reinterpret_cast<f32x4*>(a.dense_out + (uint64_t)(n * 16 + fr) * a.dense_ld + (row - a.row_lo)) = sc;
                    }
                }
            }
        } else {
            float best[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) best[n] = worst_score(METRIC);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f32x4 xv = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (METRIC == M_L2) xv = *reinterpret_cast<const f32x4*>(a.xnorm2 + row_w + m * 16);
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sc = METRIC == M_COSINE ? acc[m][n][r] : __builtin_fmaf(-2.0f, acc[m][n][r], xv[r] + qn2[n]);
                        best[n] = METRIC == M_COSINE ? __builtin_fmaxf(best[n], sc) : __builtin_fminf(best[n], sc);
                    }
            }
            bool any = false;
#pragma unroll
            for (int n = 0; n < NT; ++n) any |= better<METRIC>(best[n], thr[n]);
            if (__any(any)) {
                uint32_t wlog = 0;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    f32x4 xv = f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (METRIC == M_L2) xv = *reinterpret_cast<const f32x4*>(a.xnorm2 + row_w + m * 16);
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        if (!__any(better<METRIC>(best[n], thr[n]))) continue;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float sc = METRIC == M_COSINE ? acc[m][n][r] : __builtin_fmaf(-2.0f, acc[m][n][r], xv[r] + qn2[n]);
                            const uint32_t row = row_w + m * 16 + r;
                            const bool hit = better<METRIC>(sc, thr[n]) && row >= a.row_lo && row < a.row_end;
                            const unsigned long long hm = __ballot(hit);
                            if (hm == 0ull) continue;
                            const uint32_t pos = wlog + __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                            if (hit) {
                                if (pos < (uint32_t)kSkLog) log[pos] = make_uint2(__float_as_uint(sc), ((uint32_t)(n * 16 + fr) << 24) | (row - rel_base));
                                else global_append(a, n * 16 + fr, __float_as_uint(sc), row);
                            }
                            wlog += (uint32_t)__builtin_popcountll(hm);
                        }
                    }
                }
                // the wave drains its own segment: one atomic per entry, 64 entries per round trip
                const uint32_t nlog = wlog < (uint32_t)kSkLog ? wlog : (uint32_t)kSkLog;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                for (uint32_t i = lane; i < nlog; i += 64) {
                    const uint2 e = log[i];
                    global_append(a, e.y >> 24, e.x, rel_base + (e.y & 0xFFFFFFu));
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        return true;
    };
    for (uint32_t it0 = 0; it0 < total; it0 += D) {
        if (!line(std::integral_constant<int, 0>{}, it0)) break;
        if constexpr (D >= 2)
            if (!line(std::integral_constant<int, 1>{}, it0 + 1)) break;
        if constexpr (D >= 3)
            if (!line(std::integral_constant<int, 2>{}, it0 + 2)) break;
        if constexpr (D >= 4)
            if (!line(std::integral_constant<int, 3>{}, it0 + 3)) break;
        if constexpr (D >= 5)
            if (!line(std::integral_constant<int, 4>{}, it0 + 4)) break;
        if constexpr (D >= 6)
            if (!line(std::integral_constant<int, 5>{}, it0 + 5)) break;
        if constexpr (D >= 8) {
            if (!line(std::integral_constant<int, 6>{}, it0 + 6)) break;
            if (!line(std::integral_constant<int, 7>{}, it0 + 7)) break;
        }
        if constexpr (D >= 16) {
            if (!line(std::integral_constant<int, 8>{}, it0 + 8)) break;
            if (!line(std::integral_constant<int, 9>{}, it0 + 9)) break;
            if (!line(std::integral_constant<int, 10>{}, it0 + 10)) break;
            if (!line(std::integral_constant<int, 11>{}, it0 + 11)) break;
            if (!line(std::integral_constant<int, 12>{}, it0 + 12)) break;
            if (!line(std::integral_constant<int, 13>{}, it0 + 13)) break;
            if (!line(std::integral_constant<int, 14>{}, it0 + 14)) break;
            if (!line(std::integral_constant<int, 15>{}, it0 + 15)) break;
        }
    }
}

// Largest batch the skinny kernel takes for rows of `row_bytes` (bf16 rows, or the [hi | lo] planes
// of the split pass): what fits in LDS beside the wave logs.  0: none (VROD_MFMA_SKINNY=0, long rows).
uint32_t mfma_skinny_max_queries(bool split, uint32_t row_bytes) {
    static const bool skinny_on = [] { const char* e = getenv("VROD_MFMA_SKINNY"); return !e || e[0] != '0'; }();
    if (!skinny_on) return 0;
    const uint32_t lds_cap = 160u * 1024u;
    if (!split) return skinny_lds_bytes<4>(row_bytes) <= lds_cap ? 64u : skinny_lds_bytes<2>(row_bytes) <= lds_cap ? 32u : 0u;
    return skinny_lds_bytes<2>(row_bytes) <= lds_cap ? 32u : skinny_lds_bytes<1>(row_bytes) <= lds_cap ? 16u : 0u;
}

// does the dense (sample) form of this launch run on the 2 x 2 4-wave kernel, the one form that can write group bests?
static bool mfma_takes_w4_2x2(const MfmaScanArgs& h, int dtype) {
    static const bool simple = [] { const char* e = getenv("VROD_MFMA_SIMPLE"); return e && e[0] == '1'; }();
    static const bool w4 = [] { const char* e = getenv("VROD_MFMA_W4"); return !e || e[0] != '0'; }();
    static const bool w4a = [] { const char* e = getenv("VROD_MFMA_W4A"); return e && e[0] == '1'; }();
    const bool split = h.a_wrap != 0;
    if (dtype != DT_BF16 || simple || !(w4 || split)) return false;
    const uint32_t ld_bytes = h.ld * 2u, qrow = split ? (h.lda_bytes ? h.lda_bytes : ld_bytes) : ld_bytes;
    if (h.nq > 0 && h.nq <= mfma_skinny_max_queries(split, qrow)) return false;   // the skinny kernel takes it
    return split || !w4a;
}
uint32_t mfma_dense_group_rows(const MfmaScanArgs& h, int dtype) { return mfma_takes_w4_2x2(h, dtype) ? 32u : 0u; }

void launch_scan_mfma(const MfmaScanArgs& h, int dtype, int num_cus, hipStream_t s) {
    const LaunchEvents lev = g_launch_events;   // attached to the dispatch (first / last of a split batch)
    g_launch_events = LaunchEvents{};
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
    a.lda_bytes = h.lda_bytes ? h.lda_bytes : a.ld_bytes;
    a.a_wrap = h.a_wrap;
    a.nqb = h.nq_pad / kBN;
    a.tile_first = h.row_begin / kBM;
    a.ntiles = (h.row_end + kBM - 1) / kBM - a.tile_first;
    a.row_lo = h.row_begin;
    a.row_end = h.row_end;
    a.dense_out = h.dense_out;
    a.dense_ld = h.dense_ld;
    a.dense_group = (h.dense_out && h.dense_grouped) ? 1u : 0u;   // only ever set by a caller that asked mfma_dense_group_rows()
    static const int pace_env = [] { const char* e = getenv("VROD_MFMA_PACE"); return e ? atoi(e) : 16; }();
    a.pace = h.pace;
    int grid = num_cus / 8 * 8;
    if (grid < 8) grid = 8;
    a.slots = grid / 8;
    a.strips_per_xcd = a.nqb <= a.slots ? a.slots / a.nqb : 1;
    a.nstrips = 8 * a.strips_per_xcd;
    // pacing only where several work-groups share a strip (one per query block) and one pass each
    a.pace_every = (h.pace && a.nqb > 1 && a.nqb <= a.slots && pace_env > 0) ? (uint32_t)pace_env : 0u;
    if (a.pace_every && !h.pace_is_zero) (void)hipMemsetAsync(a.pace, 0, a.nstrips * sizeof(uint32_t), s);
    static const bool simple = [] { const char* e = getenv("VROD_MFMA_SIMPLE"); return e && e[0] == '1'; }();
#define VROD_MFMA(KERNEL, TT, MM)                                                                           \
    do {                                                                                                    \
        static bool attr_set = false;                                                                       \
        if (!attr_set) {                                                                                    \
            (void)hipFuncSetAttribute((const void*)KERNEL<TT, MM>,                                          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);               \
            attr_set = true;                                                                                \
        }                                                                                                   \
        hipExtLaunchKernelGGL((KERNEL<TT, MM>), dim3(grid), dim3(512), kLdsTotal, s, lev.start, lev.stop, 0, a); \
    } while (0)
    static const int gp = [] { const char* e = getenv("VROD_MFMA_GP"); return e ? atoi(e) : 0; }();
    static const bool w4 = [] { const char* e = getenv("VROD_MFMA_W4"); return !e || e[0] != '0'; }();
    static const int pace_kt = [] { const char* e = getenv("VROD_MFMA_PACE_KT"); return e ? atoi(e) : 192; }();
#define VROD_MFMA_W4K_(MM, DN, SP)                                                                          \
    do {                                                                                                    \
        static bool attr_set = false;                                                                       \
        if (!attr_set) {                                                                                    \
            (void)hipFuncSetAttribute((const void*)scan_mfma_w4_kernel<MM, DN, SP>,                         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotalW4);             \
            attr_set = true;                                                                                \
        }                                                                                                   \
        hipExtLaunchKernelGGL((scan_mfma_w4_kernel<MM, DN, SP>), dim3(grid), dim3(256), kLdsTotalW4, s,        \
                              first_launch ? lev.start : nullptr, last_launch ? lev.stop : nullptr, 0, a);        \
    } while (0)
#define VROD_MFMA_W4AK(MM, DN)                                                                              \
    do {                                                                                                    \
        static bool attr_set = false;                                                                       \
        if (!attr_set) {                                                                                    \
            (void)hipFuncSetAttribute((const void*)scan_mfma_w4a_kernel<MM, DN>,                            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotalW4a);            \
            attr_set = true;                                                                                \
        }                                                                                                   \
        hipExtLaunchKernelGGL((scan_mfma_w4a_kernel<MM, DN>), dim3(grid), dim3(256), kLdsTotalW4a, s,          \
                              first_launch ? lev.start : nullptr, last_launch ? lev.stop : nullptr, 0, a);        \
    } while (0)
// plain bf16 rows: the 4 x 1 kernel (corpus global -> VGPR); the SPLIT form of the pass keeps the 2 x 2 kernel
#define VROD_MFMA_W4K(MM, DN) do { if (split) VROD_MFMA_W4K_(MM, DN, true); else if (w4a) VROD_MFMA_W4AK(MM, (DN) != 0); else VROD_MFMA_W4K_(MM, DN, false); } while (0)
#define VROD_MFMA_W4G(MM) do { if (split) VROD_MFMA_W4K_(MM, 2, true); else VROD_MFMA_W4K_(MM, 2, false); } while (0)
    const bool split = h.a_wrap != 0;
    // VROD_MFMA_W4A=1: the 4 x 1 kernel (corpus global -> VGPR).  Measured on the same box it ties or loses to the
    // 2 x 2 kernel by 1-3 % (profiles/r02/mfma_experiments.md): half the LDS fills, but the A fragments cost the
    // vector-memory path as much as the DMA pieces they replace.  Kept selectable, covered by the same tests.
    static const bool w4a = [] { const char* e = getenv("VROD_MFMA_W4A"); return e && e[0] == '1'; }();
    if (dtype == DT_BF16 && h.nq > 0 && h.nq <= mfma_skinny_max_queries(split, split ? a.lda_bytes : a.ld_bytes)) {
        // rows in LDS: the K extent of a corpus row ([hi | lo] planes in the split form)
        const uint32_t qrow = split ? a.lda_bytes : a.ld_bytes;
        const uint32_t lds_cap = 160u * 1024u;
        int nt = 0;
        if (!split) nt = (h.nq > 32) ? 4 : 2;
        else nt = (h.nq > 16 || skinny_lds_bytes<1>(qrow) > lds_cap) ? 2 : 1;
        if (!split && nt == 2 && skinny_lds_bytes<2>(qrow) > lds_cap) nt = 0;   // (cannot happen: max_queries said so)
        if (nt) {
            // one wave per 16-row block at a time; all CUs, but no more work-groups than blocks / 8
            const uint32_t nblk = a.ntiles * (kBM / (16 * kSkMT));
            const int sgrid = (int)std::max<uint32_t>(1u, std::min<uint32_t>((uint32_t)num_cus, (nblk + kSkWaves - 1) / kSkWaves));
            const uint32_t lds_bytes = nt == 4 ? skinny_lds_bytes<4>(qrow) : nt == 2 ? skinny_lds_bytes<2>(qrow) : skinny_lds_bytes<1>(qrow);
#define VROD_MFMA_SK(MM, DN, NN, SP)                                                                        \
    do {                                                                                                    \
        static bool attr_set = false;                                                                       \
        if (!attr_set) {                                                                                    \
            (void)hipFuncSetAttribute((const void*)scan_mfma_skinny_kernel<MM, DN, NN, SP>,                 \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);              \
            attr_set = true;                                                                                \
        }                                                                                                   \
        hipExtLaunchKernelGGL((scan_mfma_skinny_kernel<MM, DN, NN, SP>), dim3(sgrid), dim3(kSkWaves * 64), lds_bytes, s, lev.start, lev.stop, 0, a); \
    } while (0)
#define VROD_MFMA_SKN(MM, DN)                                                                               \
    do {                                                                                                    \
        if (split) { if (nt == 2) VROD_MFMA_SK(MM, DN, 2, true); else VROD_MFMA_SK(MM, DN, 1, true); }      \
        else { if (nt == 4) VROD_MFMA_SK(MM, DN, 4, false); else VROD_MFMA_SK(MM, DN, 2, false); }          \
    } while (0)
            if (h.metric == M_COSINE) { if (h.dense_out) VROD_MFMA_SKN(M_COSINE, true); else VROD_MFMA_SKN(M_COSINE, false); }
            else { if (h.dense_out) VROD_MFMA_SKN(M_L2, true); else VROD_MFMA_SKN(M_L2, false); }
#undef VROD_MFMA_SKN
#undef VROD_MFMA_SK
            return;
        }
    }
    if (dtype == DT_BF16 && (w4 || split) && !simple) {
        const uint32_t nqb_total = a.nqb;
        for (uint32_t qb_base = 0; qb_base < nqb_total; qb_base += a.slots) {   // one launch unless nq > 256 * slots
            a.qb_base = qb_base;
            a.nqb = std::min<uint32_t>(a.slots, nqb_total - qb_base);
            const bool first_launch = qb_base == 0, last_launch = qb_base + a.slots >= nqb_total;
            a.strips_per_xcd = a.slots / a.nqb;
            a.nstrips = 8 * a.strips_per_xcd;
            a.pace_every = (h.pace && a.nqb > 1 && pace_kt > 0) ? (uint32_t)pace_kt : 0u;   // K-tiles
            if (a.pace_every && (qb_base > 0 || !h.pace_is_zero)) (void)hipMemsetAsync(a.pace, 0, a.nstrips * sizeof(uint32_t), s);
            // (the 4 x 1 kernel has no grouped sample form: mfma_dense_group_rows() says 0 for it)
            const bool grp = a.dense_group != 0 && (split || !w4a);
            if (h.metric == M_COSINE) { if (!h.dense_out) VROD_MFMA_W4K(M_COSINE, 0); else if (grp) VROD_MFMA_W4G(M_COSINE); else VROD_MFMA_W4K(M_COSINE, 1); }
            else { if (!h.dense_out) VROD_MFMA_W4K(M_L2, 0); else if (grp) VROD_MFMA_W4G(M_L2); else VROD_MFMA_W4K(M_L2, 1); }
        }
        return;
    }
#undef VROD_MFMA_W4K
#undef VROD_MFMA_W4G
#undef VROD_MFMA_W4AK
#undef VROD_MFMA_W4K_
#define VROD_MFMA_P(TT, MM, GPV, DN)                                                                            \
    do {                                                                                                    \
        static bool attr_set = false;                                                                       \
        if (!attr_set) {                                                                                    \
            (void)hipFuncSetAttribute((const void*)scan_mfma_phased_kernel<TT, MM, GPV, DN>,                    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);               \
            attr_set = true;                                                                                \
        }                                                                                                   \
        hipExtLaunchKernelGGL((scan_mfma_phased_kernel<TT, MM, GPV, DN>), dim3(grid), dim3(512), kLdsTotal, s, lev.start, lev.stop, 0, a); \
    } while (0)
#define VROD_MFMA_BOTH(TT, MM)                                                                              \
    do {                                                                                                    \
        if (h.dense_out) VROD_MFMA_P(TT, MM, 0, true);                                                      \
        else if (simple) VROD_MFMA(scan_mfma_kernel, TT, MM);                                               \
        else if (gp == 1) VROD_MFMA_P(TT, MM, 1, false);                                                    \
        else if (gp == 2) VROD_MFMA_P(TT, MM, 2, false);                                                    \
        else VROD_MFMA_P(TT, MM, 0, false);                                                                 \
    } while (0)
    if (dtype == DT_BF16) { if (h.metric == M_COSINE) VROD_MFMA_BOTH(bf16_t, M_COSINE); else VROD_MFMA_BOTH(bf16_t, M_L2); }
    else { if (h.metric == M_COSINE) VROD_MFMA_BOTH(float, M_COSINE); else VROD_MFMA_BOTH(float, M_L2); }
#undef VROD_MFMA_BOTH
#undef VROD_MFMA_P
#undef VROD_MFMA
}

}  // namespace vrod

#ifdef VROD_W4_PROF
extern "C" int vrod_debug_w4_prof(unsigned long long* out16, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(vrod::g_w4_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(vrod::g_w4_prof), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif
