// kernels_mfma_phased.hip -- the 8-wave phased schedule of the batched Q.K^T scan (see kernels_mfma.hip
// for the family).  Since round 3 it carries the fp32 matrix-core pass only (v_mfma_f32_16x16x4_f32,
// exact fp32 fma chain: fp32 handles without bf16 planes, and their band pass); bf16 rows and the
// split planes of fp32 rows always take the 4-wave kernel (kernels_mfma_w4.hip).
//
// Fills the batched half of the scan slot under SearchSimilarCommand::execute (reference
// src/command/types.rs:121-132, empty).  GEMM view: M = corpus rows, N = queries, K = vector
// dimension; both operands are stored [row][k] with k contiguous, so both MFMA fragments are 16-B
// LDS reads.
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
// Filter: a score that beats its query's read-only threshold is appended to an LDS log
// (one ds_add_rtn per hit); the log is flushed to per-query lists in HBM with global
// atomics (rare).  Thresholds are refreshed between launches (levels) by
// list_compact_kernel, so every launch of this kernel is a pure function of its inputs.
//
// Roofline: MFMA (fp32: 157 TF).  Algorithmic flops per launch = 2 * nq * rows * dim  (SURVEY.md 8d).
#include "mfma_common.h"

namespace vrod {

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
// Staggered wave groups, 4 phases per K-tile, counted vmcnt.
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

void launch_mfma_phased_f32(const MfmaKernelArgs& a, int metric, bool dense, int grid, hipStream_t s, hipEvent_t start, hipEvent_t stop) {
#define VROD_MFMA_P(MM, DN)                                                                                 \
    do {                                                                                                    \
        static bool attr_set = false;                                                                       \
        if (!attr_set) {                                                                                    \
            (void)hipFuncSetAttribute((const void*)scan_mfma_phased_kernel<float, MM, 0, DN>,               \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);               \
            attr_set = true;                                                                                \
        }                                                                                                   \
        hipExtLaunchKernelGGL((scan_mfma_phased_kernel<float, MM, 0, DN>), dim3(grid), dim3(512), kLdsTotal, s, start, stop, 0, a); \
    } while (0)
    if (metric == M_COSINE) { if (dense) VROD_MFMA_P(M_COSINE, true); else VROD_MFMA_P(M_COSINE, false); }
    else { if (dense) VROD_MFMA_P(M_L2, true); else VROD_MFMA_P(M_L2, false); }
#undef VROD_MFMA_P
}

}  // namespace vrod
