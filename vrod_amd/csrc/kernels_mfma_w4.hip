// kernels_mfma_w4.hip -- the 4-wave (2 x 2) schedule of the batched Q.K^T scan: the bf16 default (cfg3 / cfg4) and the
// SPLIT form over the [hi | lo] planes of fp32 rows (cfg5).  Family overview: kernels_mfma.hip.  The accumulator file
// a[0:255] is owned by the inline asm of this file: scripts/audit_w4.py runs on its assembly in the build (Makefile).
#include "mfma_common.h"

namespace vrod {

// ---------------------------------------------------------------------------------------------
// 4 waves, one per SIMD, 128 rows x 128 queries per wave.
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
#ifdef VROD_W4_CLK
// diagnostic build only (-DVROD_W4_CLK): two stamps per work-group around the scan loop of the filtered launches -- shader
// clock (s_memtime) and the 100 MHz real-time counter (s_memrealtime) -- summed over work-groups: [0] shader cycles
// [1] real-time ticks [2] work-groups [3] K-tiles.  In-kernel clock = [0] / [1] * 100 MHz (MI355X_MICROARCH.md, DVFS item 6);
// cycles per K-tile = [0] / [3].  No stamp executes in the product build.
__device__ unsigned long long g_w4_clk[8];   // [4] cycles in tile epilogues (wave 0) [5] in the wait + barrier behind them [6] epilogues
#define W4_CLK(...) __VA_ARGS__
#else
#define W4_CLK(...)
#endif

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

// ---------------------------------------------------------------------------------------------
// The hit dump.  A column of the wave's tile (16 queries x its 128 rows) whose running best beats the threshold used
// to be WALKED by the wave on the spot -- eight 16 x 16 tiles tested, the marked ones read back through a computed
// jump, every hit appended through ballots to an LDS log that the work-group flushed behind three barriers: ~1900
// cycles per column issued by ONE wave while the other three wait at the next barrier (23 % of the wave-tiles of a
// 10M-row batch, 67 % at a 1.25M-row shard: profiles/r02/q_w4_cycle_profile.txt).  Now the lanes that hold a hit
// (usually one) write their 32 scores of that column STRAIGHT FROM THE ACCUMULATOR FILE (ds_write_b128 with an AGPR
// data operand, EXEC = the hit lanes) into the wave's own log in LDS, one entry of 128 B per lane plus a descriptor
// word: nine LDS writes of straight-line code per column, no test per tile, no jump, no barrier.  (LDS, not global memory: stores from the accumulator file to a global dump buffer were
// measured first -- they queue behind the LDS-DMA pieces in vmcnt order, so the next counted wait sat out the latency
// of pieces just requested: 10M rows -0.1 %, where this form takes its gain; widening that wait by the number of
// stores needs a branch chain in front of both waits of every K-tile: +3 %.  profiles/r03/mfma_experiments.md.)
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kDumpEntryBytes = 128;                          // 32 fp32: [row tile m][r]
constexpr uint32_t kDumpWaveBytes = (kLdsTotalW4 - kLdsDump) / 4;   // per wave: entries, then one descriptor word each
constexpr uint32_t kDumpCap = kDumpWaveBytes / (kDumpEntryBytes + 4);
static_assert(kDumpRegionBytes == kDumpRegionCap * (kDumpEntryBytes + 4), "spill region layout");
// three words in the slack behind wave 0's log: the range of tiles wave 0 claimed for the work-group {begin, end} and its
// next own chunk (work stealing, below)
constexpr int kLdsNext = kLdsDump + kDumpCap * (kDumpEntryBytes + 4);
static_assert(kLdsNext % 8 == 0 && kLdsNext + 12 <= kLdsDump + (int)kDumpWaveBytes, "the claimed range sits in the slack of wave 0's log");
static_assert(kDumpCap >= 48 && kDumpCap * (kDumpEntryBytes + 4) <= kDumpWaveBytes && kDumpWaveBytes % 16 == 0, "hit dump layout");

template <int N>
__device__ __forceinline__ void w4_dump_column(uint32_t entry_addr /* per lane, LDS byte address */, uint32_t desc_addr, uint32_t desc) {
    // tile (m, N) = a[(m*8+N)*4 .. +3]; entry = [m][r] fp32.  In asm: as compiler-visible LDS stores these would be
    // preceded by s_waitcnt vmcnt(0) (possible alias with the LDS-DMA destination)
    asm volatile("ds_write_b128 %0, a[%c3:%c4]\n\t"
                 "ds_write_b128 %0, a[%c5:%c6] offset:16\n\t"
                 "ds_write_b128 %0, a[%c7:%c8] offset:32\n\t"
                 "ds_write_b128 %0, a[%c9:%c10] offset:48\n\t"
                 "ds_write_b128 %0, a[%c11:%c12] offset:64\n\t"
                 "ds_write_b128 %0, a[%c13:%c14] offset:80\n\t"
                 "ds_write_b128 %0, a[%c15:%c16] offset:96\n\t"
                 "ds_write_b128 %0, a[%c17:%c18] offset:112\n\t"
                 "ds_write_b32 %1, %2"
                 :: "v"(entry_addr), "v"(desc_addr), "v"(desc),
                    "i"(N * 4), "i"(N * 4 + 3), "i"(N * 4 + 32), "i"(N * 4 + 35), "i"(N * 4 + 64), "i"(N * 4 + 67),
                    "i"(N * 4 + 96), "i"(N * 4 + 99), "i"(N * 4 + 128), "i"(N * 4 + 131), "i"(N * 4 + 160), "i"(N * 4 + 163),
                    "i"(N * 4 + 192), "i"(N * 4 + 195), "i"(N * 4 + 224), "i"(N * 4 + 227)
                 : "memory");
}

// A full log (and, at the end of the launch, what is left in it) is SPILLED to the wave's region of a global buffer:
// plain 16-B copies, nothing waits for them.  The entries are only looked at once the scan loop is over
// (w4_process_region): every score of every entry tested, hits appended to the per-query lists with one atomic each.
// Inside the loop that would be a chain of dependent round trips (threshold, atomic, store) issued by one wave while
// the other three wait at the next barrier -- measured: ~600 cycles per entry; at the end of the launch every wave of
// the chip does it at once and nothing waits behind it.  A region that would overflow (512 entries per wave and launch:
// duplicate-heavy corpora, band scans) is processed on the spot.
template <int METRIC>
__device__ __forceinline__ void w4_process_region(const MfmaKernelArgs& a, const float* thr_l, const float* qn2_l, const char* region,
                                                  uint32_t n_entries, uint32_t qb, int wr, int wc, int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the spill stores have reached L2
    const uint32_t* desc = reinterpret_cast<const uint32_t*>(region + kDumpRegionCap * kDumpEntryBytes);
    // lane i works on quad (entry i / 8, row tile m = i % 8); nt loads: served by L2, where the write-through stores are.
    // One wave per SIMD and every step a round trip (entry -> test -> list counter -> list slot): kU quads per lane are
    // in flight at a time, and a quad takes ONE atomic for all its hits -- 3 round trips per 4 x 8 entries instead of
    // up to 5 per 8 (a 1.25M-row shard stage spent ~30 us here, a k = 1000 stage 0.7 ms).
    constexpr int kU = 4;
    const uint32_t nquads = n_entries * 8u;
#pragma unroll 1
    for (uint32_t base = 0; base < nquads; base += 64u * kU) {
        f32x4 v[kU];
        uint32_t d[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const uint32_t i = base + (uint32_t)u * 64u + (uint32_t)lane;
            const uint32_t ii = i < nquads ? i : 0u;   // (a lane past the end re-reads quad 0 and drops it below)
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(region + (ii >> 3) * kDumpEntryBytes + (ii & 7u) * 16u));
            d[u] = __builtin_nontemporal_load(desc + (ii >> 3));
        }
        uint32_t gq[kU], row0[kU], hits[kU], pos[kU];
        float sc[kU][4];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const uint32_t i = base + (uint32_t)u * 64u + (uint32_t)lane;
            const uint32_t m = i & 7u;
            const uint32_t ln = d[u] & 63u, n = (d[u] >> 6) & 7u, tile = a.tile_first + (d[u] >> 9);
            const uint32_t ql = wc * 128 + n * 16 + (ln & 15u);
            gq[u] = qb * kBN + ql;
            row0[u] = tile * kBM + wr * 128 + m * 16 + (ln >> 4) * 4;
            const float thr = thr_l[ql];
            const float qn2 = METRIC == M_L2 ? qn2_l[ql] : 0.0f;
            uint32_t h = 0u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t row = row0[u] + r;
                sc[u][r] = METRIC == M_COSINE ? v[u][r] : __builtin_fmaf(-2.0f, v[u][r], a.xnorm2[row] + qn2);
                h |= (better<METRIC>(sc[u][r], thr) && row >= a.row_lo && row < a.row_end) ? 1u << r : 0u;
            }
            hits[u] = i < nquads ? h : 0u;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            pos[u] = 0u;
            if (hits[u]) pos[u] = atomicAdd(&a.counts[gq[u]], (uint32_t)__builtin_popcount(hits[u]));
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if ((hits[u] >> r) & 1u) {
                    if (pos[u] < a.cap) a.lists[(uint64_t)gq[u] * a.cap + pos[u]] = make_uint2(__float_as_uint(sc[u][r]), row0[u] + r);
                    ++pos[u];
                }
    }
    // The list stores are retired HERE, with a wait hipcc's counter bookkeeping sees: left pending they make it guard the
    // next writes of their data registers with s_waitcnt vmcnt(0) -- in the scan loop, where that drains the LDS-DMA queue
    // every K-tile (measured: +16 % per batch).
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
}
// log (n entries in LDS) -> region, behind the wglob entries already there
template <int METRIC>
__device__ __forceinline__ void w4_spill(const MfmaKernelArgs& a, const float* thr_l, const float* qn2_l, const char* wlog_lds, uint32_t n,
                                         char* region, uint32_t& wglob, uint32_t qb, int wr, int wc, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the asm LDS writes of the dump
    if (wglob + n > kDumpRegionCap) {
        w4_process_region<METRIC>(a, thr_l, qn2_l, region, wglob, qb, wr, wc, lane);
        wglob = 0u;
    }
    for (uint32_t i = lane; i < n * 8u; i += 64u)
        *reinterpret_cast<f32x4*>(region + (size_t)wglob * kDumpEntryBytes + i * 16u) = *reinterpret_cast<const f32x4*>(wlog_lds + i * 16u);
    if ((uint32_t)lane < n)
        reinterpret_cast<uint32_t*>(region + kDumpRegionCap * kDumpEntryBytes)[wglob + lane] =
            reinterpret_cast<const uint32_t*>(wlog_lds + kDumpCap * kDumpEntryBytes)[lane];
    wglob += n;
}

// The fused filter of the 4-wave kernel: the wave's 128 x 128 scores (in a[0:255]) against the 8
// per-lane thresholds.  tile = the corpus tile; wlog_lds = the wave's log in LDS, wlog = entries in it.
template <int METRIC>
__device__ __forceinline__ void w4_filter_tile(const MfmaKernelArgs& a, const float* thr_l, const float* qn2_l, const float* xn_l,
                                               uint32_t tile, uint32_t ql0, uint32_t qb, char* wlog_lds, char* region, int wr, int wc, int lane,
                                               uint32_t& wlog, uint32_t& wglob ) {
    // thr_l / qn2_l: the work-group's per-query values in LDS; xn_l: this lane's 4-row group of the
    // tile's row norms in LDS (m = 0), 16 floats apart per m
    float thr[8], qn2[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        thr[n] = thr_l[ql0 + n * 16];
        qn2[n] = METRIC == M_L2 ? qn2_l[ql0 + n * 16] : 0.0f;
    }
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
#ifdef VROD_W4_AUDIT_SELFTEST
    // tests/test_build_audit.py: a compiler-written AGPR while the accumulators are live -- the build must refuse this
    asm volatile("; selftest %0" :: "a"(best[0] + best[1]));
#endif
    // columns (n) that hold a hit, as a wave-uniform bit mask; pend: this lane's own hit columns
    uint32_t colmask = 0u, pend = 0u;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const bool hl = better<METRIC>(best[n], thr[n]);
        pend |= hl ? 1u << n : 0u;
        colmask |= __any(hl) ? 1u << n : 0u;
    }
    
    if (colmask) {
        const uint32_t log_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)wlog_lds;
        const uint32_t dword = ((tile - a.tile_first) << 9) | (uint32_t)lane;
        // one round dumps what fits; lanes left over (a full log) wait for the spill and the next round
#pragma unroll 1
        for (;;) {
            bool again = false;
            static_for<0, 8>([&](auto nc) {
                constexpr int n = decltype(nc)::value;
                if (!(colmask & (1u << n))) return;
                const bool hl = (pend >> n) & 1u;
                const unsigned long long hm = __ballot(hl);
                if (hm == 0ull) return;
                
                const uint32_t pos = wlog + __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                const bool fit = hl && pos < kDumpCap;
                if (fit) {
                    w4_dump_column<n>(log_addr + pos * kDumpEntryBytes, log_addr + kDumpCap * kDumpEntryBytes + pos * 4u, dword | ((uint32_t)n << 6));
                    pend &= ~(1u << n);
                }
                const unsigned long long fm = __ballot(fit);
                wlog += (uint32_t)__builtin_popcountll(fm);
                again |= fm != hm;
            });
            if (!again) break;
            w4_spill<METRIC>(a, thr_l, qn2_l, wlog_lds, wlog, region, wglob, qb, wr, wc, lane);
            wlog = 0u;
            
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the asm LDS writes above are not tracked by the compiler
    }
    
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

// One LDS-DMA piece (1 KB: 64 lanes x 16 B) in asm: global address = the wave-uniform base `sbase` (the K-tile being staged)
// + the lane's constant 32-bit offset `voff` (row of the piece, swizzled 16-B chunk), LDS destination = the wave-uniform
// `lds_buf` + the piece's immediate IMM.  As builtin calls with per-lane 64-bit pointers every piece cost a 64-bit vector
// add for its address (16 per K-tile, issued in the shadow of the MFMAs but issued: one wave per SIMD gets one instruction
// per ~4 cycles) and an SGPR for its LDS destination (32 of them: hipcc spilled those to VGPR lanes and read them back
// with v_readlane in front of every piece); here a piece is s_add (M0) + s_nop + the load, the running position lives in two
// SGPR pairs advanced once per K-tile, and the 16 lane offsets are loop constants (profiles/r03/mfma_experiments.md).
// Invisible to hipcc's vmcnt bookkeeping, like the counted waits that retire the pieces.  M0 belongs to these statements
// (it is a reserved register: a clobber would be ignored): scripts/audit_w4.py checks that hipcc itself never names it.
template <int IMM>
__device__ __forceinline__ void w4_dma_piece(uint32_t lds_buf, uint32_t voff, const char* sbase) {
    asm volatile("s_add_u32 m0, %0, %c3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(lds_buf), "v"(voff), "s"(sbase), "i"(IMM) : "scc", "memory");
}
// the same in two statements, for the main loop: M0 is written a group of MFMAs ahead of the load that reads it (the wait
// state the hardware wants between the two then costs no s_nop: 16 fewer instructions per K-tile)
template <int IMM>
__device__ __forceinline__ void w4_dma_m0(uint32_t lds_buf) {
    asm volatile("s_add_u32 m0, %0, %c1" :: "s"(lds_buf), "i"(IMM) : "scc", "memory");
}
__device__ __forceinline__ void w4_dma_load(uint32_t voff, const char* sbase) {
    asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase) : "memory");
}
// a pointer hipcc keeps in an SGPR pair ("s" asm operands must be provably wave-uniform)
__device__ __forceinline__ const char* w4_uniform_ptr(const char* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const char*)(((uint64_t)hi << 32) | lo);
}

// ---------------------------------------------------------------------------------------------
// Work stealing at the end of a launch.  The 256 work-groups of a launch do not run at one speed: on the same binary the
// work-groups of the odd XCDs take ~2.5 % longer than those of the even ones, and within an XCD the spread is another
// +-2..3 % (profiles/r03: per-XCD loop times of the clock-probe build), so with equal static shares the fastest work-group
// idles for 6-9 % of the launch and the launch lasts as long as its slowest one: 2.7-3.4 % above the mean.  Each strip
// therefore keeps only the first 15/16 of its tile range as its static share; the rest (its TAIL) is cut into chunks of 2
// tiles that any work-group of the same query block may claim (sweep: profiles/r03/mfma_experiments.md): the owner takes its own chunks in ascending order behind its
// static share, a work-group that has run out of its own takes the highest free chunk of another strip.  A claim is one
// atomic OR on a bit of claims[query block][strip] (zeroed per launch by the launcher), so every (chunk, query block) pair
// is scanned exactly once whoever takes it; nobody waits for anybody -- a work-group that finds nothing free ends.  The
// claim is made by wave 0 when the staging cursor enters the LAST tile of the range in hand (a tile = KT K-tiles before
// the result is needed), published to the other waves through two words of LDS, and the staging pointers jump there
// without draining the pipeline.  A stolen chunk is scanned without its three sibling query blocks beside it (its corpus
// tiles are not shared through the XCD's L2): measured, no extra HBM reads to speak of (profiles/r03/x_traffic_sweep.txt:
// 18.48 GB per cfg3 batch against 18.43 with static shares).
// ---------------------------------------------------------------------------------------------
#ifndef VROD_W4_STEAL_CHUNK
#define VROD_W4_STEAL_CHUNK 2
#endif
#ifndef VROD_W4_STEAL_DIV
#define VROD_W4_STEAL_DIV 16
#endif
constexpr uint32_t kStealChunk = VROD_W4_STEAL_CHUNK;     // tiles per claimable chunk
__host__ __device__ inline uint32_t w4_tail_tiles(uint32_t strip_tiles) {   // tiles of a strip that are handed out dynamically
    if (strip_tiles < 48u) return 0u;                                       // short launches (first stages, shards of small corpora): static
    const uint32_t t = strip_tiles / (uint32_t)VROD_W4_STEAL_DIV;
    return t > 32u * kStealChunk ? 32u * kStealChunk : t;                   // one 32-bit word of claim bits per (query block, strip)
}
__device__ __forceinline__ void w4_strip_range(const MfmaKernelArgs& a, uint32_t s, uint32_t& b, uint32_t& e) {
    b = a.tile_first + (uint32_t)((uint64_t)a.ntiles * s / a.nstrips);
    e = a.tile_first + (uint32_t)((uint64_t)a.ntiles * (s + 1) / a.nstrips);
}
// wave 0: the next range of tiles [cb, ce) of this work-group, or false when nothing is left for its query block
__device__ __forceinline__ bool w4_claim(const MfmaKernelArgs& a, uint32_t strip, uint32_t qbl, uint32_t& own_next, int lane, uint32_t& cb, uint32_t& ce) {
    uint32_t* words = a.claims + qbl * a.nstrips;
    auto chunk = [&](uint32_t s, uint32_t j) {
        uint32_t b, e;
        w4_strip_range(a, s, b, e);
        cb = e - w4_tail_tiles(e - b) + j * kStealChunk;
        ce = cb + kStealChunk < e ? cb + kStealChunk : e;
    };
    {   // own chunks, ascending (a thief may have taken some: it starts from the top)
        uint32_t b, e;
        w4_strip_range(a, strip, b, e);
        const uint32_t cpt = (w4_tail_tiles(e - b) + kStealChunk - 1u) / kStealChunk;
        while (own_next < cpt) {
            const uint32_t j = own_next++;
            uint32_t old = 0u;
            if (lane == 0) old = atomicOr(&words[strip], 1u << j);
            old = __builtin_amdgcn_readfirstlane(old);
            if (!((old >> j) & 1u)) { chunk(strip, j); return true; }
        }
    }
    // steal: lane l looks at strip + 1 + l (one round trip for 64 strips); the first strip with a free chunk gives its highest one
    for (uint32_t base = 1u; base < a.nstrips; base += 64u) {
        const uint32_t v = base + (uint32_t)lane;
        const uint32_t s = (strip + v) % a.nstrips;
        uint32_t free_bits = 0u;
        if (v < a.nstrips) {
            uint32_t b, e;
            w4_strip_range(a, s, b, e);
            const uint32_t cpt = (w4_tail_tiles(e - b) + kStealChunk - 1u) / kStealChunk;
            const uint32_t w = __hip_atomic_load(&words[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            free_bits = ~w & (cpt >= 32u ? 0xFFFFFFFFu : ((1u << cpt) - 1u));
        }
        unsigned long long cand = __ballot(free_bits != 0u);
        while (cand) {
            const int L = __builtin_ctzll(cand);
            uint32_t ok = 0u, jj = 0u;
            if (lane == L) {
                jj = 31u - (uint32_t)__builtin_clz(free_bits);
                const uint32_t old = atomicOr(&words[s], 1u << jj);
                ok = ((old >> jj) & 1u) ? 0u : 1u;
                free_bits &= ~(old | (1u << jj));
            }
            if (__builtin_amdgcn_readlane(ok, L)) {
                chunk(__builtin_amdgcn_readlane(s, L), __builtin_amdgcn_readlane(jj, L));
                return true;
            }
            cand = __ballot(free_bits != 0u);
        }
    }
    return false;
}

// DENSE: 0 = filtered launch, 1 = sample pass writing every score, 2 = sample pass writing group bests (one kernel per
// form: with both sample epilogues in one function hipcc ran out of VGPRs and went into the accumulator file).
template <int METRIC, int DENSE, bool SPLIT>
__global__ __launch_bounds__(256) void scan_mfma_w4_kernel(const MfmaKernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // makes the kernel descriptor allocate a[0:255]
    asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    uint32_t strip, qb0, qb_step;
    if (!wg_assignment(a, strip, qb0, qb_step)) return;
    const uint32_t t0 = a.tile_first + (uint32_t)((uint64_t)a.ntiles * strip / a.nstrips);
    const uint32_t t1 = a.tile_first + (uint32_t)((uint64_t)a.ntiles * (strip + 1) / a.nstrips);
    if (t0 >= t1) return;

    const uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds;   // (for the asm LDS-DMA destinations)
    // this wave's hit log (filtered launches)
    char* wlog_lds = lds + kLdsDump + (uint32_t)wave * kDumpWaveBytes;
    char* region = a.dump + (size_t)(blockIdx.x * 4u + (uint32_t)wave) * kDumpRegionBytes;   // ... and where a full log is spilled to

    const uint32_t KT = a.ld_bytes >> 7;
    const uint32_t st_row = lane >> 3;
    const uint32_t lda_bytes = SPLIT ? a.lda_bytes : a.ld_bytes;
    const uint32_t st_lane_off_a = st_row * lda_bytes + (((lane & 7) ^ st_row) << 4);
    const uint32_t st_lane_off_b = st_row * a.ld_bytes + (((lane & 7) ^ st_row) << 4);
    const uint32_t fr = lane & 15, fg = lane >> 4, r7 = fr & 7;
    const uint32_t a_frag0 = ((wr * 16 + (fr >> 3)) << 10) + (r7 << 7);
    const uint32_t b_frag0 = 32768u + ((wc * 16 + (fr >> 3)) << 10) + (r7 << 7);
    const uint32_t c_off0 = ((0 * 4 + fg) ^ r7) << 4, c_off1 = ((1 * 4 + fg) ^ r7) << 4;
    const uint32_t piece_stride_a = 8u * lda_bytes, piece_stride_b = 8u * a.ld_bytes;   // 8 rows; 31 pieces fit 32 bits
    // The strip's static share [t0, t1s); behind it come claimed chunks (work stealing: filtered launches of long strips,
    // K extents of at least three K-tiles -- the MFMAs then always find the tile they move on to in the staging cursor).
    const uint32_t t1s = (DENSE == 0 && a.claims != nullptr && KT >= 3u) ? t1 - w4_tail_tiles(t1 - t0) : t1;
    uint32_t total_it = 0xFFFFFFFFu;   // K-tiles of this work-group: known once the staging cursor has run out of tiles

    // ONE query block per work-group: a loop over query blocks here would re-enter the prologue with the
    // accumulator file live in hipcc's eyes (it parks kernel-entry values in AGPRs up to the first MFMA);
    // the launcher splits batches of more than `slots` query blocks into several launches instead
    const uint32_t qb = a.qb_base + qb0;
    {
        float* thr_l = reinterpret_cast<float*>(lds + kLdsThr);
        float* qn2_l = reinterpret_cast<float*>(lds + kLdsQn2);
        const float* xn_l = reinterpret_cast<const float*>(lds + kLdsXn2);
        if (tid == 0) lds_zero3(reinterpret_cast<uint32_t*>(lds + kLdsNext));   // claimed range: none; next own chunk: 0
        thr_l[tid] = DENSE ? 0.0f : a.thr[qb * kBN + tid];
        qn2_l[tid] = METRIC == M_L2 ? a.qnorm2[qb * kBN + tid] : 0.0f;
        // (published by the prologue's __syncthreads)
        // per-lane source pointers of the K-tile being staged (two K-tiles ahead of the MFMAs)
        // the K-tile being staged (two K-tiles ahead of the MFMAs): wave-uniform bases, advanced by stage_advance
        const char* ua_src = w4_uniform_ptr(a.corpus + (uint64_t)t0 * kBM * lda_bytes);
        const char* ub_src = w4_uniform_ptr(a.queries + (uint64_t)qb * kBN * a.ld_bytes);
        // this wave's pieces: 4 of the 16 of each unit -- pieces p0 + i + 8 h (i < 4; h = row half of the unit): the lane
        // offsets of the 8 A and 8 B pieces are loop constants, the LDS destinations immediates behind lds_w + the buffer
        const uint32_t p0 = ((uint32_t)wave >> 1) * 16u + ((uint32_t)wave & 1u) * 4u;
        uint32_t voa[8], vob[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            voa[j] = st_lane_off_a + (p0 + (uint32_t)(j & 3) + (uint32_t)(j >> 2) * 8u) * piece_stride_a;
            vob[j] = st_lane_off_b + (p0 + (uint32_t)(j & 3) + (uint32_t)(j >> 2) * 8u) * piece_stride_b;
        }
        const uint32_t lds_w = __builtin_amdgcn_readfirstlane(lds_addr + p0 * 1024u);
        uint32_t st_kt = 0, st_tile = t0, a_kt = 0;   // a_kt: K-tile of the corpus row ua_src points at
        uint32_t st_end = t1s;                        // end of the range of tiles the staging cursor walks
        bool pace_on = true;                          // (thread 0) false after one pacing timeout / behind the static share: no more pacing in this launch
        const uint32_t next_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(lds + kLdsNext);
        // unit A_mh / B_nh = the 16 pieces (8 rows x 128 B each) of rows [h*64, h*64+64) of both
        // 128-row halves; this wave moves 4 of them: idx = wave*4 + i -> piece (idx>>3)*16 + (idx&7) + h*8
        // unit A_mh / B_nh = the 16 pieces (8 rows x 128 B each) of rows [h*64, h*64+64) of both 128-row halves
#define W4_STAGE_A(BUF, H, I) w4_dma_piece<((I) + 8 * (H)) * 1024>(lds_w + ((BUF) & 1u) * kStageBytes, voa[(I) + 4 * (H)], ua_src);
#define W4_STAGE_B(BUF, H, I) w4_dma_piece<32768 + ((I) + 8 * (H)) * 1024>(lds_w + ((BUF) & 1u) * kStageBytes, vob[(I) + 4 * (H)], ub_src);
        // next K-tile of the strip (clamped at its end: the last K-tile is re-staged, never read).  The common case -- the
        // next K-tile of the same tile -- is two scalar adds behind one compare; the tile boundary is a wave-uniform branch.
        // (The bases are SGPR pairs now: with per-lane pointers this had to be written with selects, hipcc moved branchy
        // pointer updates into a scratch array.)
        auto stage_advance = [&](int32_t it_adv /* the iteration this advance belongs to; -2, -1 in the prologue */) {
            if (st_kt + 1 < KT) {
                // SPLIT: K-tiles 3j, 3j+1, 3j+2 read corpus K-tile 2j, 2j, 2j+1 ([hi_j | lo_j] interleaved: the
                // hi plane is staged twice in a row, the second time from L2); a_kt = corpus K-tile
                const bool hold = SPLIT && (st_kt % 3u) == 0u;
                if constexpr (SPLIT) a_kt += hold ? 0u : 1u;
                ua_src += hold ? 0 : 128;
                ub_src += 128;
                ++st_kt;
            } else {
                // tile boundary: the next tile of the range in hand, or -- the range is through -- the first of the range
                // wave 0 claimed a tile ago ({begin, end} in LDS; begin = end: none), or nothing: the cursor stays (the last
                // K-tile is re-staged, never read) and the loop's end is known
                uint32_t nt = st_tile + 1u;
                bool have = nt < st_end;
                if (DENSE == 0 && !have) {
                    const uint2 nx = *reinterpret_cast<const uint2*>(lds + kLdsNext);
                    if (nx.y > nx.x) { nt = nx.x; st_end = nx.y; have = true; pace_on = false; }   // (behind its static share a work-group no longer paces)
                }
                if (have) {
                    const int64_t a_back = SPLIT ? (int64_t)a_kt * 128 : (int64_t)(KT - 1) * 128;
                    ua_src += (int64_t)(int32_t)(nt - st_tile) * ((int64_t)kBM * lda_bytes) - a_back;
                    ub_src -= (int64_t)(KT - 1) * 128;
                    if constexpr (SPLIT) a_kt = 0u;
                    st_kt = 0u;
                    st_tile = nt;
                    // The cursor has entered the LAST tile of the range in hand: wave 0 claims the range that follows and leaves
                    // it in LDS for everybody's advance out of this tile, KT K-tiles from now (their barriers order the write
                    // before the reads).  The third word is wave 0's own bookkeeping: its next own chunk.
                    if (DENSE == 0 && nt + 1u == st_end && wave == 0 && t1s != t1) {
                        uint32_t cb = 0u, ce = 0u;
                        uint32_t own_next = *reinterpret_cast<const uint32_t*>(lds + kLdsNext + 8);
                        (void)w4_claim(a, strip, qb0, own_next, lane, cb, ce);
                        if (lane == 0)
                            asm volatile("ds_write_b64 %0, %1\n\tds_write_b32 %0, %2 offset:8\n\ts_waitcnt lgkmcnt(0)"
                                         :: "v"(next_addr), "v"(((uint64_t)ce << 32) | cb), "v"(own_next) : "memory");
                    }
                } else if (total_it == 0xFFFFFFFFu) {
                    total_it = (uint32_t)(it_adv + 3);   // the cursor stands at K-tile it_adv + 2, the last one
                }
            }
            // (behind the join hipcc would keep the two bases in VGPRs: an "s" asm operand then prints as v[..])
            ua_src = w4_uniform_ptr(ua_src);
            ub_src = w4_uniform_ptr(ub_src);
        };

        // ---- prologue: K-tiles 0 and 1 whole, landed; fragments A0, B0 of K-tile 0
#define W4_STAGE_UNIT(B) W4_STAGE_A(B, 0, 0) W4_STAGE_B(B, 0, 0) W4_STAGE_A(B, 0, 1) W4_STAGE_B(B, 0, 1) W4_STAGE_A(B, 0, 2) W4_STAGE_B(B, 0, 2)   \
                         W4_STAGE_A(B, 0, 3) W4_STAGE_B(B, 0, 3) W4_STAGE_A(B, 1, 0) W4_STAGE_B(B, 1, 0) W4_STAGE_A(B, 1, 1) W4_STAGE_B(B, 1, 1)   \
                         W4_STAGE_A(B, 1, 2) W4_STAGE_B(B, 1, 2) W4_STAGE_A(B, 1, 3) W4_STAGE_B(B, 1, 3)
        W4_STAGE_UNIT(0u)
        stage_advance(-2);
        W4_STAGE_UNIT(1u)
        stage_advance(-1);
#undef W4_STAGE_UNIT
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        bf16x8 FA0[4][2], FA1[4][2], FBx[4][2], FBy[4][2];
#define W4_LOAD_A1(FA, MH, L, J) FA[(J) >> 1][(J) & 1] = *reinterpret_cast<const bf16x8*>((L) + a_frag0 + ((MH) * 4 + ((J) >> 1)) * 2048 + (((J) & 1) ? c_off1 : c_off0));
#define W4_LOAD_B1(FB, NH, L, J) FB[(J) >> 1][(J) & 1] = *reinterpret_cast<const bf16x8*>((L) + b_frag0 + ((NH) * 4 + ((J) >> 1)) * 2048 + (((J) & 1) ? c_off1 : c_off0));
// (Where in a phase the four DMA pieces go does not matter: all behind the fragment reads +0.2 %, all in front of
// them -1.1 %, alternating 0.0 % -- profiles/r02/mfma_experiments.md section 8.)
// DM(j, 0): M0 of the phase's j-th piece, DM(j, 1): its load -- a group of MFMAs apart
// (the explicit lgkmcnt(0) -- s_waitcnt 0xC07F: vmcnt / expcnt untouched -- in front: the phase's fragments were read a
//  phase ago and have landed; without it hipcc meters them in with four or five counted waits between the first MFMAs)
#define W4_PHASE_S(FA, FB, MH, NH, ZERO, LD, DM)                                                   \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                            \
    DM(0, 0) w4_mfma_group<MH, NH, ZERO, 0>(FA, FB); LD(0) LD(1) DM(0, 1)                          \
    w4_mfma_group<MH, NH, ZERO, 1>(FA, FB); LD(2) LD(3) DM(1, 0)                                   \
    w4_mfma_group<MH, NH, ZERO, 2>(FA, FB); LD(4) LD(5) DM(1, 1)                                   \
    w4_mfma_group<MH, NH, ZERO, 3>(FA, FB); LD(6) LD(7) DM(2, 0)                                   \
    w4_mfma_group<MH, NH, ZERO, 4>(FA, FB); DM(2, 1)                                               \
    w4_mfma_group<MH, NH, ZERO, 5>(FA, FB); DM(3, 0)                                               \
    w4_mfma_group<MH, NH, ZERO, 6>(FA, FB); DM(3, 1)                                               \
    w4_mfma_group<MH, NH, ZERO, 7>(FA, FB);
// a K-tile's four phases with its middle wait + barrier; ZERO: the tile's first K-tile (C = 0 form of the kk = 0 MFMAs)
#define W4_KTILE(BX, BY, LDQ0, LDQ3, ZERO)                                                         \
    W4_PHASE_S(FA0, BX, 0, 0, ZERO, LDQ0, W4_DMU0)                                                 \
    W4_PHASE_S(FA0, BY, 0, 1, ZERO, W4_LDQ1, W4_DMU1)                                              \
    asm volatile(W4_VMWAIT ::: "memory");                                                          \
    W4_LOOP_BARRIER();                                                                             \
    W4_PHASE_S(FA1, BY, 1, 1, ZERO, W4_LDQ2, W4_DMU2)                                              \
    W4_PHASE_S(FA1, BX, 1, 0, ZERO, LDQ3, W4_DMU3)
// Ablation builds (timing only, results wrong; scripts/ubench/lib_ab): -DVROD_W4_ABL_NODMA no LDS-DMA piece in the loop
// (the two K-tiles of the prologue stay in LDS), -DVROD_W4_ABL_NOBAR no barrier in the loop, -DVROD_W4_ABL_NOEPI no tile epilogue
#ifdef VROD_W4_ABL_NODMA
#define W4_DMU0(j, L)
#define W4_DMU1(j, L)
#define W4_DMU2(j, L)
#define W4_DMU3(j, L)
#define W4_VMWAIT "s_waitcnt lgkmcnt(0)"
#else
#define W4_HALF_A(H, I, LOAD) if constexpr (LOAD) w4_dma_load(voa[(I) + 4 * (H)], ua_src); else w4_dma_m0<((I) + 8 * (H)) * 1024>(lds_wp);
#define W4_HALF_B(H, I, LOAD) if constexpr (LOAD) w4_dma_load(vob[(I) + 4 * (H)], ub_src); else w4_dma_m0<32768 + ((I) + 8 * (H)) * 1024>(lds_wp);
#define W4_DMU0(j, L) W4_HALF_A(0, j, L)
#define W4_DMU1(j, L) W4_HALF_B(0, j, L)
#define W4_DMU2(j, L) W4_HALF_B(1, j, L)
#define W4_DMU3(j, L) W4_HALF_A(1, j, L)
#define W4_VMWAIT "s_waitcnt vmcnt(16) lgkmcnt(0)"
#endif
#ifdef VROD_W4_ABL_NOEPI
#define W4_ABL_EPI false
#else
#define W4_ABL_EPI true
#endif
#ifdef VROD_W4_ABL_NOBAR
#define W4_LOOP_BARRIER()
#else
#define W4_LOOP_BARRIER() VROD_BARRIER()
#endif
// PAR = it & 1, a constant of each of the loop's two copies of this body (the loop starts at it = 0): the LDS addresses of
// the fragment reads and of the DMA destinations are then loop constants (8 vector adds fewer per K-tile)
#define W4_ITER(BX, BY, LDQ0, LDQ3, PAR)                                                           \
    {                                                                                              \
        const char* l = lds + (PAR) * kStageBytes;                                                 \
        const char* ln = lds + (1 - (PAR)) * kStageBytes;                                          \
        const uint32_t lds_wp = lds_w + (PAR) * kStageBytes;                                       \
        const bool first = kt == 0;                                                                \
        /* pacing of the sibling work-groups (see the phased kernel), here in units of K-tiles: with  \
           the staging two K-tiles ahead no work-group ever waits for HBM, so nothing else keeps   \
           the siblings of a strip together */                                                     \
        /* (a countdown, not it % pace_every: one wave per SIMD issues one instruction per ~4 cycles, and the   \
            ~25 scalar instructions of a run-time modulo in front of every K-tile were 3 % of the kernel) */   \
        if (--pace_left == 0u) {                                                                   \
          pace_left = a.pace_every;                                                                \
          ++pace_round;                                                                            \
          if (pace_on && tid == 0) {                                                               \
            uint32_t* ctr = a.pace + strip;                                                        \
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);           \
            const uint32_t want = a.nqb * pace_round;                                              \
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
          }                                                                                        \
        }                                                                                          \
        /* L2: the tile's 256 row norms -> LDS slot of its parity (one 1-KB piece, wave 0) */      \
        if (METRIC == M_L2 && DENSE != 1 && first && wave == 0)                                    \
            w4_dma_piece<0>(lds_addr + kLdsXn2 + (tile & 1u) * 1024u, (uint32_t)lane * 16u, w4_uniform_ptr(reinterpret_cast<const char*>(a.xnorm2 + (uint64_t)tile * kBM))); \
        if (first) { W4_KTILE(BX, BY, LDQ0, LDQ3, true) } else { W4_KTILE(BX, BY, LDQ0, LDQ3, false) } \
        stage_advance((int32_t)it);                                                                \
        const bool last = kt == KT - 1;                                                            \
        W4_CLK(uint32_t ce1 = 0;)                                                                  \
        if (last) {                                                                                \
            W4_CLK(const uint32_t ce0 = (uint32_t)__builtin_amdgcn_s_memtime();)                   \
            if constexpr (DENSE == 2)                                                              \
                w4_groupmax_store_tile<METRIC>(a, qn2_l, xn_l + (tile & 1) * 256 + wr * 128 + fg * 4, wc * 128 + fr, \
                                               qb * kBN + wc * 128 + fr, ((tile - a.tile_first) * 2 + wr) * 4 + fg); \
            else if constexpr (DENSE == 1)                                                         \
                w4_dense_store_tile<METRIC>(a, qn2_l, wc * 128 + fr, tile * kBM + wr * 128 + fg * 4, qb * kBN + wc * 128 + fr); \
            else if (W4_ABL_EPI)                                                                   \
                w4_filter_tile<METRIC>(a, thr_l, qn2_l, xn_l + (tile & 1) * 256 + wr * 128 + fg * 4,        \
                                       tile, wc * 128 + fr, qb, wlog_lds, region, wr, wc, lane, wlog, wglob ); \
            W4_CLK(ce1 = (uint32_t)__builtin_amdgcn_s_memtime(); cke += ce1 - ce0; ckn += 1;)     \
            kt = 0;                                                                                \
            tile = KT >= 3u ? st_tile : tile + 1u;   /* the cursor is two K-tiles ahead: in the tile the MFMAs move on to */ \
        } else ++kt;                                                                               \
        asm volatile(W4_VMWAIT ::: "memory");                                                      \
        W4_LOOP_BARRIER();                                                                         \
        W4_CLK(if (last) ckb += (uint32_t)__builtin_amdgcn_s_memtime() - ce1;)                     \
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
        uint32_t wlog = 0u, wglob = 0u;   // entries in this wave's hit log / in its spill region (wave-uniform)
        // K-tiles to the next pacing point (wave-uniform countdown; pace_every = 0: starts at 0 and wraps, i.e. never) and how many passed
        uint32_t pace_left = a.pace_every ? a.pace_every + 1u : 0u, pace_round = 0u;
        
        W4_CLK(const uint64_t ck0 = __builtin_amdgcn_s_memtime(); const uint64_t cr0 = __builtin_amdgcn_s_memrealtime(); uint32_t cke = 0, ckb = 0, ckn = 0;)
        while (it < total_it) {
            W4_ITER(FBx, FBy, W4_LDQ0x, W4_LDQ3x, 0)
            if (it >= total_it) break;
            W4_ITER(FBy, FBx, W4_LDQ0y, W4_LDQ3y, 1)
        }
        
        W4_CLK(if (!DENSE && tid == 0) {
            const uint64_t ck1 = __builtin_amdgcn_s_memtime(), cr1 = __builtin_amdgcn_s_memrealtime();
            atomicAdd(&g_w4_clk[0], (unsigned long long)(ck1 - ck0)); atomicAdd(&g_w4_clk[1], (unsigned long long)(cr1 - cr0));
            atomicAdd(&g_w4_clk[2], 1ull); atomicAdd(&g_w4_clk[3], (unsigned long long)total_it);
            atomicAdd(&g_w4_clk[4], (unsigned long long)cke); atomicAdd(&g_w4_clk[5], (unsigned long long)ckb); atomicAdd(&g_w4_clk[6], (unsigned long long)ckn);
        })
#undef W4_LOAD_A1
#undef W4_LOAD_B1
#undef W4_PHASE_S
#undef W4_KTILE
#undef W4_VMWAIT
#undef W4_DMU0
#undef W4_DMU1
#undef W4_DMU2
#undef W4_DMU3
#undef W4_STAGE_A
#undef W4_STAGE_B
#undef W4_HALF_A
#undef W4_HALF_B
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
        
        if constexpr (DENSE == 0) {   // every wave for itself: no barrier
            if (wlog) w4_spill<METRIC>(a, thr_l, qn2_l, wlog_lds, wlog, region, wglob, qb, wr, wc, lane);
            if (wglob) w4_process_region<METRIC>(a, thr_l, qn2_l, region, wglob, qb, wr, wc, lane);
        }
        
        
    }
}

// dense_form: 0 = filtered launch, 1 = sample pass writing every score, 2 = sample pass writing group bests
void launch_mfma_w4(const MfmaKernelArgs& a, int metric, int dense_form, bool split, int grid, hipStream_t s, hipEvent_t start, hipEvent_t stop) {
#define VROD_MFMA_W4K_(MM, DN, SP)                                                                          \
    do {                                                                                                    \
        static bool attr_set = false;                                                                       \
        if (!attr_set) {                                                                                    \
            (void)hipFuncSetAttribute((const void*)scan_mfma_w4_kernel<MM, DN, SP>,                         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotalW4);             \
            attr_set = true;                                                                                \
        }                                                                                                   \
        hipExtLaunchKernelGGL((scan_mfma_w4_kernel<MM, DN, SP>), dim3(grid), dim3(256), kLdsTotalW4, s, start, stop, 0, a); \
    } while (0)
#define VROD_MFMA_W4K(MM, DN) do { if (split) VROD_MFMA_W4K_(MM, DN, true); else VROD_MFMA_W4K_(MM, DN, false); } while (0)
#define VROD_MFMA_W4M(MM) do { if (dense_form == 0) VROD_MFMA_W4K(MM, 0); else if (dense_form == 2) VROD_MFMA_W4K(MM, 2); else VROD_MFMA_W4K(MM, 1); } while (0)
    if (metric == M_COSINE) VROD_MFMA_W4M(M_COSINE); else VROD_MFMA_W4M(M_L2);
#undef VROD_MFMA_W4M
#undef VROD_MFMA_W4K
#undef VROD_MFMA_W4K_
}

}  // namespace vrod


#ifdef VROD_W4_CLK
extern "C" int vrod_debug_w4_clk(unsigned long long* out8, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(vrod::g_w4_clk), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(vrod::g_w4_clk), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif
