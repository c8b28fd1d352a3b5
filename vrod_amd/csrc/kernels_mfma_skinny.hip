// kernels_mfma_skinny.hip -- batches of 5..64 queries over bf16 rows (or the [hi | lo] planes of fp32 rows):
// the MFMA scan built around HBM like the stream scan.  Family overview: kernels_mfma.hip.
#include "mfma_common.h"

namespace vrod {

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
// of the split pass): what fits in LDS beside the wave logs.  0: none (VROD_DEBUG_SKINNY=0, long rows).
uint32_t mfma_skinny_max_queries(bool split, uint32_t row_bytes) {
    if (!debug_env().skinny) return 0;
    const uint32_t lds_cap = 160u * 1024u;
    if (!split) return skinny_lds_bytes<4>(row_bytes) <= lds_cap ? 64u : skinny_lds_bytes<2>(row_bytes) <= lds_cap ? 32u : 0u;
    return skinny_lds_bytes<2>(row_bytes) <= lds_cap ? 32u : skinny_lds_bytes<1>(row_bytes) <= lds_cap ? 16u : 0u;
}

void launch_mfma_skinny(const MfmaKernelArgs& a, int metric, bool dense, bool split, uint32_t nq, int num_cus, hipStream_t s,
                        hipEvent_t start, hipEvent_t stop) {
    // rows in LDS: the K extent of a corpus row ([hi | lo] planes in the split form)
    const uint32_t qrow = split ? a.lda_bytes : a.ld_bytes;
    const uint32_t lds_cap = 160u * 1024u;
    int nt = 0;
    if (!split) nt = (nq > 32) ? 4 : 2;
    else nt = (nq > 16 || skinny_lds_bytes<1>(qrow) > lds_cap) ? 2 : 1;
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
        hipExtLaunchKernelGGL((scan_mfma_skinny_kernel<MM, DN, NN, SP>), dim3(sgrid), dim3(kSkWaves * 64), lds_bytes, s, start, stop, 0, a); \
    } while (0)
#define VROD_MFMA_SKN(MM, DN)                                                                               \
    do {                                                                                                    \
        if (split) { if (nt == 2) VROD_MFMA_SK(MM, DN, 2, true); else VROD_MFMA_SK(MM, DN, 1, true); }      \
        else { if (nt == 4) VROD_MFMA_SK(MM, DN, 4, false); else VROD_MFMA_SK(MM, DN, 2, false); }          \
    } while (0)
    if (metric == M_COSINE) { if (dense) VROD_MFMA_SKN(M_COSINE, true); else VROD_MFMA_SKN(M_COSINE, false); }
    else { if (dense) VROD_MFMA_SKN(M_L2, true); else VROD_MFMA_SKN(M_L2, false); }
#undef VROD_MFMA_SKN
#undef VROD_MFMA_SK
}

}  // namespace vrod
