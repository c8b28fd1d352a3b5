// kernels_mfma.hip -- batched Q.K^T distance scan on the gfx950 matrix cores with a fused
// per-query threshold filter (the score matrix is never written to memory): the dispatcher.
//
// Fills the batched half of the scan slot under SearchSimilarCommand::execute (reference
// src/command/types.rs:121-132, empty).  GEMM view: M = corpus rows, N = queries,
// K = vector dimension; both operands are stored [row][k] with k contiguous.
//
// The family (one translation unit each, so that an edit of one schedule rebuilds only that one):
//   kernels_mfma_w4.hip      scan_mfma_w4_kernel<METRIC, DENSE, SPLIT>   bf16 rows, and the [hi | lo] bf16 planes of
//                            fp32 rows (SPLIT): 4 waves, one per SIMD, 128 x 128 per wave, accumulators a[0:255]
//                            owned by inline asm (audited in the build).  The default of every batch of > 64 queries.
//   kernels_mfma_skinny.hip  scan_mfma_skinny_kernel<...>                batches of 5..64 queries: HBM-bound form
//   kernels_mfma_phased.hip  scan_mfma_phased_kernel<float, ...>         fp32 rows without planes: 8 waves, exact fp32
//                            matrix-core pass (v_mfma_f32_16x16x4_f32)
// Every launch is a pure function of its inputs: thresholds are read-only inside a launch and refreshed between
// launches by list_compact_kernel (kernels_select.hip).
//
// Roofline: MFMA.  Algorithmic flops per launch = 2 * nq * rows * dim  (SURVEY.md 8d).
#include "mfma_common.h"

namespace vrod {

// does the dense (sample) form of this launch run on the 4-wave kernel, the one form that can write group bests?
static bool mfma_takes_w4(const MfmaScanArgs& h, int dtype) {
    const bool split = h.a_wrap != 0;
    if (dtype != DT_BF16) return false;
    const uint32_t ld_bytes = h.ld * 2u, qrow = split ? (h.lda_bytes ? h.lda_bytes : ld_bytes) : ld_bytes;
    return !(h.nq > 0 && h.nq <= mfma_skinny_max_queries(split, qrow));   // else the skinny kernel takes it
}
uint32_t mfma_dense_group_rows(const MfmaScanArgs& h, int dtype) { return mfma_takes_w4(h, dtype) ? 32u : 0u; }

static int mfma_grid(int num_cus) { const int g = num_cus / 8 * 8; return g < 8 ? 8 : g; }
// the 4-wave kernel's spill regions: one per wave of the grid
size_t mfma_dump_bytes(int num_cus) { return (size_t)mfma_grid(num_cus) * 4u * kDumpRegionBytes; }

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
    a.pace = h.pace;
    const int grid = mfma_grid(num_cus);
    a.dump = (char*)h.dump;
    a.slots = grid / 8;
    const bool split = h.a_wrap != 0;

    if (dtype == DT_BF16 && !mfma_takes_w4(h, dtype)) {
        launch_mfma_skinny(a, h.metric, h.dense_out != nullptr, split, h.nq, num_cus, s, lev.start, lev.stop);
        return;
    }
    if (dtype == DT_BF16) {
        // sibling pacing interval of the 4-wave kernel, in K-tiles (0 = off; VROD_DEBUG_PACE_KT for A/B runs)
        const int pace_kt = debug_env().pace_kt;
        const uint32_t nqb_total = a.nqb;
        for (uint32_t qb_base = 0; qb_base < nqb_total; qb_base += a.slots) {   // one launch unless nq > 256 * slots
            a.qb_base = qb_base;
            a.nqb = std::min<uint32_t>(a.slots, nqb_total - qb_base);
            const bool first_launch = qb_base == 0, last_launch = qb_base + a.slots >= nqb_total;
            a.strips_per_xcd = a.slots / a.nqb;
            a.nstrips = 8 * a.strips_per_xcd;
            a.pace_every = (h.pace && a.nqb > 1 && pace_kt > 0) ? (uint32_t)pace_kt : 0u;   // K-tiles
            if (a.pace_every && (qb_base > 0 || !h.pace_is_zero)) (void)hipMemsetAsync(a.pace, 0, a.nstrips * sizeof(uint32_t), s);
            const int form = !h.dense_out ? 0 : a.dense_group ? 2 : 1;
            // work stealing (kernels_mfma_w4.hip): the claim bits of this launch start at zero.  VROD_DEBUG_W4_STEAL=0: static shares (A/B runs)
            const bool steal_on = debug_env().w4_steal;
            a.claims = (form == 0 && steal_on && h.claims && (size_t)a.nqb * a.nstrips <= kMfmaClaimWords) ? h.claims : nullptr;
            if (a.claims && (qb_base > 0 || !h.claims_is_zero)) (void)hipMemsetAsync(a.claims, 0, (size_t)a.nqb * a.nstrips * sizeof(uint32_t), s);
            launch_mfma_w4(a, h.metric, form, split, grid, s, first_launch ? lev.start : nullptr, last_launch ? lev.stop : nullptr);
        }
        return;
    }
    // fp32 rows: the 8-wave phased kernel; pacing in tiles, only where several work-groups share a strip
    const int pace_tiles = debug_env().pace_tiles;
    a.strips_per_xcd = a.nqb <= a.slots ? a.slots / a.nqb : 1;
    a.nstrips = 8 * a.strips_per_xcd;
    a.pace_every = (h.pace && a.nqb > 1 && a.nqb <= a.slots && pace_tiles > 0) ? (uint32_t)pace_tiles : 0u;
    if (a.pace_every && !h.pace_is_zero) (void)hipMemsetAsync(a.pace, 0, a.nstrips * sizeof(uint32_t), s);
    launch_mfma_phased_f32(a, h.metric, h.dense_out != nullptr, grid, s, lev.start, lev.stop);
}

}  // namespace vrod
