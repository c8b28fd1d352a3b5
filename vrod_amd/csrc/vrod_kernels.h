// vrod_kernels.h -- host-callable launchers of the HIP kernels (internal to libvrod_hip).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

namespace vrod {

// How a shard's local row index becomes the id the caller sees.  Single-device index: row + offset.
// Shard `shard` of a multi-device handle (rows dealt to `n_shards` shards in blocks of `block_rows`:
// global r -> shard (r / B) % G, local (r / (B*G)) * B + r % B):
//   id = ((row / B) * G + shard) * B + row % B + offset
// -- monotonic within a shard, so a list sorted by (score, local row) stays sorted by (score, id).
struct IdMap {
    uint64_t offset = 0;
    uint32_t block_rows = 0;   // 0: identity mapping
    uint32_t shard = 0, n_shards = 1;
    __host__ __device__ inline uint64_t operator()(uint32_t row) const {
        if (block_rows == 0) return (uint64_t)row + offset;
        const uint64_t b = row / block_rows, r = row % block_rows;
        return (b * n_shards + shard) * block_rows + r + offset;
    }
};

// ---- kernels_prep.hip
void launch_synth_rows(uint64_t seed, uint64_t first_row, uint64_t n, uint32_t dim, float* d_out,
                       hipStream_t s);
void launch_prepare_rows(const float* d_in, uint64_t n, uint32_t dim, uint32_t ld, int metric,
                         int dtype, double* d_nrm_ws, uint32_t* d_bad_flag, float* d_out_f32,
                         void* d_out_bf16, hipStream_t s);
// Queries: normalise (COSINE) / round (BF16) / zero-pad to [nq_pad][ld] in ONE launch; also the
// fast squared norms qn2[nq_pad], the NaN/Inf flag and the max squared norm (float bits).
// `init`: per-search state the same launch resets (block q: status[q], counts[q], thr[q];
// block 0: n_zero_words words at zero_words) -- replaces five memset launches.
struct QueryInit {
    uint32_t* status;        // [nq_pad] -> 0
    uint32_t* counts;        // [nq_pad] -> 0 (may be null)
    float* thr;              // [nq_pad] -> thr_live_bits for q < nq, thr_pad_bits beyond (may be null)
    uint32_t thr_live_bits, thr_pad_bits;
    uint32_t* zero_words;    // scalars to clear (may be null)
    uint32_t n_zero_words;
    uint32_t* zero_words2;   // a second range (the pacing counters of the coming scan launches)
    uint32_t n_zero_words2;
};
// bf16 split of fp32 rows: corpus rows -> [hi | lo] (2*ldp), query rows -> [hi | hi | lo] (3*ldp)
void launch_split_rows(const float* d_in, uint64_t n, uint32_t ld, uint32_t ldp, void* d_out, bool queries, hipStream_t s);
void launch_prep_queries(const float* d_in, uint32_t nq, uint32_t nq_pad, uint32_t dim, uint32_t ld, int metric,
                         int dtype, float* d_out_f32, void* d_out_bf16, float* d_qn2, uint32_t* d_bad_flag,
                         uint32_t* d_max_bits, const QueryInit& init, hipStream_t s);
// tail of a search: status[nq] followed by {bad flag, max |q|^2 bits, max err bits, max |x|^2 bits}
// gathered into one contiguous block so that the host needs ONE device-to-host copy
void launch_gather_readback(const uint32_t* d_status, uint32_t nq, uint32_t* d_flags3, const uint32_t* d_max_xn2,
                            uint32_t* d_out, hipStream_t s);
void launch_row_fastnorm(const void* d_rows, int dtype, uint64_t n, uint32_t ld, float* d_xn2,
                         uint32_t* d_max_bits, hipStream_t s);
void launch_rows_get(const void* d_rows, int dtype, uint64_t n, uint32_t dim, uint32_t ld,
                     float* d_out, hipStream_t s);

// ---- kernels_stream.hip : Q <= 8 HBM-bound scan, writes fast scores [nq_pad][score_ld]
// nq_pad in {1,2,4,8}; d_q is [nq_pad][ld] prepared fp32 (zero rows for padding).
// Also accumulates, per query, a histogram of the top stream_hist_bits(nq_pad) bits of the
// score keys into d_hist [nq_pad][1 << bits] (must be zeroed by the caller): only the bins at
// or above each block's own kp-th best are published, which is all the global select needs.
void launch_scan_stream(const void* d_corpus, int dtype, int metric, uint32_t ld, uint64_t nrows,
                        const float* d_q, int nq_pad, float* d_scores, uint64_t score_ld,
                        uint32_t* d_hist, uint32_t kp, hipStream_t s);
int stream_hist_bits(int nq_pad);
int stream_max_queries_per_pass(uint32_t ld);   // 8 .. 1 by LDS capacity, 0: the row does not fit
// Radix-select step 2: find the histogram bin holding the kp-th best score, then compact
// every row whose key falls in that bin or a better one into d_keys[q][...] (composite keys,
// unordered), counting into d_cnt[q].  More than `cap` such rows -> d_status[q] bit 1.
void launch_hist_compact(const float* d_scores, uint64_t score_ld, uint64_t n, int nq, int metric,
                         const uint32_t* d_hist, int hist_bits, uint32_t kp, uint64_t* d_keys,
                         uint32_t cap, uint32_t* d_cnt, uint32_t* d_status, hipStream_t s);

// ---- kernels_select.hip
// Level 0: fast scores (implicit ids = column index) -> per-chunk top-kp composite keys.
// Later levels: keys -> keys.  Returns the number of keys per query written to d_out.
// chunk capacity is kSelectChunk; kp <= kSelectChunk/2.
constexpr uint32_t kSelectChunk = 8192;
uint64_t launch_select_from_scores(const float* d_scores, uint64_t score_ld, uint64_t n, int nq,
                                   int metric, uint32_t kp, uint64_t* d_out, uint64_t out_ld,
                                   hipStream_t s);
uint64_t launch_select_from_keys(const uint64_t* d_in, uint64_t in_ld, uint64_t n, int nq,
                                 uint32_t kp, uint64_t* d_out, uint64_t out_ld, hipStream_t s);
// Final step of a select chain: keys (n <= kSelectChunk per query, sorted or not) ->
// candidate rows [nq][kp] (sorted best first, ~0u padding) and T[q] = fast score of the
// kp-th candidate (worst score if fewer than kp candidates exist: nothing was left out).
// d_cnt != nullptr: the number of keys of query q is min(d_cnt[q], n) (device-side count).
void launch_keys_to_candidates(const uint64_t* d_keys, uint64_t key_ld, uint64_t n, int nq,
                               int metric, uint32_t kp, uint32_t* d_cand_rows, float* d_cand_fast,
                               float* d_T, const uint32_t* d_cnt, hipStream_t s);

// MFMA path: per-query candidate lists {fast score bits, row} appended by the scan kernel.
// Sort each list, keep the best `keep`, write thr[q] = keep-th fast score (worst if short),
// flag overflow (count > cap) in d_status[q] bit 1.
// A list shorter than `keep` leaves thr[q] unchanged (rows were still left out at that threshold).
// d_cand_rows != nullptr (last compaction of a search): also emit candidates [nq][keep]
// (rows, fast scores; ~0u / NaN padding) and T[q] = the threshold every left-out row fails.
void launch_list_compact(uint2* d_lists, uint32_t* d_counts, uint32_t cap, int nq, int metric,
                         uint32_t keep, float* d_thr, uint32_t* d_status, uint32_t* d_cand_rows,
                         float* d_cand_fast, float* d_T, hipStream_t s);
// Sample pass: thr[q] = the j-th best of the n_sample fast scores scores[q][0..n_sample)
// (exact, 3-pass radix select on the order-preserving key).  One block per query.
void launch_sample_select(const float* d_scores, uint64_t score_ld, uint32_t n_sample, int nq,
                          int metric, uint32_t j, float* d_thr, hipStream_t s);
// Final: canonical scores of the kp candidates -> sorted top-k (ids u64 = idmap(row)),
// certificate per query in d_status bit 0 (1 = NOT certified).
// The certificate's bound on |fast - canonical| is formed on the device from the max squared
// norms (float bits) of the queries and of the corpus:  eps_mode 0: c * |q| * |x|  (dot paths),
// 1: c relative to T (direct L2),  2: c * (|q| + |x|)^2 (L2 through the norm expansion).
void launch_final_topk(const uint32_t* d_cand_rows, const float* d_cand_fast,
                       const float* d_cand_canon, const float* d_T, int nq, uint32_t kp, uint32_t k,
                       int metric, const IdMap& idmap, int eps_mode, float eps_c,
                       const uint32_t* d_max_qn2_bits, const uint32_t* d_max_xn2_bits,
                       uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_status,
                       float* d_max_err, hipStream_t s);

// Band pass (second chance of queries whose certificate failed; kernels_select.hip): thresholds c_k -/+ eps and
// reset counters for the nf failed queries (original indices d_qidx), their prepared rows gathered into a dense
// block, and the scatter of the resolved queries' results back to the caller's rows.
void launch_band_prepare(const uint32_t* d_qidx, uint32_t nf, uint32_t nf_pad, const float* d_out_scores, uint32_t k, int metric,
                         int eps_mode, float eps_c, const uint32_t* d_max_qn2_bits, const uint32_t* d_max_xn2_bits,
                         const float* d_qn2_all, float* d_thr, float* d_qn2, uint32_t* d_counts, uint32_t* d_ok, hipStream_t s);
void launch_gather_query_rows(const float* d_src, const uint32_t* d_qidx, uint32_t nf, uint32_t nf_pad, uint32_t ld, float* d_dst_f32,
                              void* d_dst_bf16, hipStream_t s);
void launch_scatter_results(const uint64_t* d_ids, const float* d_scores, const uint32_t* d_qidx, const uint32_t* d_resolved, uint32_t nf,
                            uint32_t k, uint64_t* d_out_ids, float* d_out_scores, hipStream_t s);

// Exact path: canonical scores (implicit ids) -> exact top-k, via the same select chain with
// composite keys (ties -> smaller id).  Writes one query's output row.
void launch_keys_to_output(const uint64_t* d_keys, uint64_t n, int metric, uint32_t k,
                           const IdMap& idmap, uint64_t* d_out_ids, float* d_out_scores,
                           hipStream_t s);

// Fill a result block with "no result" (ids = UINT64_MAX, scores = NaN): the empty list slots of
// the multi-device exchange.
void launch_fill_none(uint64_t* d_ids, float* d_scores, uint64_t n, hipStream_t s);
// Merge n_lists per-shard results into one. List l's ids start at d_ids + l*list_stride_ids
// (elements), its scores at d_scores + l*list_stride_scores; each is [nq][k].
void launch_merge_topk(int metric, const uint64_t* d_ids, const float* d_scores, uint64_t list_stride_ids,
                       uint64_t list_stride_scores, uint32_t n_lists, uint32_t nq, uint32_t k,
                       uint64_t* d_out_ids, float* d_out_scores, hipStream_t s);

// ---- kernels_rescore.hip : canonical (oracle-order) scores
// d_q: [nq][ld] prepared fp32.  Candidates: rows [nq][kp] (~0u = empty slot -> NaN score).
void launch_rescore_candidates(const void* d_corpus, int dtype, int metric, uint32_t dim,
                               uint32_t ld, const float* d_q, int nq, const uint32_t* d_cand_rows,
                               uint32_t kp, float* d_out, hipStream_t s);
// All rows of the shard for nq in {1, 2, 4, 8} queries in ONE pass over the corpus (the exact
// path): query n is row query_index[n] of d_q; out[n * out_ld + row].  nq must not exceed
// rescore_all_max_queries(ld) (LDS: nq query rows beside the staging tile).
int rescore_all_max_queries(uint32_t ld);
void launch_rescore_all(const void* d_corpus, int dtype, int metric, uint32_t dim, uint32_t ld,
                        const float* d_q, const uint32_t* query_index, int nq, uint64_t nrows,
                        float* d_out, uint64_t out_ld, hipStream_t s);

// ---- kernels_mfma.hip : batched Q.K^T scan with fused threshold filter
// Timing of the dominant scan launches without marker packets: the launcher of the next scan
// kernel attaches these events to the dispatch itself (hipExtLaunchKernelGGL), then clears them.
// A marker recorded on the stream costs ~5 us of serialisation on each side of the kernel.
struct LaunchEvents { hipEvent_t start = nullptr, stop = nullptr; };
inline thread_local LaunchEvents g_launch_events;

struct MfmaScanArgs {
    const void* corpus;     // [capacity][ld] bf16 or f32
    const void* queries;    // [nq_pad][ld] same dtype (nq_pad multiple of 256)
    const float* xnorm2;    // [capacity] fast squared norms (L2 only)
    const float* qnorm2;    // [nq_pad] fast squared query norms (L2 only)
    const float* thr;       // [nq_pad] fast-score thresholds (append when strictly better)
    uint2* lists;           // [nq_pad][cap] {score bits, row}
    uint32_t* counts;       // [nq_pad]
    uint32_t cap;
    uint32_t ld;            // elements per row (multiple of 64 bf16 / 32 f32): the K extent, and the query row stride
    uint32_t lda_bytes;     // split-bf16 pass over fp32 rows (bf16 4-wave kernel, a_wrap != 0): the corpus row
    uint32_t a_wrap;        // stride in bytes ([hi_j | lo_j] planes; the query rows are [hi_j | lo_j | hi_j])
    uint32_t nq_pad;
    uint32_t nq;            // real queries (<= nq_pad): batches of <= 64 over bf16 rows take the skinny kernel
    uint32_t row_begin;     // appends are limited to rows [row_begin, row_end); the launch
    uint32_t row_end;       // starts at the 256-row tile containing row_begin
    int metric;
    uint32_t* pace;         // >= 128 words for the sibling pacing counters (may be null)
    bool pace_is_zero;      // the caller already cleared them (else the launcher issues a memset)
    float* dense_out;       // non-null: sample pass, write every fast score [nq_pad][dense_ld]
    uint32_t dense_ld;      // (column = row - row_begin; multiple of 256)
    bool dense_grouped;     // sample pass, grouped form: dense_out is [nq_pad][dense_ld] with ONE score per group of
                            // mfma_dense_group_rows() consecutive rows (the group's best), dense_ld = groups per query
    void* dump;             // mfma_dump_bytes(num_cus) bytes of scratch: where the 4-wave kernel spills full hit logs (filtered launches)
    uint32_t* claims;       // kMfmaClaimWords words: claim bits of the 4-wave kernel's work stealing for THIS launch; null: off
    bool claims_is_zero;    // the caller already cleared them (else the launcher issues a memset)
};
void launch_scan_mfma(const MfmaScanArgs& a, int dtype, int num_cus, hipStream_t s);

// Experiment switches (VROD_DEBUG_*): parsed ONCE per process, here and nowhere else (debug_env() in vrod_index.hip).
// The defaults are what every measurement in DESIGN.md was taken with; they exist for A/B runs (scripts/env_sweep.sh),
// not as a product surface -- the three variables a user may set are documented in include/vrod.h.
struct DebugEnv {
    int pace_kt = 192;            // VROD_DEBUG_PACE_KT: sibling pacing interval of the 4-wave scan, K-tiles (0: off)
    int pace_tiles = 16;          // VROD_DEBUG_PACE: ... of the 8-wave fp32 scan, tiles (0: off)
    bool w4_steal = true;         // VROD_DEBUG_W4_STEAL=0: static shares, no work stealing
    bool skinny = true;           // VROD_DEBUG_SKINNY=0: batches of 5-64 queries take the 256-query tile
    uint64_t stage_growth = 0;    // VROD_DEBUG_STAGE_GROWTH: rows grow x g per filtered stage (0: the default, 5)
    uint64_t sample_rows = 0;     // VROD_DEBUG_SAMPLE_ROWS: cap on the sample pass's rows (0: none)
    uint32_t kp_margin = 0;       // VROD_DEBUG_KP_MARGIN: starting candidate margin k' - k of the batched cosine scan (0: 8)
    int early_sample = -1;        // VROD_DEBUG_EARLY_SAMPLE=0 / 1: order of the two searches in flight (-1: by corpus size)
    bool sample_grouped = true;   // VROD_DEBUG_SAMPLE_GROUPED=0: the sample pass writes every score
    bool graph = true;            // VROD_DEBUG_GRAPH=0: no hipGraph replay of small searches
    bool band = true;             // VROD_DEBUG_BAND=0: failed certificates go straight to the exact path
};
const DebugEnv& debug_env();
size_t mfma_dump_bytes(int num_cus);
constexpr size_t kMfmaClaimWords = 256;    // >= query blocks per launch x strips (8 x work-groups per XCD in all)
// Rows per group of the grouped sample form for this launch (32), or 0 when the kernel that would take it only
// writes every score (dense_grouped must then stay false).
uint32_t mfma_dense_group_rows(const MfmaScanArgs& a, int dtype);
// Largest batch the skinny (HBM-bound, <= 64 queries) form of the MFMA scan takes for bf16 rows of
// `row_bytes` bytes (split == true: the [hi | lo] planes of an fp32 corpus); 0 = none.
uint32_t mfma_skinny_max_queries(bool split, uint32_t row_bytes);

}  // namespace vrod
