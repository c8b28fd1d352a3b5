/*
 * vrod_oracle.h -- CPU oracle for the vRod brute-force similarity scan.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under vrod_amd/ (the product) may
 * include, link, load or call this file.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * PARITY UNPINNED BY THE REFERENCE: sekulas/vRod @ 2024-10-24 has no scan
 * (SearchSimilarCommand::execute is an empty stub, src/command/types.rs:127-132;
 * Database holds only a path, src/database/mod.rs:6-10) and no tests or golden
 * vectors.  This file is therefore a build-authored
 * restatement of what a plain single-threaded Rust loop over the reference's
 * only data type, Vec<Vec<f32>> (src/utils/embeddings.rs:29), would compute.
 * The spec it implements is frozen in DESIGN.md "Scan spec"; it is pinned
 * against an independent fp64 numpy/scikit-learn brute force in
 * tests/test_oracle.py and against the fixtures in tests/golden/.
 */
#ifndef VROD_ORACLE_H
#define VROD_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_DTYPE_F32 = 0, ORC_DTYPE_BF16 = 1 };
enum { ORC_METRIC_COSINE = 0, ORC_METRIC_L2 = 1 };

#define ORC_ID_NONE UINT64_MAX          /* id of an unfilled result slot (k > N) */
#define ORC_SCORE_NONE_BITS 0x7FC00000u /* score bits of an unfilled slot (quiet NaN) */

/* --- synthetic data (seed mixing documented in DESIGN.md) --- */
uint64_t orc_splitmix64(uint64_t x);
/* integer in [-131070, 131070]: sum of four 16-bit fields of the hash, centred */
int32_t orc_synth_int(uint64_t seed, uint64_t row, uint32_t dim, uint32_t col);
/* one unit-norm row, fp64 normalise, rounded to fp32 */
void orc_synth_row_f32(uint64_t seed, uint64_t row, uint32_t dim, float *out);
void orc_synth_rows_f32(uint64_t seed, uint64_t first_row, uint64_t n, uint32_t dim,
                        float *out, int threads);

/* --- element conversions --- */
uint16_t orc_f32_to_bf16(float x);  /* round to nearest even; NaN stays NaN */
float orc_bf16_to_f32(uint16_t h);

/* --- row preparation: what "insert" and "query" do to a raw fp32 vector --- */
/* COSINE: x / sqrt(sum x^2), sum and divide in fp64, left to right; zero -> zero.
 * L2: identity.  Then BF16: round each element to bf16 and widen back.  */
void orc_prepare_row(const float *in, uint32_t dim, int dtype, int metric, float *out);
void orc_prepare_rows(const float *in, uint64_t n, uint32_t dim, int dtype, int metric,
                      float *out, int threads);

/* --- canonical scores: strictly sequential fp32, separate mul and add, no FMA --- */
float orc_dot_canonical(const float *q, const float *x, uint32_t dim);
float orc_l2_canonical(const float *q, const float *x, uint32_t dim);

/* --- the scan: prepared corpus (n x dim fp32, row-major), prepared queries.
 * out_ids / out_scores are nq x k, best first; ties -> smaller id; NaN last;
 * slots beyond n are (ORC_ID_NONE, NaN).  id = row index + id_offset.
 * threads <= 1 : the single-threaded loop (vRod is Rc<RefCell>, !Send).
 * threads  > 1 : same per-row arithmetic, row ranges on pthreads, exact merge. */
int orc_scan_topk(const float *corpus, uint64_t n, uint32_t dim, const float *queries,
                  uint32_t nq, uint32_t k, int metric, uint64_t id_offset,
                  uint64_t *out_ids, float *out_scores, int threads);

/* merge n_lists per-shard results (each nq x k, best first) into one nq x k */
int orc_merge_topk(const uint64_t *ids, const float *scores, uint32_t n_lists, uint32_t nq,
                   uint32_t k, int metric, uint64_t *out_ids, float *out_scores);

#ifdef __cplusplus
}
#endif
#endif
