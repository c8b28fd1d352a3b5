/*
 * vrod.h -- C ABI of libvrod_hip.so: the MI355X (gfx950) brute-force similarity
 * scan + top-k that hangs under vRod's SEARCHSIMILAR command.
 *
 * Reference interfaces these entry points stand behind (sekulas/vRod @ 2024-10-24;
 * the reference has NO FFI and NO scan -- these are the slots it leaves empty):
 *   vrod_index_create / _destroy   <- the corpus handle `Database` must own
 *                                     (src/database/mod.rs:6-10, "//TODO collections")
 *   vrod_index_add                 <- BulkInsertCommand::execute / InsertCommand::execute
 *                                     (src/command/types.rs:56-80), rows are the
 *                                     Vec<Vec<f32>> of src/utils/embeddings.rs:29
 *   vrod_search                    <- SearchSimilarCommand::execute
 *                                     (src/command/types.rs:121-132), built by
 *                                     CommandBuilder::build "SEARCHSIMILAR"
 *                                     (src/command/builder.rs:68-72)
 *   vrod_last_error                <- the thiserror/anyhow surface (src/main.rs:36-42,
 *                                     src/command/builder.rs:10-15): status + message,
 *                                     never a panic or exception across the ABI
 * The Rust-side binding a maintainer would add is in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only; every function returns a vrod_status
 * (0 = ok); the caller owns every in/out buffer; the library never keeps a caller
 * pointer after return; calls on one handle must be serialised by the caller
 * (the reference is Rc<RefCell<_>>: single-threaded, src/command/types.rs:10).
 * All host-pointer entry points are synchronous.
 *
 * Semantics (frozen; DESIGN.md "Scan spec"): ids are row indices in insertion order
 * (+ id_offset); COSINE scores are dot products of L2-normalised vectors (higher is
 * better), L2 scores are squared Euclidean distances (lower is better); results are
 * best-first, ties broken by smaller id; unfilled slots (k > count) are
 * (VROD_ID_NONE, NaN).  Results are bit-identical to the CPU oracle (oracle/).
 *
 * Environment (read once per process; everything else the library reads is
 * VROD_DEBUG_*: A/B switches of the build's own experiments):
 *   VROD_F32_SPLIT = 0 | 1   fp32 handles: never | always scan batches through the
 *                            bf16 [hi | lo] planes (default: while the planes fit)
 *   VROD_RCCL = 0            multi-device handles exchange their lists by peer copies
 *                            instead of the RCCL all-gather
 *   VROD_RCCL_LIB = path     the RCCL library to bind (nothing else is tried; a path
 *                            that does not load means peer copies, said once on stderr)
 */
#ifndef VROD_H
#define VROD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vrod_index vrod_index;

typedef enum {
    VROD_OK = 0,
    VROD_ERR_INVALID_ARG = 1,   /* null pointer, zero dim, bad enum, k too large ... */
    VROD_ERR_INVALID_VALUE = 2, /* NaN or Inf in rows or queries */
    VROD_ERR_NO_DEVICE = 3,     /* no usable gfx950 device / HIP runtime failure at open */
    VROD_ERR_OUT_OF_MEMORY = 4,
    VROD_ERR_HIP = 5,           /* a HIP call failed; vrod_last_error() has the text */
    VROD_ERR_UNSUPPORTED = 6,
    VROD_ERR_INTERNAL = 7
} vrod_status;

enum { VROD_DTYPE_F32 = 0, VROD_DTYPE_BF16 = 1 };  /* storage + fast-pass type */
enum { VROD_METRIC_COSINE = 0, VROD_METRIC_L2 = 1 };

#define VROD_ID_NONE UINT64_MAX
#define VROD_MAX_K 3584u
#define VROD_MAX_DIM 32768u   /* one prepared row (fp32) must fit in a work-group's LDS next to its tiles */

/* Which fast pass vrod_search uses. AUTO picks by batch size and dtype. */
enum { VROD_PATH_AUTO = 0, VROD_PATH_STREAM = 1, VROD_PATH_MFMA = 2, VROD_PATH_EXACT = 3 };

/* Counters of the most recently COMPLETED search on a handle (bench.py / tests read these). */
typedef struct {
    uint32_t path;              /* VROD_PATH_* actually taken */
    uint32_t nq, k, kprime;     /* kprime = candidates re-scored per query (k + a margin that follows the certificates:
                                 * doubled after a search in which some failed, halved after 64 clean searches) */
    uint32_t scan_launches;     /* launches of the dominant scan kernel */
    uint32_t fallback_queries;  /* queries whose certificate failed -> exact path */
    float scan_ms;              /* HIP-event time of the scan kernel launches (sum, sample pass included) */
    float total_ms;             /* HIP-event time of the whole device pipeline */
    double scan_bytes;          /* algorithmic corpus bytes the scan launches covered */
    double scan_flops;          /* algorithmic flops (2*Q*N*d) of the scan launches */
    float max_fast_err;         /* max |fast - canonical| over re-scored candidates (relative to
                                 * |canonical| on the direct-L2 stream path, whose bound is relative) */
    float eps_bound;            /* the certificate's bound on that error */
    uint32_t split_pass;        /* 1: the batched fast pass of an F32 handle ran on its bf16 [hi | lo] planes */
    uint32_t band_queries;      /* of the fallback_queries: resolved by the band pass (one more shared scan that
                                 * collects the rows within the error bound of the k-th score), not the exact path */
    float sample_ms;            /* of scan_ms: the sample-pass launch of a staged MFMA search (its own kernel form; it
                                 * sets the first thresholds and covers no algorithmic work -- its rows are scanned
                                 * again by the first filtered launch); 0 when the search had none */
    uint32_t exchange;          /* multi-device handle: how the per-shard lists reached the merge --
                                 * 0 none (single device), 1 RCCL all-gather, 2 peer copies (VROD_RCCL=0) */
    float overlap_ms;           /* of scan_ms: how long this search's sample-pass launch and the PREVIOUS search's last scan
                                 * launch were both in flight (pipelined searches over up to 6M rows put the two side by
                                 * side; each launch's own time then includes waiting for compute units).  The sum of
                                 * scan_ms - overlap_ms over a run = the time at least one scan launch was in flight */
} vrod_search_stats;

/* --- lifecycle ------------------------------------------------------------- */
/* device_ids/n_devices: the GPUs the corpus is sharded over.  n_devices <= 1: one
 * handle drives one GPU (bench.py's default deployment is one process and one handle per GPU, see
 * vrod_search_device + vrod_merge_topk_device).  n_devices > 1 (at most 64, an id may repeat): ONE
 * handle owns a shard on every listed device -- the single-process model of SURVEY.md 8e, all a
 * host like vRod (Rc<RefCell<Database>>, one thread) needs.  Rows are dealt to the shards in blocks
 * of 65536 in insertion order; a search runs on every shard at once; the per-shard top-k lists are
 * exchanged with ONE ncclAllGather per device (RCCL, communicator from ncclCommInitAll over the
 * distinct devices, one exchange stream per device, inside ncclGroupStart/End) and merged on
 * device_ids[0]; ids are global insertion indices as for a single device.  librccl is loaded when
 * the first such handle is created (VROD_RCCL_LIB: its path; VROD_RCCL=0: peer copies to the first
 * device instead of RCCL).  Device pointers passed to such a handle live on device_ids[0].  The
 * pipelined _begin_/_end form works as for one device: while the exchange and merge of batch s run,
 * every device already scans batch s+1. */
int vrod_index_create(vrod_index **out, uint32_t dim, int dtype, int metric,
                      const int *device_ids, int n_devices);
int vrod_index_destroy(vrod_index *idx);

/* --- corpus ---------------------------------------------------------------- */
int vrod_index_reserve(vrod_index *idx, uint64_t n_rows);
/* rows: n x dim fp32, row-major, host memory. Normalised (COSINE) and converted
 * (BF16) on the device. Ids continue from the current count. */
int vrod_index_add(vrod_index *idx, const float *rows, uint64_t n);
/* Append rows [first_row, first_row+n) of the synthetic stream `seed`
 * (random unit vectors), generated on the device. */
int vrod_index_add_synthetic(vrod_index *idx, uint64_t seed, uint64_t first_row, uint64_t n);
int vrod_index_count(const vrod_index *idx, uint64_t *out_count);
/* Shard support: reported id = local row index + offset. */
int vrod_index_set_id_offset(vrod_index *idx, uint64_t offset);
/* Copy prepared rows [first, first+n) back as fp32 (bf16 widened): n x dim. */
int vrod_index_get_rows(vrod_index *idx, uint64_t first, uint64_t n, float *out_rows);

/* --- search ---------------------------------------------------------------- */
/* queries: nq x dim fp32 host; out_ids: nq x k; out_scores: nq x k. */
int vrod_search(vrod_index *idx, const float *queries, uint32_t nq, uint32_t k,
                uint64_t *out_ids, float *out_scores);
/* Same with device pointers (queries, outputs) and a HIP stream (hipStream_t, may be
 * NULL). Returns after the results are complete in device memory. */
int vrod_search_device(vrod_index *idx, const float *d_queries, uint32_t nq, uint32_t k,
                       uint64_t *d_out_ids, float *d_out_scores, void *stream);
/* Queries from the synthetic stream, generated on the device (bench / tests). */
int vrod_search_synthetic_device(vrod_index *idx, uint64_t seed, uint64_t first_row,
                                 uint32_t nq, uint32_t k, uint64_t *d_out_ids,
                                 float *d_out_scores, void *stream);
/* Pipelined form of vrod_search_device: _begin_ enqueues the whole search on the library's own
 * stream, ordered after everything `stream` holds at the time of the call, and returns without
 * waiting for the device; vrod_search_end completes the OLDEST begun search (FIFO) -- when it
 * returns VROD_OK that search's results are complete in device memory.  At most two searches may
 * be pending: begin(s+1) before end(s) keeps the device busy while the host, and the caller's
 * exchange of batch s (all-gather + merge), catch up.  A begun search must be
 * ended; the queries of a _begin_device call and both output buffers must stay untouched until
 * then.  While a search is pending every other entry point on the handle that touches the
 * corpus or the stream fails with VROD_ERR_INVALID_ARG.  Errors that depend on the data (NaN or
 * Inf in the queries) are reported by vrod_search_end. */
int vrod_search_begin_device(vrod_index *idx, const float *d_queries, uint32_t nq, uint32_t k,
                             uint64_t *d_out_ids, float *d_out_scores, void *stream);
int vrod_search_begin_synthetic_device(vrod_index *idx, uint64_t seed, uint64_t first_row,
                                       uint32_t nq, uint32_t k, uint64_t *d_out_ids,
                                       float *d_out_scores, void *stream);
int vrod_search_end(vrod_index *idx);
int vrod_search_pending(const vrod_index *idx, uint32_t *out_pending);

/* Merge n_lists per-shard results (device, each nq x k, list-major: [list][q][k]) into
 * one nq x k on `device` -- the step after the RCCL all-gather. */
int vrod_merge_topk_device(int device, int metric, const uint64_t *d_ids,
                           const float *d_scores, uint32_t n_lists, uint32_t nq,
                           uint32_t k, uint64_t *d_out_ids, float *d_out_scores,
                           void *stream);

/* Same merge over ONE packed buffer: per list, nq*k ids (u64) immediately followed by nq*k
 * scores (f32) -- the layout that lets the exchange be a single all-gather. nq*k must be even. */
int vrod_merge_topk_packed_device(int device, int metric, const void *d_packed, uint32_t n_lists,
                                  uint32_t nq, uint32_t k, uint64_t *d_out_ids,
                                  float *d_out_scores, void *stream);

/* --- knobs & introspection --------------------------------------------------
 * Environment, read when a handle is created: VROD_F32_SPLIT.  An F32 handle keeps bf16
 * [hi | lo] planes of its rows (a second copy of the corpus, built at the first batched search)
 * and runs batched searches as three bf16 matrix-core products instead of one fp32 one (2.5x
 * faster at 10M x 1536, batch 256); the results are the same bits: only the fast pass and the
 * certificate's bound change.  Default: on while the planes leave max(1/8 of the device, 4 GiB)
 * free, and switched off for the handle after two searches in which more than 1/8 of the queries
 * failed the (wider) certificate.  =0: never.  =1: always (no margin check, never switched off).
 * If the planes cannot be allocated the handle quietly keeps the fp32 pass. */
int vrod_index_set_path(vrod_index *idx, int path);      /* VROD_PATH_* (default AUTO) */
int vrod_index_set_profiling(vrod_index *idx, int on);   /* 1: scan_ms (events attached to the scan dispatches), 2: + total_ms (stream markers) */
int vrod_index_last_stats(const vrod_index *idx, vrod_search_stats *out);
/* The same counters for ONE shard of the most recently completed search (a multi-device handle has a shard per
 * entry of device_ids, in that order; a single-device handle has shard 0 = itself), and the device the shard
 * lives on (out_device may be NULL): what each GPU of such a handle spent.  shard out of range: INVALID_ARG. */
int vrod_index_shard_stats(const vrod_index *idx, uint32_t shard, int *out_device, vrod_search_stats *out);
const char *vrod_last_error(void);                       /* thread-local text */
const char *vrod_version(void);

/* --- synthetic stream on the device (tests: bit-parity with the oracle) ----- */
int vrod_synth_rows_device(int device, uint64_t seed, uint64_t first_row, uint64_t n,
                           uint32_t dim, float *d_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VROD_H */
