// vrod_index.hip -- the index object and the C ABI of libvrod_hip.so (include/vrod.h).
//
// This is the corpus owner vRod's `Database` never got (reference src/database/mod.rs:6-10:
// "//TODO collections") and the body SearchSimilarCommand::execute never got
// (src/command/types.rs:127-132).  Pipeline of one search (DESIGN.md "Pipeline"):
//
//   prepare queries -> FAST PASS (stream scan | MFMA scan with threshold filter)
//     -> k' candidates per query + T (bound on the fast score of everything left out)
//     -> canonical re-score of the candidates (oracle order, bit-exact)
//     -> final ordering by (canonical score, id) + exactness certificate
//     -> queries whose certificate fails take the EXACT path (canonical scan of all rows)
//
// so the returned ids and score bits are identical to the CPU oracle's for every input.
// There is no CPU fallback anywhere: without a gfx950 device every entry point fails.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl is dlopen'ed when a multi-device handle is created

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vrod.h"
#include "vrod_common.h"
#include "vrod_kernels.h"

using namespace vrod;

// ------------------------------------------------------------------ errors
static thread_local std::string g_last_error;

static int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? VROD_ERR_OUT_OF_MEMORY : VROD_ERR_HIP,     \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,       \
                        __LINE__);                                                             \
    } while (0)

#define VROD_TRY(expr)              \
    do {                            \
        int rc_ = (expr);           \
        if (rc_ != VROD_OK) return rc_; \
    } while (0)

// ------------------------------------------------------------------ device buffer that only grows
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return VROD_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(VROD_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        }
        cap = want;
        return VROD_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T* as() const { return (T*)p; }
};

static inline uint64_t round_up(uint64_t x, uint64_t m) { return (x + m - 1) / m * m; }

// a search slot's block of device words: 64 scalars, then one region of sibling-pacing counters and one of work-stealing
// claim bits per scan launch of the search (8 of each; a search with more launches reuses them behind a memset)
constexpr uint32_t kPaceRegions = 8, kPaceWords = 192, kClaimWords = (uint32_t)kMfmaClaimWords;
constexpr size_t kSlotFlagBytes = (64 + kPaceRegions * (kPaceWords + kClaimWords)) * 4;

// ------------------------------------------------------------------ one search in flight
// A search is ENQUEUED (every launch up to the D2H of its status block) and later COMPLETED (wait
// for that copy, read the certificate's verdicts, run the exact path for the rare failures).  Two
// slots let the caller enqueue search s+1 before it completes search s, so the device never waits
// for the host between batches and the caller's exchange of batch s (all-gather + merge on its own
// stream) overlaps the scan of batch s+1.
struct Pending {
    bool active = false;
    bool trivial = false;          // nq == 0 or empty corpus: the outputs are already final
    uint32_t nq = 0, k = 0, kp = 0;
    int path = 0, eps_mode = 0;
    bool split = false;            // the fast pass of this search ran on the bf16 planes
    float eps_c = 0.f;
    uint64_t N = 0;
    uint64_t* out_ids = nullptr;
    float* out_scores = nullptr;
    // Each slot has its own stream and its own workspaces, so the tail of search s (compaction,
    // re-score, certificate, read-back) can run beside the head -- and, on the stream path, the
    // scan -- of search s+1.  Only the corpus (read-only during a search) is shared.
    hipStream_t stream = nullptr;
    uint32_t* flags = nullptr;     // [0] bad-value flag, [1] max query norm^2 bits, [2] max err bits, [64..] pacing counters, then claim bits (kSlotFlagBytes)
    DevBuf q_raw, q_lp, scores, keys_a, keys_b, lists, small, hist, cand_rows, cand_fast, cand_canon;
    DevBuf q_f32;                  // prepared queries (the exact path re-reads them)
    DevBuf dump;                   // MFMA path: spill regions of the 4-wave kernel's hit logs (scratch, mfma_dump_bytes)
    DevBuf q_planes;               // split pass: [nq_pad][3 * ldp] bf16, [hi_j | lo_j | hi_j] per K-tile j
    // band pass (second chance of the queries whose certificate failed, search_complete)
    DevBuf band_idx, band_q, band_q_lp, band_planes, band_small, band_ids, band_scores;
    uint32_t nq_pad = 0;           // of the enqueued search (the list / counter / threshold blocks are sized by it)
    uint32_t pace_launches = 0;    // scan launches of this search so far (pacing-counter regions)
    uint32_t* h_readback = nullptr;   // pinned host block: status[nq] + 4 scalars, one D2H per search
    size_t h_readback_words = 0;
    hipEvent_t done = nullptr;     // recorded behind the D2H
    hipEvent_t scans_done = nullptr;   // recorded behind the last scan launch
    hipEvent_t mid_done = nullptr;     // recorded behind the second-to-last scan launch of a staged MFMA search (else with scans_done)
    bool mid_recorded = false;         // ... in the search being enqueued
    std::vector<hipEvent_t> ev;    // profiling events
    size_t ev_used = 0, t0 = 0, t1 = 0;
    std::vector<std::pair<size_t, size_t>> scan_pairs;
    int sample_pair = -1;          // index in scan_pairs of the sample-pass launch (-1: none)
    int tail_pair = -1;            // index in scan_pairs of the last filtered launch of a staged MFMA search: timed by idx->tail_ev[seq & 3]
    uint32_t seq = 0;              // number of this search on its handle (n_begun when it was enqueued)
    bool early_sample = false;     // its sample pass was ordered in front of the previous search's last stage
    uint32_t kp_boost_used = 1;    // the candidate-margin multiplier this search was enqueued with
    vrod_search_stats st{};

    // hipGraph replay of small, launch-bound searches (search_enqueue): the launches of a search
    // whose every pointer and size equals the captured one are replayed as one graph launch
    struct GraphKey {
        const void *q = nullptr, *oi = nullptr, *os = nullptr, *corpus = nullptr, *xn = nullptr;
        const void* bufs[12] = {};
        uint64_t N = 0, id_offset = 0;
        uint32_t nq = 0, k = 0;
        int path = 0;
        bool operator==(const GraphKey& o) const { return memcmp(this, &o, sizeof *this) == 0; }
    };
    GraphKey gkey{};               // of the last search enqueued in this slot
    bool gkey_valid = false;
    hipGraphExec_t gexec = nullptr;   // captured for gkey_graph
    GraphKey gkey_graph{};
    bool graph_off = false;        // a capture failed once: this slot stays on plain launches
    // what a replay must restore of the enqueue's host-side results
    uint32_t g_kp = 0; int g_path = 0, g_eps_mode = 0; float g_eps_c = 0.f; vrod_search_stats g_st{};
};

// ------------------------------------------------------------------ the index
struct vrod_index {
    int device = 0;
    int num_cus = 256;
    uint32_t dim = 0, ld = 0;
    int dtype = VROD_DTYPE_F32, metric = VROD_METRIC_COSINE;
    size_t esize = 4;
    uint64_t count = 0, capacity = 0, id_offset = 0;
    void* corpus = nullptr;     // [capacity][ld]
    float* xnorm2 = nullptr;    // [capacity]
    uint32_t* max_xn2_bits = nullptr;  // device scalar: max squared row norm (float bits)
    hipStream_t stream = nullptr;
    int path = VROD_PATH_AUTO;
    int profiling = 0;
    vrod_search_stats stats{};

    // fp32 corpus: bf16 planes [hi | lo] of the prepared rows for the batched fast pass on the bf16
    // matrix cores (kernels_prep.hip split_rows_kernel), a second copy of the corpus.  Built lazily
    // for rows [0, planes_rows) at the next batched search, on by default while the device keeps a
    // margin of free memory beside them (VROD_F32_SPLIT=0: never, =1: always try).
    bool split_enabled = false;
    bool split_forced = false;     // VROD_F32_SPLIT=1: no memory-margin check, never switched off by the failure count
    uint32_t split_bad = 0;        // split searches that sent more than 1/8 of their queries to the exact path
    void* planes = nullptr;        // [planes_cap][2 * ldp] bf16, [hi_j | lo_j] per 64-element K-tile j
    uint64_t planes_cap = 0, planes_rows = 0;
    uint32_t ldp = 0;              // dim rounded up to 64 (bf16 128-B lines)

    // workspaces
    DevBuf raw_stage, nrm_ws, out_ids, out_scores;
    uint32_t* flags = nullptr;  // [0] bad-value flag of the insert path; [8] max squared row norm
    Pending slot[2];
    // start / stop of the last filtered scan launch of the four most recent searches (Timer::arm_tail): the next search's
    // sample launch may overlap it, and measures by how much when it completes
    struct TailEv { hipEvent_t start = nullptr, stop = nullptr; bool armed = false; };
    TailEv tail_ev[4];
    // Self-tuning candidate margin of the batched scan (search_enqueue_body / search_complete): k' = k + margin * kp_boost.
    // A failed certificate costs a band pass (one more scan of the corpus); doubling the margin costs a few per cent
    // of hits.  kp_boost doubles (up to 8) after a search with failures, halves after 64 clean ones; failures AT 8 mean
    // the margin is not what those queries lack (exact duplicates): back to 1 and left alone for 256 searches.
    static constexpr uint32_t kMaxKpBoost = 8;
    uint32_t kp_boost = 1, kp_clean = 0, kp_hold = 0;
    uint32_t n_begun = 0, n_ended = 0;   // searches enqueued / completed: slot = counter & 1
    hipEvent_t caller_ev = nullptr;      // orders the caller's stream before ours

    uint32_t n_pending() const { return n_begun - n_ended; }

    // --- composite handle (vrod_index_create with n_devices > 1): the rows are dealt to the
    // shards in blocks of kShardBlock rows (global row r -> shard (r / B) % G, local row
    // (r / (B*G)) * B + r % B), every shard is a complete single-device index of its own, and a
    // search runs on all of them at once, then gathers and merges on the first device.
    std::vector<vrod_index*> shards;
    IdMap deal{};                        // a shard of a composite handle: how its local rows become global ids
    // One group per DISTINCT device of the handle: the shards living on it, an exchange stream, the
    // device's rank in the handle's RCCL communicator, and per pipeline slot the packed block this
    // device contributes ([M][block]: one list per local shard, "no result" lists up to M = the
    // largest group) and what it receives ([U][M][block]: every device's contribution).
    struct DevGroup {
        int device = 0;
        std::vector<size_t> members;     // indices into shards
        hipStream_t xstream = nullptr;
        ncclComm_t comm = nullptr;
        DevBuf send[2], recv[2];
        size_t filled_nk[2] = {0, 0};    // nq*k the unused list slots of send[] were last filled for (+1)
    };
    std::vector<DevGroup> groups;
    std::vector<std::pair<size_t, size_t>> shard_home;   // shard -> (group, position in the group)
    bool use_rccl = false;
    std::vector<DevBuf> sh_q[2];         // per slot, per shard: the batch's raw queries on the shard's device
    struct CompPending {
        uint32_t nq = 0, k = 0;
        uint64_t* out_ids = nullptr;     // device pointers on groups[0].device
        float* out_scores = nullptr;
        hipEvent_t caller_ev = nullptr;  // the caller's stream at _begin: the merge writes out_ids / out_scores behind it
        bool ordered = false;            // ... recorded for this search (device outputs)
    } cslot[2];
    bool composite() const { return !shards.empty(); }

    size_t row_bytes() const { return (size_t)ld * esize; }
};

static int set_device(const vrod_index* idx) {
    HIP_TRY(hipSetDevice(idx->device));
    return VROD_OK;
}

// ids a search of this index reports: row + id_offset, or the dealing map of a composite handle's shard
static IdMap idmap_of(const vrod_index* idx) {
    IdMap m = idx->deal;
    if (!m.block_rows) m.offset = idx->id_offset;
    return m;
}

static int index_reserve(vrod_index* idx, uint64_t n_rows) {
    const uint64_t want = round_up(std::max<uint64_t>(n_rows, 1), kRowTile);
    if (want <= idx->capacity) return VROD_OK;
    if (want > 0xFFFFFF00ull) return fail(VROD_ERR_UNSUPPORTED, "more than 2^32-256 rows per shard");
    void* nc = nullptr;
    float* nx = nullptr;
    HIP_TRY(hipMalloc(&nc, want * idx->row_bytes()));
    hipError_t e = hipMalloc((void**)&nx, want * sizeof(float));
    if (e != hipSuccess) { (void)hipFree(nc); return fail(VROD_ERR_OUT_OF_MEMORY, "hipMalloc norms: %s", hipGetErrorString(e)); }
    const size_t used = idx->count * idx->row_bytes();
    hipError_t ce = hipSuccess;
    if (idx->count) {
        ce = hipMemcpyAsync(nc, idx->corpus, used, hipMemcpyDeviceToDevice, idx->stream);
        if (ce == hipSuccess) ce = hipMemcpyAsync(nx, idx->xnorm2, idx->count * sizeof(float), hipMemcpyDeviceToDevice, idx->stream);
    }
    if (ce == hipSuccess) ce = hipMemsetAsync((char*)nc + used, 0, want * idx->row_bytes() - used, idx->stream);
    if (ce == hipSuccess) ce = hipMemsetAsync(nx + idx->count, 0, (want - idx->count) * sizeof(float), idx->stream);
    if (ce == hipSuccess) ce = hipStreamSynchronize(idx->stream);
    if (ce != hipSuccess) {   // the old corpus stays in place; the new blocks are given back
        (void)hipStreamSynchronize(idx->stream);
        (void)hipFree(nc);
        (void)hipFree(nx);
        return fail(VROD_ERR_HIP, "growing the corpus failed: %s", hipGetErrorString(ce));
    }
    if (idx->corpus) (void)hipFree(idx->corpus);
    if (idx->xnorm2) (void)hipFree(idx->xnorm2);
    if (idx->planes) { (void)hipFree(idx->planes); idx->planes = nullptr; idx->planes_cap = idx->planes_rows = 0; }   // rebuilt lazily
    idx->corpus = nc;
    idx->xnorm2 = nx;
    idx->capacity = want;
    return VROD_OK;
}

// Prepare `n` raw fp32 rows already in device memory and append them.
static int append_prepared(vrod_index* idx, const float* d_raw, uint64_t n) {
    VROD_TRY(idx->nrm_ws.ensure(n * sizeof(double)));
    char* dst = (char*)idx->corpus + idx->count * idx->row_bytes();
    launch_prepare_rows(d_raw, n, idx->dim, idx->ld, idx->metric, idx->dtype, idx->nrm_ws.as<double>(),
                        &idx->flags[0], idx->dtype == VROD_DTYPE_F32 ? (float*)dst : nullptr,
                        idx->dtype == VROD_DTYPE_BF16 ? dst : nullptr, idx->stream);
    launch_row_fastnorm(dst, idx->dtype, n, idx->ld, idx->xnorm2 + idx->count, idx->max_xn2_bits, idx->stream);
    HIP_TRY(hipGetLastError());
    return VROD_OK;
}

static int check_bad_flag(vrod_index* idx, const char* what) {
    uint32_t bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, &idx->flags[0], 4, hipMemcpyDeviceToHost, idx->stream));
    HIP_TRY(hipStreamSynchronize(idx->stream));
    if (bad) {
        HIP_TRY(hipMemsetAsync(&idx->flags[0], 0, 4, idx->stream));
        HIP_TRY(hipStreamSynchronize(idx->stream));
        return fail(VROD_ERR_INVALID_VALUE, "%s contain NaN or Inf", what);
    }
    return VROD_OK;
}

static const uint64_t kStageRows = 1u << 16;

static int index_add(vrod_index* idx, const float* rows, uint64_t n, bool synthetic, uint64_t seed,
                     uint64_t first_row) {
    if (!n) return VROD_OK;
    VROD_TRY(set_device(idx));
    if (idx->count + n > idx->capacity) {
        uint64_t want = std::max(idx->count + n, idx->capacity + idx->capacity / 2);
        if (synthetic || idx->capacity == 0) want = idx->count + n;
        VROD_TRY(index_reserve(idx, want));
    }
    const uint64_t count0 = idx->count;
    // the max squared row norm is accumulated (atomicMax) by the very launches whose rows may be
    // rejected: keep the value it had, so that a rejected add cannot widen (or, with an Inf row,
    // void) the certificate bound of every later search
    HIP_TRY(hipMemcpyAsync(&idx->flags[9], idx->max_xn2_bits, 4, hipMemcpyDeviceToDevice, idx->stream));
    const uint64_t chunk = std::min<uint64_t>(n, std::max<uint64_t>(1024, std::min<uint64_t>(kStageRows, (256ull << 20) / (idx->dim * 4ull))));
    VROD_TRY(idx->raw_stage.ensure(chunk * idx->dim * sizeof(float)));
    for (uint64_t done = 0; done < n; done += chunk) {
        const uint64_t m = std::min(chunk, n - done);
        if (synthetic) {
            launch_synth_rows(seed, first_row + done, m, idx->dim, idx->raw_stage.as<float>(), idx->stream);
        } else {
            HIP_TRY(hipMemcpyAsync(idx->raw_stage.p, rows + done * idx->dim, m * idx->dim * sizeof(float),
                                   hipMemcpyHostToDevice, idx->stream));
        }
        VROD_TRY(append_prepared(idx, idx->raw_stage.as<float>(), m));
        idx->count += m;
        // the staging buffer is reused by the next chunk: stream order keeps it safe
    }
    int rc = check_bad_flag(idx, "rows");
    if (rc != VROD_OK) {  // roll back: re-zero the rows just written
        idx->count = count0;
        (void)hipMemsetAsync((char*)idx->corpus + count0 * idx->row_bytes(), 0, n * idx->row_bytes(), idx->stream);
        (void)hipMemsetAsync(idx->xnorm2 + count0, 0, n * sizeof(float), idx->stream);
        (void)hipMemcpyAsync(idx->max_xn2_bits, &idx->flags[9], 4, hipMemcpyDeviceToDevice, idx->stream);
        (void)hipStreamSynchronize(idx->stream);
        return rc;
    }
    return VROD_OK;
}

// ------------------------------------------------------------------ experiment switches
namespace vrod {
const DebugEnv& debug_env() {
    static const DebugEnv env = [] {
        DebugEnv d;
        auto num = [](const char* name, long long dflt) { const char* e = getenv(name); return e && *e ? atoll(e) : dflt; };
        auto on = [](const char* name) { const char* e = getenv(name); return !e || e[0] != '0'; };
        d.pace_kt = (int)num("VROD_DEBUG_PACE_KT", d.pace_kt);
        d.pace_tiles = (int)num("VROD_DEBUG_PACE", d.pace_tiles);
        d.w4_steal = on("VROD_DEBUG_W4_STEAL");
        d.skinny = on("VROD_DEBUG_SKINNY");
        d.stage_growth = (uint64_t)num("VROD_DEBUG_STAGE_GROWTH", 0);
        d.sample_rows = (uint64_t)num("VROD_DEBUG_SAMPLE_ROWS", 0);
        d.kp_margin = (uint32_t)num("VROD_DEBUG_KP_MARGIN", 0);
        if (const char* e = getenv("VROD_DEBUG_EARLY_SAMPLE")) d.early_sample = e[0] != '0' ? 1 : 0;
        d.sample_grouped = on("VROD_DEBUG_SAMPLE_GROUPED");
        d.graph = on("VROD_DEBUG_GRAPH");
        d.band = on("VROD_DEBUG_BAND");
        return d;
    }();
    return env;
}
}  // namespace vrod

// ------------------------------------------------------------------ search pipeline
struct Timer {
    vrod_index* idx;
    Pending& P;
    Timer(vrod_index* i, Pending& p) : idx(i), P(p) {}
    size_t mark() {
        if (idx->profiling < 2) return 0;   // stream markers only at level 2 (total_ms)
        if (P.ev_used == P.ev.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return 0;
            P.ev.push_back(e);
        }
        (void)hipEventRecord(P.ev[P.ev_used], P.stream);
        return P.ev_used++;
    }
    // two fresh events for the next scan launch (attached to the dispatch, not recorded as markers)
    void arm(size_t& a, size_t& b) {
        a = b = 0;
        if (!idx->profiling) return;
        while (P.ev.size() < P.ev_used + 2) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            P.ev.push_back(e);
        }
        a = P.ev_used++;
        b = P.ev_used++;
        g_launch_events.start = P.ev[a];
        g_launch_events.stop = P.ev[b];
    }
    float ms(size_t a, size_t b) {
        float m = 0.f;
        if (idx->profiling && a < P.ev_used && b < P.ev_used) (void)hipEventElapsedTime(&m, P.ev[a], P.ev[b]);
        return m;
    }
    // The LAST filtered launch of a staged MFMA search is timed by events of the handle's ring instead (tail_ev[seq & 3]):
    // the next search's sample pass may run beside it, and that search wants both intervals when it is completed --
    // by then this slot's own events belong to the search after it.
    void arm_tail() {
        if (!idx->profiling) return;
        vrod_index::TailEv& T = idx->tail_ev[P.seq & 3];
        if (!T.start && hipEventCreate(&T.start) != hipSuccess) return;
        if (!T.stop && hipEventCreate(&T.stop) != hipSuccess) return;
        g_launch_events.start = T.start;
        g_launch_events.stop = T.stop;
        T.armed = true;
        P.tail_pair = (int)P.scan_pairs.size();
    }
    float pair_ms(size_t i) {
        if ((int)i == P.tail_pair) {
            const vrod_index::TailEv& T = idx->tail_ev[P.seq & 3];
            float m = 0.f;
            if (idx->profiling && T.armed) (void)hipEventElapsedTime(&m, T.start, T.stop);
            return m;
        }
        return ms(P.scan_pairs[i].first, P.scan_pairs[i].second);
    }
    // ms during which this search's sample launch and the previous search's last filtered launch were BOTH in flight
    float sample_overlap_ms() {
        if (!idx->profiling || P.sample_pair < 0 || !P.early_sample || P.seq == 0) return 0.f;
        const vrod_index::TailEv& T = idx->tail_ev[(P.seq - 1) & 3];
        const size_t a = P.scan_pairs[P.sample_pair].first, b = P.scan_pairs[P.sample_pair].second;
        if (!T.armed || a >= P.ev_used || b >= P.ev_used) return 0.f;
        float tail_len = 0.f, s0 = 0.f, s1 = 0.f;   // everything relative to the start of that launch
        if (hipEventElapsedTime(&tail_len, T.start, T.stop) != hipSuccess) return 0.f;
        if (hipEventElapsedTime(&s0, T.start, P.ev[a]) != hipSuccess) return 0.f;
        if (hipEventElapsedTime(&s1, T.start, P.ev[b]) != hipSuccess) return 0.f;
        return std::max(0.f, std::min(s1, tail_len) - std::max(s0, 0.f));
    }
};

static uint32_t choose_kp(uint64_t count, uint32_t k) {
    uint64_t kp = (uint64_t)k + std::max<uint32_t>(16, k / 8);
    if (kp > count) kp = count;
    if (kp > kSelectChunk / 2) kp = kSelectChunk / 2;
    return (uint32_t)kp;
}

// select chain over fast (or canonical) scores of `nq` queries -> keys of <= kSelectChunk per query
// returns pointer/ld/n of the final key set through out params.
static int select_chain(vrod_index* idx, Pending& P, const float* d_scores, uint64_t score_ld, uint64_t n, int nq,
                        uint32_t kp, const uint64_t** out_keys, uint64_t* out_ld, uint64_t* out_n) {
    const uint64_t nch0 = (n + kSelectChunk - 1) / kSelectChunk;
    const uint64_t ld_a = nch0 * kp;
    VROD_TRY(P.keys_a.ensure((size_t)nq * ld_a * 8));
    uint64_t cur_n = launch_select_from_scores(d_scores, score_ld, n, nq, idx->metric, kp, P.keys_a.as<uint64_t>(), ld_a, P.stream);
    const uint64_t* cur = P.keys_a.as<uint64_t>();
    uint64_t cur_ld = ld_a;
    bool a_is_cur = true;
    while (cur_n > kSelectChunk) {
        const uint64_t nch = (cur_n + kSelectChunk - 1) / kSelectChunk;
        const uint64_t nld = nch * kp;
        DevBuf& dst = a_is_cur ? P.keys_b : P.keys_a;
        VROD_TRY(dst.ensure((size_t)nq * nld * 8));
        cur_n = launch_select_from_keys(cur, cur_ld, cur_n, nq, kp, dst.as<uint64_t>(), nld, P.stream);
        cur = dst.as<uint64_t>();
        cur_ld = nld;
        a_is_cur = !a_is_cur;
    }
    *out_keys = cur;
    *out_ld = cur_ld;
    *out_n = cur_n;
    return VROD_OK;
}

// Stage plan of the MFMA path (DESIGN.md "Kernels").  Stage 0 is a DENSE sample (all scores of
// the first S rows written out, threshold = exact k'-th best of them); every later stage is a
// filtered launch over g times more rows than everything before it, followed by a compaction
// (keep the best k', publish the k'-th score as the next threshold).  Each filtered stage thus
// expects about g*k' rows per query to beat its threshold: enough that a list cannot come up
// short, far too few to overflow it or to slow the scan.
struct StagePlan { uint32_t S, j; std::vector<uint64_t> bounds; };
static StagePlan plan_stages(uint64_t N, uint32_t kp, uint32_t cap, uint32_t max_sample_rows) {
    StagePlan p;
    // Growth per filtered stage.  A stage over rows (b, g*b] runs against the k'-th best of the
    // first b rows, so about k'*(g-1) rows per query beat its threshold: that must stay well under
    // the list capacity, and -- measured -- appends are not free: the first stage after the
    // 16K-row sample appends one row in 630 per query (~100 per 256x256 tile) and runs at 1.9 us
    // per 1000 rows against 1.3 once appends are rare.  The extra time of a stage is ~ (g-1), the
    // number of stages ~ 1/ln g, each costing a launch ramp and a compaction (~40 us): the total is
    // flat between g = 4 and 8 and twice as large at g = 25 (two stages at 10M rows: tried,
    // +0.25 ms per batch).
    // Round 2, same box, batch 1024 x 768 (profiles/r02/mfma_experiments.md): 10M rows, g = 3 / 4 / 5 / 6 / 8 -> 12.69 / 12.70 /
    // 12.70 / 12.75 / 12.77-12.87 ms per batch; with the candidate margin at 8, 5M rows g = 4 / 5 / 6 / 8 -> 6.47 / 6.41 / 6.48 /
    // 6.49, 2.5M -> 3.24 / 3.26 / 3.27 / 3.25, 1.25M -> 1.759 / 1.755 / 1.757 / 1.790.  What a stage pays per hit is the OTHER
    // three waves of the work-group waiting at the next barrier for the wave that walks a hit column (~0.35 us of
    // work-group time per append, not 0.1), against ~40 us of ramp + compaction per extra stage: g = 5 at every size.
    // VROD_DEBUG_STAGE_GROWTH overrides (tuning).
    const uint64_t g_env = debug_env().stage_growth;
    const uint64_t g_auto = 5;
    const uint64_t g = std::max<uint64_t>(2, std::min<uint64_t>(g_env ? g_env : g_auto, cap / (3ull * kp)));
    // sample: N/g^2 rows, at most one round of work-groups (one 256-row tile per work-group of
    // the dense launch)
    const uint64_t s_env = debug_env().sample_rows;   // tuning knob
    if (s_env) max_sample_rows = (uint32_t)std::min<uint64_t>(s_env, max_sample_rows);
    uint64_t S = std::min<uint64_t>(N / (g * g), max_sample_rows);
    S = std::max<uint64_t>(S, std::min<uint64_t>(N, std::max<uint64_t>(4ull * kp, kRowTile)));
    S = std::min<uint64_t>(round_up(S, kRowTile), N);
    p.S = (uint32_t)S;
    p.j = (uint32_t)std::min<uint64_t>(kp, S);
    for (uint64_t b = S * g; b < N; b *= g) {
        if (N - b < b / 2) break;                    // the tail would be a sliver: fold it in
        p.bounds.push_back(b / kRowTile * kRowTile);
    }
    p.bounds.push_back(N);
    return p;
}

// The event behind a search's last scan launch; a search that did not mark an earlier point (mid_done: behind the
// second-to-last stage of a staged MFMA search) marks it here as well.
static int record_scans_done(Pending& P, hipStream_t s) {
    if (!P.mid_recorded) { HIP_TRY(hipEventRecord(P.mid_done, s)); P.mid_recorded = true; }
    HIP_TRY(hipEventRecord(P.scans_done, s));
    return VROD_OK;
}

// Enqueue one search into slot P: every launch up to the D2H of the status block.  Returns
// without waiting for the device (except on the trivial empty-corpus case).
static int search_enqueue_body(vrod_index* idx, Pending& P, const float* d_queries_raw, uint32_t nq, uint32_t k,
                               uint64_t* d_out_ids, float* d_out_scores, bool in_graph) {
    vrod_search_stats& st = P.st;
    st = vrod_search_stats{};
    st.nq = nq;
    st.k = k;
    P.nq = nq; P.k = k; P.out_ids = d_out_ids; P.out_scores = d_out_scores;
    P.trivial = true;
    P.mid_recorded = false;
    P.ev_used = 0; P.t0 = P.t1 = 0; P.scan_pairs.clear(); P.sample_pair = -1;
    P.tail_pair = -1; P.early_sample = false; P.seq = idx->n_begun;
    idx->tail_ev[P.seq & 3].armed = false;
    if (!nq) return VROD_OK;
    hipStream_t s = P.stream;
    Timer tm(idx, P);
    P.t0 = tm.mark();

    const uint64_t N = idx->count;
    uint32_t kp = choose_kp(N, k);

    // ---- path
    int path = idx->path;
    // AUTO routing, measured on MI355X at 2M x 768 (scripts/route_probe.py): the stream scan costs
    // about one HBM pass per 8 queries (bf16: 0.63 / 0.69 / 2.0 ms at 1 / 4 / 8 queries, fp32:
    // 1.09 / 1.21 / 1.47 / 2.85 ms at 1 / 4 / 8 / 16); an MFMA batch costs the same for any
    // nq <= 256 (bf16 0.85 ms, fp32 5.97 ms: the fp32 MFMA rate is 16x lower).
    // (opt-in split pass over an fp32 corpus: ~1.5 HBM passes + 3 bf16 MFMA products, cheaper
    // than the stream scan from ~12 queries on)
    const bool can_split = idx->split_enabled && idx->dtype == VROD_DTYPE_F32 && N > 0 &&
                           (uint64_t)k + std::max<uint32_t>(32, k / 2) <= kSelectChunk / 2;
    // (5-32 queries over the planes take the skinny form where the queries' [hi | lo] fit in LDS: one HBM pass
    // over the planes, 1.30 ms at 2M x 768 against 1.34-1.36 for a stream pass of 5-8 queries and 1.85 tiled)
    const bool skinny_split = can_split && nq <= mfma_skinny_max_queries(true, 2u * idx->ldp * 2u);
    if (path == VROD_PATH_AUTO)
        path = nq <= (idx->dtype == VROD_DTYPE_BF16 ? 4u : skinny_split ? 4u : can_split ? 12u : 32u) ? VROD_PATH_STREAM : VROD_PATH_MFMA;
    bool split = can_split && path == VROD_PATH_MFMA;
    if (split && idx->planes_cap < idx->capacity) {
        // the planes are a second copy of the corpus: without room for them the handle quietly
        // keeps the fp32 pass
        if (idx->planes) { (void)hipFree(idx->planes); idx->planes = nullptr; idx->planes_cap = idx->planes_rows = 0; }
        const size_t want = idx->capacity * 2ull * idx->ldp * 2ull;
        bool room = true;
        if (!idx->split_forced) {
            // by default the planes must leave the caller a margin: 1/8 of the device or 4 GiB
            size_t free_b = 0, total_b = 0;
            room = hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b >= want + std::max<size_t>(total_b / 8, (size_t)4 << 30);
        }
        if (room && hipMalloc(&idx->planes, want) == hipSuccess) {
            idx->planes_cap = idx->capacity;
        } else {
            (void)hipGetLastError();
            idx->planes = nullptr;
            idx->split_enabled = false;
            split = false;
        }
    }
    P.split = split;
    if (split) {
        // the split pass's certificate bound is ~3x the fp32 MFMA pass's: more candidates per query
        kp = (uint32_t)std::min<uint64_t>(N, (uint64_t)k + std::max<uint32_t>(32, k / 2));
    } else if (path == VROD_PATH_MFMA && idx->metric == VROD_METRIC_COSINE) {
        // Every stage of the batched scan appends ~k' (g - 1) rows per query, and a hit costs its work-group
        // ~0.35 us (profiles/r02/mfma_experiments.md): fewer candidates, fewer hits.  The margin only has to keep
        // the k-th canonical score clear of the k'-th fast score by the error bound (1.8e-4 at d = 768 against
        // ~8e-4 per rank at 10M rows); margins 16 / 10 / 6 / 4 / 2 gave 0 / 0 / 0 / 11 / 937 failed certificates in
        // 30 720 queries and 12.58-12.65 / 12.53 / 12.51 / 13.27 / 15.44 ms per batch (1.25M-row shard: 1.82 / - /
        // 1.75 / 1.80 ms); a failed certificate costs a band pass, not a wrong result.  (The L2 bound through the
        // norm expansion is ~4x wider relative to the gaps: it keeps 16.)
        const uint32_t margin_env = debug_env().kp_margin;
        kp = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(N, kSelectChunk / 2), (uint64_t)k + (uint64_t)std::max<uint32_t>(margin_env ? margin_env : 8, k / 8) * idx->kp_boost);
    } else if (path == VROD_PATH_MFMA) {
        kp = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(N, kSelectChunk / 2), (uint64_t)k + (uint64_t)std::max<uint32_t>(16, k / 8) * idx->kp_boost);
    }
    st.kprime = kp;
    P.kp_boost_used = idx->kp_boost;
    P.N = N; P.kp = kp;
    st.path = path;
    st.split_pass = split ? 1u : 0u;
    P.path = path;

    if (N == 0) {  // empty corpus: every slot unfilled
        std::vector<uint64_t> hi((size_t)nq * k, UINT64_MAX);
        std::vector<uint32_t> hs((size_t)nq * k, kScoreNoneBits);
        HIP_TRY(hipMemcpyAsync(d_out_ids, hi.data(), hi.size() * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_out_scores, hs.data(), hs.size() * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        return VROD_OK;
    }

    // ---- prepare queries (one launch): q_f32 [nq_pad][ld] fp32 (zero padded), q_lp bf16 copy,
    // fast norms, NaN/Inf flag, max |q|^2.  No host round trip: the certificate forms its bound
    // on the device and the flag is read with the results.
    const uint32_t nq_pad = (uint32_t)round_up(nq, path == VROD_PATH_MFMA ? 256 : 8);
    P.nq_pad = nq_pad;
    P.trivial = false;
    if (!P.done) HIP_TRY(hipEventCreateWithFlags(&P.done, hipEventDisableTiming));
    VROD_TRY(P.q_f32.ensure((size_t)nq_pad * idx->ld * 4));
    void* q_lp = nullptr;
    if (idx->dtype == VROD_DTYPE_BF16) {
        VROD_TRY(P.q_lp.ensure((size_t)nq_pad * idx->ld * 2));
        q_lp = P.q_lp.p;
    }
    VROD_TRY(P.small.ensure((size_t)nq_pad * 4 * 5 + 64));  // qnorm2 | T | thr | status | readback
    float* d_qn2 = P.small.as<float>();
    float* d_T = d_qn2 + nq_pad;
    float* d_thr = d_T + nq_pad;
    uint32_t* d_status = (uint32_t*)(d_thr + nq_pad);
    uint32_t* d_readback = d_status + nq_pad;   // [nq + 4]
    if (P.h_readback_words < (size_t)nq + 4) {
        if (P.h_readback) (void)hipHostFree(P.h_readback);
        P.h_readback = nullptr;
        P.h_readback_words = 0;
        HIP_TRY(hipHostMalloc((void**)&P.h_readback, ((size_t)nq + 4 + 1024) * 4, hipHostMallocDefault));
        P.h_readback_words = (size_t)nq + 4 + 1024;
    }
    // the MFMA path's per-query list counters / thresholds live behind the lists; they are reset
    // by the same launch that prepares the queries (padding queries get the BEST score as
    // threshold so that they never append)
    const bool mfma = path == VROD_PATH_MFMA;
    const uint32_t cap = kSelectChunk;
    uint2* d_lists = nullptr;
    uint32_t* d_counts = nullptr;
    if (mfma) {
        VROD_TRY(P.lists.ensure((size_t)nq_pad * cap * 8 + (size_t)nq_pad * 4));
        d_lists = P.lists.as<uint2>();
        d_counts = (uint32_t*)((char*)P.lists.p + (size_t)nq_pad * cap * 8);
    }
    const uint32_t worst_bits = idx->metric == VROD_METRIC_COSINE ? 0xFF800000u : 0x7F800000u;  // -inf / +inf
    QueryInit qi{};
    qi.status = d_status;
    qi.counts = d_counts;
    qi.thr = mfma ? d_thr : nullptr;
    qi.thr_live_bits = worst_bits;
    qi.thr_pad_bits = worst_bits ^ 0x80000000u;
    // (flags[0..2] -- bad-value flag, max |q|^2 bits, max err bits -- are zero here: cleared by the
    // read-back launch of the slot's previous search, never by the launch that accumulates into them)
    qi.zero_words = nullptr;
    qi.n_zero_words = 0;
    // pacing counters: 8 regions of 192 words behind the scalars, one per scan launch of this search
    // and behind them the claim bits of the 4-wave kernel's work stealing, one region per scan launch likewise: both are
    // zeroed by the launch that prepares the queries (no memset node per scan launch)
    uint32_t* pace_base = P.flags + 64;
    uint32_t* claim_base = pace_base + kPaceRegions * kPaceWords;
    qi.zero_words2 = mfma ? pace_base : nullptr;
    qi.n_zero_words2 = kPaceRegions * (kPaceWords + kClaimWords);
    const size_t hist_words = 8 * 4096 + 8;   // stream path: [8][<=4096] bin counters + 8 key counters
    if (path == VROD_PATH_STREAM) {
        VROD_TRY(P.hist.ensure(hist_words * 4));
        qi.zero_words2 = P.hist.as<uint32_t>();   // first pass of 8 queries: cleared by the prep launch
        qi.n_zero_words2 = (uint32_t)hist_words;
    }
    uint32_t pace_launch = 0;
    launch_prep_queries(d_queries_raw, nq, nq_pad, idx->dim, idx->ld, idx->metric, idx->dtype, P.q_f32.as<float>(),
                        q_lp, d_qn2, &P.flags[0], &P.flags[1], qi, s);
    HIP_TRY(hipGetLastError());

    const float u = 5.9604645e-8f;  // 2^-24
    int eps_mode = 0;
    float eps_c = 0.f;

    VROD_TRY(P.cand_rows.ensure((size_t)nq * kp * 4));
    VROD_TRY(P.cand_fast.ensure((size_t)nq * kp * 4));
    VROD_TRY(P.cand_canon.ensure((size_t)nq * kp * 4));

    const double row_bytes_alg = (double)idx->ld * idx->esize;

    if (path == VROD_PATH_STREAM) {
        // -------- fast pass A: HBM-bound scan of <= 8 queries at a time, all N fast scores kept
        if (idx->metric == VROD_METRIC_COSINE) { eps_mode = 0; eps_c = 4.f * idx->dim * u; }
        else { eps_mode = 1; eps_c = 4.f * (idx->dim + 2) * u; }
        const uint64_t score_ld = round_up(N, 64);
        VROD_TRY(P.scores.ensure((size_t)8 * score_ld * 4));
        // radix select, pass 1 fused into the scan (histogram buffer prepared above)
        VROD_TRY(P.keys_a.ensure((size_t)8 * kSelectChunk * 8));
        uint32_t* d_hist = P.hist.as<uint32_t>();
        uint32_t* d_cnt = d_hist + 8 * 4096;
        // one HBM-bound scan at a time (two would only share the bandwidth and stretch each
        // other); everything behind the scan overlaps the other slot's scan
        if (!in_graph) {
            Pending& O = idx->slot[&P == &idx->slot[0] ? 1 : 0];
            HIP_TRY(hipStreamWaitEvent(s, O.scans_done, 0));
        }
        const uint32_t qpp = (uint32_t)stream_max_queries_per_pass(idx->ld);   // queries per pass: 8, fewer for long rows
        for (uint32_t q0 = 0; q0 < nq; q0 += qpp) {
            const int nqc = (int)std::min<uint32_t>(qpp, nq - q0);
            int nqp = 1;
            while (nqp < nqc) nqp <<= 1;
            if (q0 > 0) HIP_TRY(hipMemsetAsync(d_hist, 0, hist_words * 4, s));
            size_t a, b;
            tm.arm(a, b);
            launch_scan_stream(idx->corpus, idx->dtype, idx->metric, idx->ld, N,
                               P.q_f32.as<float>() + (size_t)q0 * idx->ld, nqp, P.scores.as<float>(), score_ld,
                               d_hist, kp, s);
            P.scan_pairs.push_back({a, b});
            if (q0 + qpp >= nq && !in_graph) VROD_TRY(record_scans_done(P, s));
            st.scan_launches++;
            st.scan_bytes += (double)N * row_bytes_alg;
            st.scan_flops += 2.0 * nqc * (double)N * idx->dim;
            launch_hist_compact(P.scores.as<float>(), score_ld, N, nqc, idx->metric, d_hist, stream_hist_bits(nqp), kp,
                                P.keys_a.as<uint64_t>(), kSelectChunk, d_cnt, d_status + q0, s);
            launch_keys_to_candidates(P.keys_a.as<uint64_t>(), kSelectChunk, kSelectChunk, nqc, idx->metric, kp,
                                      P.cand_rows.as<uint32_t>() + (size_t)q0 * kp, P.cand_fast.as<float>() + (size_t)q0 * kp,
                                      d_T + q0, d_cnt, s);
        }
        HIP_TRY(hipGetLastError());
    } else if (path == VROD_PATH_MFMA) {
        // -------- fast pass B: batched MFMA scan.  (1) dense sample pass over the first S rows,
        // (2) exact j-th best per query = threshold, (3) ONE filtered launch over all rows,
        // (4) keep the best k' of every list.
        if (idx->metric == VROD_METRIC_COSINE) { eps_mode = 0; eps_c = 4.f * idx->dim * u; }
        else { eps_mode = 2; eps_c = 4.f * (idx->dim + 4) * u; }
        if (split) {
            // |fast - exact dot|: representation (x = hi + lo + r, |r| <= 2^-16 |x|, the lo.lo term
            // dropped) <= 3.1 * 2^-16 |q||x|; fp32 accumulation of 3*dim exact bf16 products in any
            // order <= 4.1 * 3*dim * 2^-24 |q||x|.  (L2 = |q|^2 + |x|^2 - 2 q.x on the same dot.)
            const float repr = 3.1f * 1.52587890625e-5f;
            if (idx->metric == VROD_METRIC_COSINE) eps_c = 4.1f * 3.f * idx->dim * u + repr;
            else eps_c = 4.1f * (3.f * idx->dim + 4) * u + repr;
        }
        // The MFMA scans own the whole chip.  Two orders of the two searches in flight:
        //  late : the sample pass behind the other slot's last scan, the first filtered stage behind its read-back --
        //         the other search's tail (compaction, re-score, certificate, read-back) and this one's sample + select
        //         side by side, ~95 us per batch in which nothing else runs (profiles/r02/s_pipeline_timeline_shard.txt);
        //  early: the sample pass + select (~50 us) in front of the other slot's LAST stage (behind its second-to-last:
        //         mid_done), the first filtered stage behind that last stage: it starts the moment the other search's
        //         scans are through and runs beside that search's tail.
        // Same box, batch 1024 x 768 bf16, early against late: 1.25M rows 1.84-1.85 / 1.87 ms, 2.5M 3.52 / 3.56-3.58,
        // 5M 6.76 / 6.83, 10M 13.28-13.32 / 13.40-13.41 (-1.4 / -1.5 / -1.0 / -0.7 %).  Early is taken up to 6M rows per
        // handle: beyond, the gain is under 1 % and the sample pass squeezed beside a 7-ms stage makes that stage's own
        // launch time (what bench.py's roofline divides by) unreadable.  VROD_DEBUG_EARLY_SAMPLE=0 / 1 forces late / early.
        const int early_env = debug_env().early_sample;
        const bool early = early_env >= 0 ? early_env == 1 : N <= 6000000ull;
        Pending& O = idx->slot[&P == &idx->slot[0] ? 1 : 0];
        HIP_TRY(hipStreamWaitEvent(s, early ? O.mid_done : O.scans_done, 0));
        const void* qmat = idx->dtype == VROD_DTYPE_BF16 ? q_lp : P.q_f32.p;
        MfmaScanArgs a{};
        a.corpus = idx->corpus; a.queries = qmat; a.xnorm2 = idx->xnorm2; a.qnorm2 = d_qn2; a.thr = d_thr;
        a.lists = d_lists; a.counts = d_counts; a.cap = cap; a.ld = idx->ld; a.nq_pad = nq_pad; a.nq = nq; a.metric = idx->metric;
        VROD_TRY(P.dump.ensure(mfma_dump_bytes(idx->num_cus)));
        a.dump = P.dump.p;
        int scan_dtype = idx->dtype;
        if (split) {
            // planes of the rows added since the last batched search, and of this batch's queries
            if (idx->planes_rows < N) {
                launch_split_rows((const float*)idx->corpus + idx->planes_rows * idx->ld, N - idx->planes_rows, idx->ld, idx->ldp,
                                  (char*)idx->planes + idx->planes_rows * 2ull * idx->ldp * 2ull, false, s);
                if (round_up(N, kRowTile) > N)   // the tile padding rows stay zero
                    HIP_TRY(hipMemsetAsync((char*)idx->planes + N * 2ull * idx->ldp * 2ull, 0, (round_up(N, kRowTile) - N) * 2ull * idx->ldp * 2ull, s));
                idx->planes_rows = N;
            }
            VROD_TRY(P.q_planes.ensure((size_t)nq_pad * 3 * idx->ldp * 2));
            launch_split_rows(P.q_f32.as<float>(), nq_pad, idx->ld, idx->ldp, P.q_planes.p, true, s);
            a.corpus = idx->planes; a.queries = P.q_planes.p;
            a.ld = 3 * idx->ldp;                          // K extent = query row
            a.lda_bytes = 2 * idx->ldp * 2;               // corpus row [hi_j | lo_j] per K-tile
            a.a_wrap = 1;                                 // SPLIT form of the kernel
            scan_dtype = VROD_DTYPE_BF16;
        }
        std::vector<uint64_t> bounds{N};
        if (N > cap) {
            const uint32_t nqb = nq_pad / 256;
            const StagePlan sp = plan_stages(N, kp, cap, std::max<uint32_t>(1, (uint32_t)idx->num_cus / nqb) * kRowTile);
            bounds = sp.bounds;
            MfmaScanArgs d = a;
            d.row_begin = 0; d.row_end = sp.S;
            // Grouped form where the kernel has it: the threshold is the j-th best of the per-group bests (groups of 32
            // rows: valid -- at least j rows are that good -- and exact unless two of the j best share a group), 1/32 of
            // the dense block to write and to select from.  Only while the groups outnumber j by 8x (else: every score).
            const bool group_env = debug_env().sample_grouped;
            const uint32_t grows = group_env ? mfma_dense_group_rows(d, scan_dtype) : 0u;
            const uint32_t n_groups = grows ? (uint32_t)(round_up(sp.S, kRowTile) / grows) : 0u;
            const bool grouped = grows && (uint64_t)sp.j * 8 <= n_groups && sp.S % kRowTile == 0;   // whole tiles of real rows
            const uint32_t dense_ld = grouped ? (uint32_t)round_up(n_groups, 64) : (uint32_t)round_up(sp.S, kRowTile);
            const uint32_t n_sel = grouped ? n_groups : sp.S;
            VROD_TRY(P.scores.ensure((size_t)nq_pad * dense_ld * 4));
            d.dense_out = P.scores.as<float>(); d.dense_ld = dense_ld; d.dense_grouped = grouped;
            d.pace = pace_base; d.pace_is_zero = true; ++pace_launch;
            size_t e0, e1;
            tm.arm(e0, e1);
            launch_scan_mfma(d, scan_dtype, idx->num_cus, s);
            P.sample_pair = (int)P.scan_pairs.size();
            P.early_sample = early;
            P.scan_pairs.push_back({e0, e1});
            st.scan_launches++;
            // (the sample rows are scanned again by the first filtered stage: their time counts, their flops and
            // bytes do not -- algorithmic work is 2 * nq * N * d and N * row bytes, each row once)
            launch_sample_select(P.scores.as<float>(), dense_ld, n_sel, (int)nq, idx->metric, sp.j, d_thr, s);
        }
        HIP_TRY(hipStreamWaitEvent(s, early ? O.scans_done : O.done, 0));
        uint64_t lo = 0;
        for (size_t li = 0; li < bounds.size(); ++li) {
            while (lo < bounds[li]) {
                // a launch addresses rows relative to its first tile with 24 bits
                const uint64_t end = std::min<uint64_t>(bounds[li], lo / kRowTile * kRowTile + (1ull << 24));
                a.row_begin = (uint32_t)lo; a.row_end = (uint32_t)end;
                a.pace = pace_base + (pace_launch % kPaceRegions) * kPaceWords;
                a.pace_is_zero = pace_launch < kPaceRegions;   // later launches reuse a region: memset
                a.claims = claim_base + (pace_launch % kPaceRegions) * kClaimWords;
                a.claims_is_zero = a.pace_is_zero;
                ++pace_launch;
                size_t e0, e1;
                tm.arm(e0, e1);
                if (li + 1 == bounds.size() && end == bounds[li]) tm.arm_tail();
                launch_scan_mfma(a, scan_dtype, idx->num_cus, s);
                P.scan_pairs.push_back({e0, e1});
                st.scan_launches++;
                st.scan_bytes += (double)(end - lo / kRowTile * kRowTile) * row_bytes_alg;
                st.scan_flops += 2.0 * nq * (double)(end - lo) * idx->dim;
                lo = end;
            }
            const bool last = li + 1 == bounds.size();
            if (li + 2 == bounds.size()) { HIP_TRY(hipEventRecord(P.mid_done, s)); P.mid_recorded = true; }
            if (last) VROD_TRY(record_scans_done(P, s));
            launch_list_compact(d_lists, d_counts, cap, (int)nq, idx->metric, kp, d_thr, d_status,
                                last ? P.cand_rows.as<uint32_t>() : nullptr, last ? P.cand_fast.as<float>() : nullptr,
                                last ? d_T : nullptr, s);
        }
        HIP_TRY(hipGetLastError());
        P.pace_launches = pace_launch;
    }

    P.eps_mode = eps_mode;
    P.eps_c = eps_c;
    if (path != VROD_PATH_EXACT) {
        // -------- canonical re-score + final ordering + certificate
        launch_rescore_candidates(idx->corpus, idx->dtype, idx->metric, idx->dim, idx->ld, P.q_f32.as<float>(), (int)nq,
                                  P.cand_rows.as<uint32_t>(), kp, P.cand_canon.as<float>(), s);
        launch_final_topk(P.cand_rows.as<uint32_t>(), P.cand_fast.as<float>(), P.cand_canon.as<float>(), d_T, (int)nq, kp, k,
                          idx->metric, idmap_of(idx), eps_mode, eps_c, &P.flags[1], idx->max_xn2_bits, d_out_ids, d_out_scores,
                          d_status, (float*)&P.flags[2], s);
    }
    launch_gather_readback(d_status, nq, P.flags, idx->max_xn2_bits, d_readback, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(P.h_readback, d_readback, ((size_t)nq + 4) * 4, hipMemcpyDeviceToHost, s));
    P.t1 = tm.mark();
    if (!in_graph) HIP_TRY(hipEventRecord(P.done, s));
    return VROD_OK;
}

// Small searches are launch-bound (10k x 128, one query: ~9 launches, 80 us, of which the kernels
// are a fraction): when a slot sees the same search again -- same query / output pointers, sizes,
// corpus and workspaces -- its launches are captured into a hipGraph on the third occurrence and
// replayed as ONE graph launch from then on.  Only the stream path with a single scan pass over a
// small corpus and with profiling off qualifies; anything else, and any capture failure, takes the
// plain launches.
static int search_enqueue(vrod_index* idx, Pending& P, const float* d_queries_raw, uint32_t nq, uint32_t k,
                          uint64_t* d_out_ids, float* d_out_scores) {
    const bool graphs_on = debug_env().graph;
    const uint64_t N = idx->count;
    int path = idx->path;
    if (path == VROD_PATH_AUTO && nq <= 4) path = VROD_PATH_STREAM;
    // (measured at 10k x 128, one query: a replay costs the HOST less -- 50 vs 65 us per search with two
    // in flight -- but is no faster end to end than plain launches, 91 vs 82 us synchronous: only
    // searches begun while another one is pending, i.e. host-bound pipelines, take it)
    const bool graphable = graphs_on && !P.graph_off && idx->profiling == 0 && nq >= 1 && nq <= 8 && N > 0 && idx->n_pending() >= 1 &&
                           path == VROD_PATH_STREAM && (double)N * idx->ld * idx->esize <= 64.0 * 1048576.0;
    Pending::GraphKey key{};
    if (graphable) {
        key.q = d_queries_raw; key.oi = d_out_ids; key.os = d_out_scores; key.corpus = idx->corpus; key.xn = idx->xnorm2;
        const void* bufs[12] = {P.q_f32.p, P.q_lp.p, P.small.p, P.hist.p, P.scores.p, P.keys_a.p, P.cand_rows.p, P.cand_fast.p,
                                P.cand_canon.p, P.h_readback, P.flags, idx->max_xn2_bits};
        memcpy(key.bufs, bufs, sizeof bufs);
        key.N = N; key.id_offset = idmap_of(idx).offset; key.nq = nq; key.k = k; key.path = idx->path;
    }
    Pending& O = idx->slot[&P == &idx->slot[0] ? 1 : 0];
    hipStream_t s = P.stream;
    if (graphable && P.gexec && P.gkey_graph == key) {
        // ---- replay
        P.st = P.g_st;
        P.nq = nq; P.k = k; P.out_ids = d_out_ids; P.out_scores = d_out_scores;
        P.trivial = false; P.ev_used = 0; P.t0 = P.t1 = 0; P.scan_pairs.clear();
        P.N = N; P.kp = P.g_kp; P.path = P.g_path; P.eps_mode = P.g_eps_mode; P.eps_c = P.g_eps_c; P.split = false;
        HIP_TRY(hipStreamWaitEvent(s, O.scans_done, 0));
        HIP_TRY(hipGraphLaunch(P.gexec, s));
        P.mid_recorded = false;
        VROD_TRY(record_scans_done(P, s));
        HIP_TRY(hipEventRecord(P.done, s));
        return VROD_OK;
    }
    const bool capture = graphable && P.gkey_valid && P.gkey == key;   // seen before with these very buffers
    if (capture) {
        if (P.gexec) { (void)hipGraphExecDestroy(P.gexec); P.gexec = nullptr; }
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            int rc = search_enqueue_body(idx, P, d_queries_raw, nq, k, d_out_ids, d_out_scores, true);
            hipGraph_t g = nullptr;
            const hipError_t e1 = hipStreamEndCapture(s, &g);
            hipError_t e2 = hipErrorUnknown;
            if (rc == VROD_OK && e1 == hipSuccess && g) e2 = hipGraphInstantiate(&P.gexec, g, nullptr, nullptr, 0);
            if (g) (void)hipGraphDestroy(g);
            if (rc == VROD_OK && e1 == hipSuccess && e2 == hipSuccess) {
                P.gkey_graph = key;
                P.g_kp = P.kp; P.g_path = P.path; P.g_eps_mode = P.eps_mode; P.g_eps_c = P.eps_c; P.g_st = P.st;
                HIP_TRY(hipStreamWaitEvent(s, O.scans_done, 0));
                HIP_TRY(hipGraphLaunch(P.gexec, s));
                VROD_TRY(record_scans_done(P, s));
                HIP_TRY(hipEventRecord(P.done, s));
                return VROD_OK;
            }
            (void)hipGetLastError();
            if (P.gexec) { (void)hipGraphExecDestroy(P.gexec); P.gexec = nullptr; }
        } else {
            (void)hipGetLastError();
        }
        P.graph_off = true;   // nothing was launched: fall through to the plain launches
    }
    const int rc = search_enqueue_body(idx, P, d_queries_raw, nq, k, d_out_ids, d_out_scores, false);
    if (graphable && rc == VROD_OK) {
        // the key is taken AFTER the body: its ensure() calls may have moved a workspace
        const void* bufs[12] = {P.q_f32.p, P.q_lp.p, P.small.p, P.hist.p, P.scores.p, P.keys_a.p, P.cand_rows.p, P.cand_fast.p,
                                P.cand_canon.p, P.h_readback, P.flags, idx->max_xn2_bits};
        memcpy(key.bufs, bufs, sizeof bufs);
        P.gkey = key;
        P.gkey_valid = true;
    } else {
        P.gkey_valid = false;
    }
    return rc;
}

// ------------------------------------------------------------------ band pass
// Second chance for the queries whose certificate failed on the MFMA path (exact duplicates and near-ties around
// the k-th result: real embedding corpora are full of them).  The exact path costs one pass over the corpus per 8
// queries; a batch where most queries fail would take seconds.  Instead ONE more filtered scan, shared by all failed
// queries, collects for each the rows whose fast score lies within the error bound of c_k, the k-th canonical
// score the first pass found (kernels_select.hip band_prepare_kernel: a superset of the true top-k, boundary ties
// included); their canonical re-score and an exact select by (score, id) is the answer -- no certificate needed.
// A query whose band holds more than kBandKeep rows (thousands of exact duplicates) stays on the exact path.
// From how many failed queries on the band pass beats the exact path.  On bf16 rows (or the bf16 planes of an fp32
// corpus) a band pass of <= 64 queries is the skinny kernel, ONE HBM-bound pass over the corpus (2.8 ms at 15.4 GB),
// while the exact path's pass of 8 queries is VALU-bound (9 ms there; cfg3dup, 8 failed queries per batch: 22.2 ms
// per batch through the exact path, 15.9 through the band pass): from 2 queries on.  An fp32 corpus without planes
// scans at the fp32 matrix rate (16x lower), several exact passes long: only for batches of failures.
static const uint32_t kBandMinQueriesFast = 2, kBandMinQueriesF32 = 48;
static const uint32_t kBandKeep = kSelectChunk / 2;

static int band_pass(vrod_index* idx, Pending& P, std::vector<uint32_t>& failed, uint32_t max_qn2_bits) {
    const bool band_on = debug_env().band;
    const uint32_t nf = (uint32_t)failed.size(), k = P.k;
    const uint64_t N = P.N;
    const uint32_t min_q = (idx->dtype == VROD_DTYPE_BF16 || P.split) ? kBandMinQueriesFast : kBandMinQueriesF32;
    if (!band_on || P.path != VROD_PATH_MFMA || P.eps_mode == 1 || nf < min_q || N < k || P.nq_pad == 0) return VROD_OK;
    vrod_search_stats& st = P.st;
    hipStream_t s = P.stream;
    Timer tm(idx, P);
    const uint32_t nf_pad = (uint32_t)round_up(nf, 256);
    const uint32_t cap = kSelectChunk;
    VROD_TRY(P.band_idx.ensure((size_t)nf_pad * 4));
    VROD_TRY(P.band_q.ensure((size_t)nf_pad * idx->ld * 4));
    void* bq_lp = nullptr;
    if (idx->dtype == VROD_DTYPE_BF16) {
        VROD_TRY(P.band_q_lp.ensure((size_t)nf_pad * idx->ld * 2));
        bq_lp = P.band_q_lp.p;
    }
    VROD_TRY(P.band_small.ensure((size_t)nf_pad * 4 * 5));   // thr | qn2 | ok | resolved | status (scratch)
    float* b_thr = P.band_small.as<float>();
    float* b_qn2 = b_thr + nf_pad;
    uint32_t* b_ok = (uint32_t*)(b_qn2 + nf_pad);
    uint32_t* b_res = b_ok + nf_pad;
    uint32_t* b_status = b_res + nf_pad;
    // the list block of the search ([nq_pad][cap] entries, counters behind it) is free again: its candidates were emitted
    uint2* d_lists = P.lists.as<uint2>();
    uint32_t* d_counts = (uint32_t*)((char*)P.lists.p + (size_t)P.nq_pad * cap * 8);
    float* d_qn2_all = P.small.as<float>();
    HIP_TRY(hipMemcpyAsync(P.band_idx.p, failed.data(), (size_t)nf * 4, hipMemcpyHostToDevice, s));
    // the batch's max |q|^2 (the bound's norm): the device copy was consumed with the read-back, flags[3] holds it for this pass
    HIP_TRY(hipMemcpyAsync(&P.flags[3], &max_qn2_bits, 4, hipMemcpyHostToDevice, s));
    launch_gather_query_rows(P.q_f32.as<float>(), P.band_idx.as<uint32_t>(), nf, nf_pad, idx->ld, P.band_q.as<float>(), bq_lp, s);
    launch_band_prepare(P.band_idx.as<uint32_t>(), nf, nf_pad, P.out_scores, k, idx->metric, P.eps_mode, P.eps_c, &P.flags[3], idx->max_xn2_bits,
                        d_qn2_all, b_thr, b_qn2, d_counts, b_ok, s);
    MfmaScanArgs a{};
    a.corpus = idx->corpus; a.queries = idx->dtype == VROD_DTYPE_BF16 ? bq_lp : P.band_q.p; a.xnorm2 = idx->xnorm2; a.qnorm2 = b_qn2; a.thr = b_thr;
    a.lists = d_lists; a.counts = d_counts; a.cap = cap; a.ld = idx->ld; a.nq_pad = nf_pad; a.nq = nf; a.metric = idx->metric;
    VROD_TRY(P.dump.ensure(mfma_dump_bytes(idx->num_cus)));
    a.dump = P.dump.p;
    int scan_dtype = idx->dtype;
    if (P.split) {
        VROD_TRY(P.band_planes.ensure((size_t)nf_pad * 3 * idx->ldp * 2));
        launch_split_rows(P.band_q.as<float>(), nf_pad, idx->ld, idx->ldp, P.band_planes.p, true, s);
        a.corpus = idx->planes; a.queries = P.band_planes.p;
        a.ld = 3 * idx->ldp; a.lda_bytes = 2 * idx->ldp * 2; a.a_wrap = 1;
        scan_dtype = VROD_DTYPE_BF16;
    }
    uint32_t* pace_base = P.flags + 64;
    uint32_t* claim_base = pace_base + kPaceRegions * kPaceWords;
    const double row_bytes_alg = (double)idx->ld * idx->esize;
    for (uint64_t lo = 0; lo < N;) {
        const uint64_t end = std::min<uint64_t>(N, lo / kRowTile * kRowTile + (1ull << 24));
        a.row_begin = (uint32_t)lo; a.row_end = (uint32_t)end;
        a.pace = pace_base + (P.pace_launches % kPaceRegions) * kPaceWords;
        a.pace_is_zero = false;   // the regions were used by the search's own launches
        a.claims = claim_base + (P.pace_launches % kPaceRegions) * kClaimWords;
        a.claims_is_zero = false;
        ++P.pace_launches;
        size_t e0, e1;
        tm.arm(e0, e1);
        launch_scan_mfma(a, scan_dtype, idx->num_cus, s);
        P.scan_pairs.push_back({e0, e1});
        st.scan_launches++;
        st.scan_bytes += (double)(end - lo / kRowTile * kRowTile) * row_bytes_alg;
        st.scan_flops += 2.0 * nf * (double)(end - lo) * idx->dim;
        lo = end;
    }
    HIP_TRY(hipGetLastError());
    std::vector<uint32_t> hcnt(nf), hok(nf);
    HIP_TRY(hipMemcpyAsync(hcnt.data(), d_counts, (size_t)nf * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hok.data(), b_ok, (size_t)nf * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const uint32_t need = (uint32_t)std::min<uint64_t>(k, N);
    uint32_t maxc = 0, n_res = 0;
    std::vector<uint32_t> hres(nf_pad, 0u);
    for (uint32_t f = 0; f < nf; ++f) {
        // (a band shorter than k would mean the bound does not hold: never resolve on it)
        if (hok[f] && hcnt[f] >= need && hcnt[f] <= kBandKeep) { hres[f] = 1u; maxc = std::max(maxc, hcnt[f]); ++n_res; }
    }
    if (n_res) {
        uint32_t kpb = 32;
        while (kpb < maxc) kpb <<= 1;
        kpb = std::min<uint32_t>(std::max<uint32_t>(kpb, k), kBandKeep);
        VROD_TRY(P.cand_rows.ensure((size_t)nf * kpb * 4));
        VROD_TRY(P.cand_fast.ensure((size_t)nf * kpb * 4));
        VROD_TRY(P.cand_canon.ensure((size_t)nf * kpb * 4));
        VROD_TRY(P.band_ids.ensure((size_t)nf * k * 8));
        VROD_TRY(P.band_scores.ensure((size_t)nf * k * 4));
        HIP_TRY(hipMemcpyAsync(b_res, hres.data(), (size_t)nf_pad * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemsetAsync(b_status, 0, (size_t)nf_pad * 4, s));
        // every band row of a resolved query is kept (count <= kpb): sorted by fast score, padded with empty slots
        launch_list_compact(d_lists, d_counts, cap, (int)nf, idx->metric, kpb, b_thr, b_status, P.cand_rows.as<uint32_t>(), P.cand_fast.as<float>(),
                            b_qn2 /* T: unused */, s);
        launch_rescore_candidates(idx->corpus, idx->dtype, idx->metric, idx->dim, idx->ld, P.band_q.as<float>(), (int)nf, P.cand_rows.as<uint32_t>(), kpb,
                                  P.cand_canon.as<float>(), s);
        // flags[4]: the band's own observed |fast - canonical| (folded into max_fast_err by search_complete: the band is a
        // superset of the true top-k only while that stays inside the bound)
        HIP_TRY(hipMemsetAsync(&P.flags[4], 0, 4, s));
        launch_final_topk(P.cand_rows.as<uint32_t>(), P.cand_fast.as<float>(), P.cand_canon.as<float>(), b_qn2, (int)nf, kpb, k, idx->metric, idmap_of(idx),
                          P.eps_mode, P.eps_c, &P.flags[3], idx->max_xn2_bits, P.band_ids.as<uint64_t>(), P.band_scores.as<float>(), b_status,
                          (float*)&P.flags[4], s);
        launch_scatter_results(P.band_ids.as<uint64_t>(), P.band_scores.as<float>(), P.band_idx.as<uint32_t>(), b_res, nf, k, P.out_ids, P.out_scores, s);
        HIP_TRY(hipGetLastError());
        std::vector<uint32_t> still;
        for (uint32_t f = 0; f < nf; ++f)
            if (!hres[f]) still.push_back(failed[f]);
        failed.swap(still);
        st.band_queries = n_res;
    }
    return VROD_OK;
}

// Complete the search in slot P: wait for its status block, then run the exact path for the
// queries whose certificate failed (enqueued behind whatever the stream holds by now).
static int search_complete(vrod_index* idx, Pending& P) {
    vrod_search_stats& st = P.st;
    hipStream_t s = P.stream;
    Timer tm(idx, P);
    const uint32_t nq = P.nq, k = P.k;
    const uint64_t N = P.N;
    if (P.trivial) {
        idx->stats = st;
        return VROD_OK;
    }
    HIP_TRY(hipEventSynchronize(P.done));
    std::vector<uint32_t> hstatus(P.h_readback, P.h_readback + nq);
    uint32_t hflags[3];
    memcpy(hflags, P.h_readback + nq, 12);
    const uint32_t hmaxx = P.h_readback[nq + 3];
    if (P.path == VROD_PATH_EXACT) {
        std::fill(hstatus.begin(), hstatus.end(), 1u);
    } else {
        const int eps_mode = P.eps_mode;
        const float eps_c = P.eps_c;
        memcpy(&st.max_fast_err, &hflags[2], 4);
        float qn2, xn2;
        memcpy(&qn2, &hflags[1], 4);
        memcpy(&xn2, &hmaxx, 4);
        const float qn = std::sqrt(qn2), xn = std::sqrt(xn2);
        st.eps_bound = (eps_mode == 0 ? eps_c * qn * xn : eps_mode == 1 ? eps_c /* relative */ : eps_c * (qn + xn) * (qn + xn)) +
                       (eps_mode == 1 ? 0.f : eps_c * 2.3509887e-38f);   // + the absolute slack of the denormal range
    }
    if (hflags[0]) {  // NaN/Inf in the queries: whatever was computed is void (flag reset on device)
        idx->stats = st;
        return fail(VROD_ERR_INVALID_VALUE, "queries contain NaN or Inf");
    }

    // -------- exact path for uncertified queries: canonical score of every row, exact select.
    // Up to 8 queries share one pass over the corpus (their add chains are independent, so the
    // pass costs little more than one query's); the score block is kept under 1 GiB.
    std::vector<uint32_t> failed;
    for (uint32_t qi = 0; qi < nq; ++qi)
        if (hstatus[qi]) failed.push_back(qi);
    st.fallback_queries = (uint32_t)failed.size();
    if (!failed.empty()) VROD_TRY(band_pass(idx, P, failed, hflags[1]));   // resolves most of them with one more shared scan
    const bool many_failed = failed.size() * 8 > nq;   // what the band pass could not resolve (duplicates it handled cheaply do not count)
    if (P.split && many_failed && !idx->split_forced && ++idx->split_bad >= 2) {
        // the split pass's bound is ~3x wider than the fp32 pass's: on a corpus whose gaps sit
        // inside it (twice now) the fp32 pass is the better fast pass.  The planes are released by
        // the next search that finds the handle idle.
        idx->split_enabled = false;
    }
    if (!failed.empty()) {
        const uint64_t score_ld = round_up(N, 64);
        int gmax = rescore_all_max_queries(idx->ld);
        while (gmax > 1 && (uint64_t)gmax * score_ld * 4 > (1ull << 30)) gmax >>= 1;
        const uint32_t kx = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(k, N), kSelectChunk / 2);
        for (size_t f0 = 0; f0 < failed.size();) {
            int g = gmax;
            while ((size_t)g > failed.size() - f0) g >>= 1;
            VROD_TRY(P.scores.ensure((size_t)g * score_ld * 4));
            launch_rescore_all(idx->corpus, idx->dtype, idx->metric, idx->dim, idx->ld, P.q_f32.as<float>(), &failed[f0], g, N,
                               P.scores.as<float>(), score_ld, s);
            const uint64_t* keys; uint64_t kld, kn;
            VROD_TRY(select_chain(idx, P, P.scores.as<float>(), score_ld, N, g, kx, &keys, &kld, &kn));
            for (int i = 0; i < g; ++i) {
                const uint32_t qi = failed[f0 + i];
                launch_keys_to_output(keys + (size_t)i * kld, kn, idx->metric, k, idmap_of(idx), P.out_ids + (size_t)qi * k,
                                      P.out_scores + (size_t)qi * k, s);
            }
            HIP_TRY(hipGetLastError());
            f0 += g;
        }
    }
    if (st.fallback_queries) {
        P.t1 = tm.mark();
        HIP_TRY(hipStreamSynchronize(s));
        if (st.band_queries) {   // the band's re-score saw its own fast-vs-canonical differences
            float band_err = 0.f;
            HIP_TRY(hipMemcpy(&band_err, &P.flags[4], 4, hipMemcpyDeviceToHost));
            st.max_fast_err = std::max(st.max_fast_err, band_err);
        }
    }
    if (P.path == VROD_PATH_MFMA && !P.split && P.nq) {   // the margin follows what the certificates say (vrod_index::kp_boost)
        // (two searches may be in flight: a verdict counts only for the multiplier the search itself ran with)
        if (idx->kp_hold) --idx->kp_hold;
        if (st.fallback_queries) {
            idx->kp_clean = 0;
            if (P.kp_boost_used >= vrod_index::kMaxKpBoost) { idx->kp_boost = 1; idx->kp_hold = 256; }
            else if (!idx->kp_hold && P.kp_boost_used == idx->kp_boost) idx->kp_boost *= 2;
        } else if (P.kp_boost_used == idx->kp_boost && ++idx->kp_clean >= 64 && idx->kp_boost > 1) {
            idx->kp_boost /= 2;
            idx->kp_clean = 0;
        }
    }
    if (idx->profiling) {
        for (size_t i = 0; i < P.scan_pairs.size(); ++i) st.scan_ms += tm.pair_ms(i);
        if (P.sample_pair >= 0 && (size_t)P.sample_pair < P.scan_pairs.size()) {
            st.sample_ms = tm.pair_ms((size_t)P.sample_pair);
            st.overlap_ms = tm.sample_overlap_ms();
        }
        if (idx->profiling >= 2) st.total_ms = tm.ms(P.t0, P.t1);
    }
    idx->stats = st;
    return VROD_OK;
}

// begin = take the next slot and enqueue; end = complete the oldest slot (FIFO).
static int search_begin(vrod_index* idx, const float* d_queries_raw, uint32_t nq, uint32_t k,
                        uint64_t* d_out_ids, float* d_out_scores) {
    if (idx->n_pending() >= 2) return fail(VROD_ERR_INVALID_ARG, "two searches are already pending: call vrod_search_end first");
    Pending& P = idx->slot[idx->n_begun & 1];
    if (idx->planes && !idx->split_enabled && idx->n_pending() == 0) {   // split pass switched off: give the planes back
        (void)hipFree(idx->planes);
        idx->planes = nullptr;
        idx->planes_cap = idx->planes_rows = 0;
    }
    int rc = search_enqueue(idx, P, d_queries_raw, nq, k, d_out_ids, d_out_scores);
    if (rc != VROD_OK) {
        // a half-enqueued search: drain the stream, consume the per-search scalars, leave the slot free
        (void)hipStreamSynchronize(P.stream);
        (void)hipMemsetAsync(&P.flags[0], 0, 12, P.stream);
        (void)hipStreamSynchronize(P.stream);
        return rc;
    }
    P.active = true;
    idx->n_begun++;
    return VROD_OK;
}

static int search_end(vrod_index* idx) {
    if (idx->n_pending() == 0) return fail(VROD_ERR_INVALID_ARG, "no search is pending");
    Pending& P = idx->slot[idx->n_ended & 1];
    idx->n_ended++;
    P.active = false;
    return search_complete(idx, P);
}

// the slot (stream, raw-query buffer) the next search_begin will use
static Pending& next_slot(vrod_index* idx) { return idx->slot[idx->n_begun & 1]; }

static int run_search(vrod_index* idx, const float* d_queries_raw, uint32_t nq, uint32_t k,
                      uint64_t* d_out_ids, float* d_out_scores) {
    VROD_TRY(search_begin(idx, d_queries_raw, nq, k, d_out_ids, d_out_scores));
    return search_end(idx);
}

static int require_idle(const vrod_index* idx, const char* what) {
    if (idx->n_pending()) return fail(VROD_ERR_INVALID_ARG, "%s while a search is pending: call vrod_search_end first", what);
    return VROD_OK;
}

// The caller's stream holds the producers of the inputs and the last consumers of the output
// buffers: order it before ours without stopping the host.
static int order_after_caller(vrod_index* idx, void* stream) {
    if (!idx->caller_ev) HIP_TRY(hipEventCreateWithFlags(&idx->caller_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(idx->caller_ev, (hipStream_t)stream));
    HIP_TRY(hipStreamWaitEvent(next_slot(idx).stream, idx->caller_ev, 0));
    return VROD_OK;
}

// ------------------------------------------------------------------ composite (multi-device) handle
static const uint64_t kShardBlock = 65536;

// global rows [first, first + n) cut at block boundaries: f(shard, global_first, count)
template <typename F>
static int for_each_piece(const vrod_index* idx, uint64_t first, uint64_t n, F&& f) {
    const uint64_t G = idx->shards.size();
    uint64_t r = first;
    const uint64_t end = first + n;
    while (r < end) {
        const uint64_t blk = r / kShardBlock;
        const uint64_t m = std::min<uint64_t>(end - r, (blk + 1) * kShardBlock - r);
        VROD_TRY(f((size_t)(blk % G), r, m));
        r += m;
    }
    return VROD_OK;
}
static uint64_t local_row_of(const vrod_index* idx, uint64_t r) {
    const uint64_t G = idx->shards.size();
    return (r / (kShardBlock * G)) * kShardBlock + r % kShardBlock;
}

static int composite_add(vrod_index* idx, const float* rows, uint64_t n, bool synthetic, uint64_t seed, uint64_t first_row) {
    const uint64_t count0 = idx->count;
    for (vrod_index* sh : idx->shards) {   // what a roll-back restores (see index_add)
        VROD_TRY(set_device(sh));
        HIP_TRY(hipMemcpyAsync(&sh->flags[10], sh->max_xn2_bits, 4, hipMemcpyDeviceToDevice, sh->stream));
    }
    int rc = for_each_piece(idx, count0, n, [&](size_t g, uint64_t r, uint64_t m) {
        vrod_index* sh = idx->shards[g];
        if (sh->count != local_row_of(idx, r)) return fail(VROD_ERR_INTERNAL, "shard %zu is out of step", g);
        return index_add(sh, rows ? rows + (r - count0) * idx->dim : nullptr, m, synthetic, seed, first_row + (r - count0));
    });
    if (rc != VROD_OK) {
        // a rejected piece (NaN/Inf) rolled itself back; drop the pieces already taken by other shards
        for (size_t g = 0; g < idx->shards.size(); ++g) {
            vrod_index* sh = idx->shards[g];
            uint64_t want = 0;   // local rows of shard g among global rows [0, count0)
            (void)for_each_piece(idx, 0, count0, [&](size_t gg, uint64_t, uint64_t m) { if (gg == g) want += m; return (int)VROD_OK; });
            if (sh->count > want) {
                (void)hipSetDevice(sh->device);
                (void)hipMemsetAsync((char*)sh->corpus + want * sh->row_bytes(), 0, (sh->count - want) * sh->row_bytes(), sh->stream);
                (void)hipMemsetAsync(sh->xnorm2 + want, 0, (sh->count - want) * sizeof(float), sh->stream);
                sh->count = want;
            }
            (void)hipSetDevice(sh->device);
            (void)hipMemcpyAsync(sh->max_xn2_bits, &sh->flags[10], 4, hipMemcpyDeviceToDevice, sh->stream);
            (void)hipStreamSynchronize(sh->stream);
        }
        return rc;
    }
    idx->count = count0 + n;
    return VROD_OK;
}

static int composite_get_rows(vrod_index* idx, uint64_t first, uint64_t n, float* out_rows) {
    return for_each_piece(idx, first, n, [&](size_t g, uint64_t r, uint64_t m) {
        return vrod_index_get_rows(idx->shards[g], local_row_of(idx, r), m, out_rows + (r - first) * idx->dim);
    });
}

// ---- RCCL, bound at run time.  A multi-device handle exchanges the per-shard top-k with ONE
// ncclAllGather per device inside a group call ("single process, ncclCommInitAll, one
// stream per device").  librccl is dlopen'ed when the first such handle is created: a process that
// already holds it (PyTorch ships one with the same soname) shares that copy, and single-device
// handles never load it.
struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;   // why it could not be loaded
};
static RcclApi& rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api;
    tried = true;
    // VROD_RCCL_LIB names THE library: when it is set nothing else is tried (a path that does not load means peer copies)
    const char* env = getenv("VROD_RCCL_LIB");
    const bool only_env = env && env[0];
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !n[0] || (only_env && n != env)) continue;
        api.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api.h) break;
        api.why = dlerror();
    }
    if (!api.h) return api;
    bool ok = true;
    auto sym = [&](const char* name) { void* f = dlsym(api.h, name); if (!f) { ok = false; api.why = std::string("missing symbol ") + name; } return f; };
    api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { dlclose(api.h); api.h = nullptr; }
    return api;
}
#define NCCL_TRY(expr)                                                                          \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess)                                                                  \
            return fail(VROD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, rccl_api().GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// Groups the shards by device and opens the communicator over the distinct devices.
static int composite_open_exchange(vrod_index* idx) {
    for (size_t g = 0; g < idx->shards.size(); ++g) {
        const int dev = idx->shards[g]->device;
        size_t u = 0;
        while (u < idx->groups.size() && idx->groups[u].device != dev) ++u;
        if (u == idx->groups.size()) { idx->groups.emplace_back(); idx->groups.back().device = dev; }
        idx->shard_home.push_back({u, idx->groups[u].members.size()});
        idx->groups[u].members.push_back(g);
    }
    for (auto& G : idx->groups) {
        HIP_TRY(hipSetDevice(G.device));
        HIP_TRY(hipStreamCreateWithFlags(&G.xstream, hipStreamNonBlocking));
    }
    const char* e = getenv("VROD_RCCL");
    idx->use_rccl = !(e && e[0] == '0');   // VROD_RCCL=0: peer copies to the first device instead
    // peer access between the first device (where the lists are merged) and the others: best effort -- without it the
    // runtime stages cross-device copies through the host, slower but correct
    for (size_t u = 1; u < idx->groups.size(); ++u) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, idx->groups[0].device, idx->groups[u].device) == hipSuccess && can) {
            (void)hipSetDevice(idx->groups[0].device);
            (void)hipDeviceEnablePeerAccess(idx->groups[u].device, 0);   // (hipErrorPeerAccessAlreadyEnabled is fine)
            (void)hipSetDevice(idx->groups[u].device);
            (void)hipDeviceEnablePeerAccess(idx->groups[0].device, 0);
        }
        (void)hipGetLastError();
    }
    if (idx->use_rccl) {
        RcclApi& api = rccl_api();
        if (!api.h) {
            // no collective library: the exchange falls back to peer copies to the first device (stats.exchange = 2), said once
            static bool warned = false;
            if (!warned) fprintf(stderr, "vrod: librccl could not be loaded (%s); multi-device handles exchange their lists by peer copies "
                                         "(set VROD_RCCL_LIB to the library's path for the RCCL all-gather)\n", api.why.c_str());
            warned = true;
            idx->use_rccl = false;
            return VROD_OK;
        }
        std::vector<int> devs;
        for (auto& G : idx->groups) devs.push_back(G.device);
        std::vector<ncclComm_t> comms(devs.size(), nullptr);
        // RCCL prints a version banner on stdout at the first communicator of a process, and the host's
        // stdout is the command's result channel (SEARCHSIMILAR prints its hits there): stdout points at
        // /dev/null while the communicator is built (callers are single-threaded by contract)
        fflush(stdout);
        const int saved = dup(1), nul = open("/dev/null", O_WRONLY);
        if (saved >= 0 && nul >= 0) (void)dup2(nul, 1);
        const ncclResult_t ir = api.CommInitAll(comms.data(), (int)devs.size(), devs.data());
        fflush(stdout);
        if (saved >= 0) { (void)dup2(saved, 1); close(saved); }
        if (nul >= 0) close(nul);
        if (ir != ncclSuccess) {
            fprintf(stderr, "vrod: ncclCommInitAll over %zu devices failed (%s); this handle exchanges its lists by peer copies\n", devs.size(), api.GetErrorString(ir));
            idx->use_rccl = false;
            return VROD_OK;
        }
        for (size_t u = 0; u < devs.size(); ++u) idx->groups[u].comm = comms[u];
    }
    return VROD_OK;
}

static size_t composite_lists_per_device(const vrod_index* idx) {
    size_t m = 1;
    for (auto& G : idx->groups) m = std::max(m, G.members.size());
    return m;
}
// one device's packed result block: nk ids (u64) then nk scores (f32), padded so that every list of
// the gathered buffer starts 16-B aligned whatever the parity of nq*k
static size_t composite_block_bytes(size_t nk) { return round_up(nk * 12, 16); }

// The shards' searches of the slot being begun: queries from the host (from_host), from device
// memory of the first device (d/h pointer `queries`), or rows of the synthetic stream (`queries`
// null).  Returns with every shard's search enqueued, or with none pending on an error.
static int composite_begin(vrod_index* idx, const float* queries, bool from_host, uint64_t seed, uint64_t first_row,
                           uint32_t nq, uint32_t k, uint64_t* out_ids, float* out_scores, void* caller_stream) {
    if (idx->n_pending() >= 2) return fail(VROD_ERR_INVALID_ARG, "two searches are already pending: call vrod_search_end first");
    const uint32_t c = idx->n_begun & 1;
    vrod_index::CompPending& CP = idx->cslot[c];
    CP.nq = nq; CP.k = k; CP.out_ids = out_ids; CP.out_scores = out_scores;
    const size_t G = idx->shards.size(), U = idx->groups.size();
    const size_t nk = (size_t)nq * k, block = composite_block_bytes(nk), M = composite_lists_per_device(idx);
    const size_t qbytes = (size_t)nq * idx->dim * 4;
    idx->sh_q[c].resize(G);
    if (nq) {
        for (size_t u = 0; u < U; ++u) {
            vrod_index::DevGroup& D = idx->groups[u];
            VROD_TRY(set_device(idx->shards[D.members[0]]));
            const void* before = D.send[c].p;
            VROD_TRY(D.send[c].ensure(M * block));
            VROD_TRY(D.recv[c].ensure(U * M * block));
            if (D.members.size() < M && (D.send[c].p != before || D.filled_nk[c] != nk + 1)) {
                // list slots no shard of this device writes: "no result" entries, which the merge skips
                for (size_t m = D.members.size(); m < M; ++m)
                    launch_fill_none((uint64_t*)((char*)D.send[c].p + m * block), (float*)((char*)D.send[c].p + m * block + nk * 8), nk, D.xstream);
                HIP_TRY(hipStreamSynchronize(D.xstream));
                D.filled_nk[c] = nk + 1;
            }
        }
        if (queries && !from_host) {   // the caller's stream (first device) produced the queries
            VROD_TRY(set_device(idx->shards[idx->groups[0].members[0]]));
            if (!idx->caller_ev) HIP_TRY(hipEventCreateWithFlags(&idx->caller_ev, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(idx->caller_ev, (hipStream_t)caller_stream));
        }
    }
    // Device outputs (both device forms, the synthetic one included): whatever the caller's stream still does with the
    // buffers of an earlier batch -- a kernel reading its results -- comes before the merge that overwrites them.
    // (The single-device path has the same ordering through order_after_caller.)
    CP.ordered = false;
    if (nq && !from_host && out_ids) {
        VROD_TRY(set_device(idx->shards[idx->groups[0].members[0]]));
        if (!CP.caller_ev) HIP_TRY(hipEventCreateWithFlags(&CP.caller_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(CP.caller_ev, (hipStream_t)caller_stream));
        CP.ordered = true;
    }
    int rc = VROD_OK;
    size_t begun = 0;
    for (size_t g = 0; g < G && rc == VROD_OK; ++g) {
        vrod_index* sh = idx->shards[g];
        const auto [u, m] = idx->shard_home[g];
        if ((rc = set_device(sh)) != VROD_OK) break;
        sh->path = idx->path;
        sh->profiling = idx->profiling;
        sh->deal = IdMap{idx->id_offset, (uint32_t)kShardBlock, (uint32_t)g, (uint32_t)G};
        uint64_t* oi = nq ? (uint64_t*)((char*)idx->groups[u].send[c].p + m * block) : nullptr;
        float* os = nq ? (float*)((char*)idx->groups[u].send[c].p + m * block + nk * 8) : nullptr;
        hipStream_t ss = next_slot(sh).stream;
        const float* q = nullptr;
        if (nq) {
            if ((rc = idx->sh_q[c][g].ensure(std::max<size_t>(qbytes, 4))) != VROD_OK) break;
            q = idx->sh_q[c][g].as<float>();
            hipError_t e = hipSuccess;
            if (!queries) {
                launch_synth_rows(seed, first_row, nq, idx->dim, idx->sh_q[c][g].as<float>(), ss);
            } else if (from_host) {
                e = hipMemcpyAsync(idx->sh_q[c][g].p, queries, qbytes, hipMemcpyHostToDevice, ss);
            } else {
                e = hipStreamWaitEvent(ss, idx->caller_ev, 0);
                if (e == hipSuccess) e = hipMemcpyAsync(idx->sh_q[c][g].p, queries, qbytes, hipMemcpyDefault, ss);
            }
            if (e != hipSuccess) { rc = fail(VROD_ERR_HIP, "queries to device %d: %s", sh->device, hipGetErrorString(e)); break; }
        }
        rc = search_begin(sh, q, nq, k, oi, os);
        if (rc == VROD_OK) ++begun;
    }
    if (rc != VROD_OK) {   // a search begun must be ended: nothing stays pending behind an error
        const std::string why = g_last_error;
        for (size_t g = 0; g < begun; ++g) { (void)set_device(idx->shards[g]); (void)search_end(idx->shards[g]); }
        g_last_error = why;
        return rc;
    }
    idx->n_begun++;
    return VROD_OK;
}

// Complete the oldest composite search: every shard's search, then the exchange (RCCL all-gather of
// one packed block per device, or peer copies with VROD_RCCL=0) and the merge on the first device.
// to_host: the caller's outputs are host memory (vrod_search).
static int composite_end(vrod_index* idx, uint64_t* host_ids, float* host_scores) {
    if (idx->n_pending() == 0) return fail(VROD_ERR_INVALID_ARG, "no search is pending");
    const uint32_t c = idx->n_ended & 1;
    idx->n_ended++;
    vrod_index::CompPending& CP = idx->cslot[c];
    const uint32_t nq = CP.nq, k = CP.k;
    const size_t G = idx->shards.size(), U = idx->groups.size();
    idx->stats = vrod_search_stats{};
    idx->stats.nq = nq;
    idx->stats.k = k;
    int rc = VROD_OK;
    for (size_t g = 0; g < G; ++g) {
        vrod_index* sh = idx->shards[g];
        const int r0 = set_device(sh);
        const int r = r0 != VROD_OK ? r0 : search_end(sh);     // every begun search is ended, whatever the others return
        if (r != VROD_OK && rc == VROD_OK) rc = r;
        const vrod_search_stats& st = sh->stats;
        idx->stats.path = st.path; idx->stats.kprime = st.kprime;
        idx->stats.scan_launches += st.scan_launches;
        idx->stats.fallback_queries += st.fallback_queries;
        idx->stats.band_queries += st.band_queries;
        idx->stats.split_pass |= st.split_pass;
        if (st.scan_ms > idx->stats.scan_ms) { idx->stats.sample_ms = st.sample_ms; idx->stats.overlap_ms = st.overlap_ms; }   // of the shard whose scans took longest
        idx->stats.scan_ms = std::max(idx->stats.scan_ms, st.scan_ms);
        idx->stats.total_ms = std::max(idx->stats.total_ms, st.total_ms);
        idx->stats.scan_bytes += st.scan_bytes; idx->stats.scan_flops += st.scan_flops;
        idx->stats.max_fast_err = std::max(idx->stats.max_fast_err, st.max_fast_err);
        idx->stats.eps_bound = std::max(idx->stats.eps_bound, st.eps_bound);
    }
    idx->stats.exchange = idx->use_rccl ? 1u : 2u;
    if (rc != VROD_OK || !nq) return rc;
    // every shard's list is complete in its device's send block (the host has seen each search end)
    const size_t nk = (size_t)nq * k, block = composite_block_bytes(nk), M = composite_lists_per_device(idx);
    vrod_index::DevGroup& D0 = idx->groups[0];
    if (idx->use_rccl) {
        RcclApi& api = rccl_api();
        NCCL_TRY(api.GroupStart());
        for (size_t u = 0; u < U; ++u) {
            vrod_index::DevGroup& D = idx->groups[u];
            const ncclResult_t r = api.AllGather(D.send[c].p, D.recv[c].p, M * block, ncclUint8, D.comm, D.xstream);
            if (r != ncclSuccess) { (void)api.GroupEnd(); return fail(VROD_ERR_HIP, "ncclAllGather failed: %s", api.GetErrorString(r)); }
        }
        NCCL_TRY(api.GroupEnd());
    } else {
        HIP_TRY(hipSetDevice(D0.device));
        for (size_t u = 0; u < U; ++u)
            HIP_TRY(hipMemcpyPeerAsync((char*)D0.recv[c].p + u * M * block, D0.device, idx->groups[u].send[c].p, idx->groups[u].device, M * block, D0.xstream));
    }
    HIP_TRY(hipSetDevice(D0.device));
    if (CP.ordered) HIP_TRY(hipStreamWaitEvent(D0.xstream, CP.caller_ev, 0));   // the caller's earlier use of the output buffers
    uint64_t* oi = CP.out_ids;
    float* os = CP.out_scores;
    if (host_ids) {
        VROD_TRY(idx->out_ids.ensure(nk * 8));
        VROD_TRY(idx->out_scores.ensure(nk * 4));
        oi = idx->out_ids.as<uint64_t>();
        os = idx->out_scores.as<float>();
    }
    launch_merge_topk(idx->metric, (const uint64_t*)D0.recv[c].p, (const float*)((const char*)D0.recv[c].p + nk * 8), block / 8, block / 4,
                      (uint32_t)(U * M), nq, k, oi, os, D0.xstream);
    HIP_TRY(hipGetLastError());
    if (host_ids) {
        HIP_TRY(hipMemcpyAsync(host_ids, oi, nk * 8, hipMemcpyDeviceToHost, D0.xstream));
        HIP_TRY(hipMemcpyAsync(host_scores, os, nk * 4, hipMemcpyDeviceToHost, D0.xstream));
    }
    // the other devices' all-gathers read their send blocks: done before the slot is reused
    for (size_t u = U; u-- > 0;) {
        HIP_TRY(hipSetDevice(idx->groups[u].device));
        HIP_TRY(hipStreamSynchronize(idx->groups[u].xstream));
    }
    return VROD_OK;
}

static int composite_search(vrod_index* idx, const float* queries, bool from_host, uint64_t seed, uint64_t first_row, uint32_t nq, uint32_t k,
                            uint64_t* out_ids, float* out_scores, void* caller_stream) {
    if (idx->n_pending()) return fail(VROD_ERR_INVALID_ARG, "a synchronous search while a search is pending: call vrod_search_end first");
    VROD_TRY(composite_begin(idx, queries, from_host, seed, first_row, nq, k, from_host ? nullptr : out_ids, from_host ? nullptr : out_scores, caller_stream));
    return composite_end(idx, from_host ? out_ids : nullptr, from_host ? out_scores : nullptr);
}

// ------------------------------------------------------------------ C ABI
extern "C" {

const char* vrod_last_error(void) { return g_last_error.c_str(); }
#ifndef VROD_HIPCC_VERSION
#define VROD_HIPCC_VERSION "unknown"
#endif
// names the compiler that built the device code: the 4-wave scan's register audit was run on THAT compiler's output
const char* vrod_version(void) { return "vrod_amd 0.3 (gfx950; hipcc " VROD_HIPCC_VERSION "; w4 accumulator audit passed at build)"; }

int vrod_index_create(vrod_index** out, uint32_t dim, int dtype, int metric, const int* device_ids,
                      int n_devices) {
    if (!out) return fail(VROD_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (dim == 0 || dim > VROD_MAX_DIM) return fail(VROD_ERR_INVALID_ARG, "dim must be in 1..%u", VROD_MAX_DIM);
    if (dtype != VROD_DTYPE_F32 && dtype != VROD_DTYPE_BF16) return fail(VROD_ERR_INVALID_ARG, "bad dtype %d", dtype);
    if (metric != VROD_METRIC_COSINE && metric != VROD_METRIC_L2) return fail(VROD_ERR_INVALID_ARG, "bad metric %d", metric);
    if (n_devices < 0 || (n_devices > 0 && !device_ids)) return fail(VROD_ERR_INVALID_ARG, "bad device list");
    if (n_devices > 1) {
        if (n_devices > 64) return fail(VROD_ERR_INVALID_ARG, "at most 64 devices per handle");
        vrod_index* c = new (std::nothrow) vrod_index();
        if (!c) return fail(VROD_ERR_OUT_OF_MEMORY, "host allocation failed");
        c->dim = dim; c->dtype = dtype; c->metric = metric;
        for (int g = 0; g < n_devices; ++g) {
            vrod_index* sh = nullptr;
            const int rc = vrod_index_create(&sh, dim, dtype, metric, &device_ids[g], 1);
            if (rc != VROD_OK) { vrod_index_destroy(c); return rc; }
            c->shards.push_back(sh);
        }
        c->device = c->shards[0]->device;
        c->ld = c->shards[0]->ld; c->esize = c->shards[0]->esize;
        const int rc = composite_open_exchange(c);
        if (rc != VROD_OK) { const std::string why = g_last_error; vrod_index_destroy(c); g_last_error = why; return rc; }
        *out = c;
        return VROD_OK;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(VROD_ERR_NO_DEVICE, "no HIP device visible: libvrod_hip has no CPU fallback");
    const int dev = n_devices == 1 ? device_ids[0] : 0;
    if (dev < 0 || dev >= ndev) return fail(VROD_ERR_INVALID_ARG, "device %d out of range (have %d)", dev, ndev);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(VROD_ERR_NO_DEVICE, "device %d is %s; libvrod_hip is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
    vrod_index* idx = new (std::nothrow) vrod_index();
    if (!idx) return fail(VROD_ERR_OUT_OF_MEMORY, "host allocation failed");
    idx->device = dev;
    idx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    idx->dim = dim;
    idx->dtype = dtype;
    idx->metric = metric;
    idx->esize = dtype == VROD_DTYPE_BF16 ? 2 : 4;
    // rows are padded to whole 128-B lines: 64 bf16 / 32 fp32 elements
    idx->ld = (uint32_t)round_up(dim, dtype == VROD_DTYPE_BF16 ? 64 : 32);
    idx->ldp = (uint32_t)round_up(dim, 64);
    {
        const char* e = getenv("VROD_F32_SPLIT");
        idx->split_enabled = dtype == VROD_DTYPE_F32 && !(e && e[0] == '0');
        idx->split_forced = idx->split_enabled && e && e[0] == '1';
    }
    int rc = VROD_OK;
    do {
        if (hipSetDevice(dev) != hipSuccess) { rc = fail(VROD_ERR_HIP, "hipSetDevice failed"); break; }
        if (hipStreamCreateWithFlags(&idx->stream, hipStreamNonBlocking) != hipSuccess) { rc = fail(VROD_ERR_HIP, "hipStreamCreate failed"); break; }
        if (hipMalloc((void**)&idx->flags, 8192) != hipSuccess) { rc = fail(VROD_ERR_OUT_OF_MEMORY, "hipMalloc failed"); break; }
        if (hipMemset(idx->flags, 0, 8192) != hipSuccess) { rc = fail(VROD_ERR_HIP, "hipMemset failed"); break; }
        idx->max_xn2_bits = idx->flags + 8;
        for (Pending& P : idx->slot) {
            if (hipStreamCreateWithFlags(&P.stream, hipStreamNonBlocking) != hipSuccess) { rc = fail(VROD_ERR_HIP, "hipStreamCreate failed"); break; }
            if (hipMalloc((void**)&P.flags, kSlotFlagBytes) != hipSuccess) { rc = fail(VROD_ERR_OUT_OF_MEMORY, "hipMalloc failed"); break; }
            if (hipMemset(P.flags, 0, kSlotFlagBytes) != hipSuccess) { rc = fail(VROD_ERR_HIP, "hipMemset failed"); break; }
            if (hipEventCreateWithFlags(&P.done, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&P.scans_done, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&P.mid_done, hipEventDisableTiming) != hipSuccess) { rc = fail(VROD_ERR_HIP, "hipEventCreate failed"); break; }
        }
    } while (0);
    if (rc != VROD_OK) { vrod_index_destroy(idx); return rc; }
    *out = idx;
    return VROD_OK;
}

int vrod_index_destroy(vrod_index* idx) {
    if (!idx) return VROD_OK;
    if (idx->composite()) {
        while (idx->n_pending()) (void)composite_end(idx, nullptr, nullptr);   // a begun search is ended before its buffers go
        for (size_t g = 0; g < idx->shards.size(); ++g) {
            (void)hipSetDevice(idx->shards[g]->device);
            for (int c = 0; c < 2; ++c)
                if (g < idx->sh_q[c].size()) idx->sh_q[c][g].release();
        }
        for (auto& D : idx->groups) {
            (void)hipSetDevice(D.device);
            if (D.xstream) (void)hipStreamSynchronize(D.xstream);
            if (D.comm) (void)rccl_api().CommDestroy(D.comm);
            for (int c = 0; c < 2; ++c) { D.send[c].release(); D.recv[c].release(); }
            if (D.xstream) (void)hipStreamDestroy(D.xstream);
        }
        (void)hipSetDevice(idx->device);
        if (idx->caller_ev) (void)hipEventDestroy(idx->caller_ev);
        for (auto& CP : idx->cslot) if (CP.caller_ev) (void)hipEventDestroy(CP.caller_ev);
        idx->out_ids.release(); idx->out_scores.release(); idx->raw_stage.release();
        for (vrod_index* sh : idx->shards) vrod_index_destroy(sh);
        delete idx;
        return VROD_OK;
    }
    (void)hipSetDevice(idx->device);
    if (idx->stream) (void)hipStreamSynchronize(idx->stream);
    for (Pending& P : idx->slot)
        if (P.stream) (void)hipStreamSynchronize(P.stream);
    for (DevBuf* b : {&idx->raw_stage, &idx->nrm_ws, &idx->out_ids, &idx->out_scores}) b->release();
    for (Pending& P : idx->slot) {
        for (DevBuf* b : {&P.q_raw, &P.q_lp, &P.scores, &P.keys_a, &P.keys_b, &P.lists, &P.small, &P.hist, &P.cand_rows, &P.cand_fast, &P.cand_canon})
            b->release();
        if (P.flags) (void)hipFree(P.flags);
        if (P.stream) (void)hipStreamDestroy(P.stream);
        P.q_f32.release();
        P.q_planes.release();
        P.dump.release();
        for (DevBuf* b : {&P.band_idx, &P.band_q, &P.band_q_lp, &P.band_planes, &P.band_small, &P.band_ids, &P.band_scores}) b->release();
        if (P.gexec) (void)hipGraphExecDestroy(P.gexec);
        for (hipEvent_t e : P.ev) (void)hipEventDestroy(e);
        for (auto& T : idx->tail_ev) { if (T.start) (void)hipEventDestroy(T.start); if (T.stop) (void)hipEventDestroy(T.stop); T = {}; }
        if (P.done) (void)hipEventDestroy(P.done);
        if (P.scans_done) (void)hipEventDestroy(P.scans_done);
        if (P.mid_done) (void)hipEventDestroy(P.mid_done);
        if (P.h_readback) (void)hipHostFree(P.h_readback);
    }
    if (idx->caller_ev) (void)hipEventDestroy(idx->caller_ev);
    if (idx->corpus) (void)hipFree(idx->corpus);
    if (idx->planes) (void)hipFree(idx->planes);
    if (idx->xnorm2) (void)hipFree(idx->xnorm2);
    if (idx->flags) (void)hipFree(idx->flags);
    if (idx->stream) (void)hipStreamDestroy(idx->stream);
    delete idx;
    return VROD_OK;
}

int vrod_index_reserve(vrod_index* idx, uint64_t n_rows) {
    if (!idx) return fail(VROD_ERR_INVALID_ARG, "idx is null");
    if (idx->composite()) {
        const uint64_t G = idx->shards.size();
        const uint64_t per = (n_rows / (kShardBlock * G) + 1) * kShardBlock;   // whole blocks per shard
        for (vrod_index* sh : idx->shards) VROD_TRY(vrod_index_reserve(sh, per));
        return VROD_OK;
    }
    VROD_TRY(require_idle(idx, "vrod_index_reserve"));
    VROD_TRY(set_device(idx));
    return index_reserve(idx, n_rows);
}

int vrod_index_add(vrod_index* idx, const float* rows, uint64_t n) {
    if (!idx || (!rows && n)) return fail(VROD_ERR_INVALID_ARG, "null argument");
    if (idx->composite()) return composite_add(idx, rows, n, false, 0, 0);
    VROD_TRY(require_idle(idx, "vrod_index_add"));
    return index_add(idx, rows, n, false, 0, 0);
}

int vrod_index_add_synthetic(vrod_index* idx, uint64_t seed, uint64_t first_row, uint64_t n) {
    if (!idx) return fail(VROD_ERR_INVALID_ARG, "idx is null");
    if (idx->composite()) return composite_add(idx, nullptr, n, true, seed, first_row);
    VROD_TRY(require_idle(idx, "vrod_index_add_synthetic"));
    return index_add(idx, nullptr, n, true, seed, first_row);
}

int vrod_index_count(const vrod_index* idx, uint64_t* out_count) {
    if (!idx || !out_count) return fail(VROD_ERR_INVALID_ARG, "null argument");
    *out_count = idx->count;
    return VROD_OK;
}

int vrod_index_set_id_offset(vrod_index* idx, uint64_t offset) {
    if (!idx) return fail(VROD_ERR_INVALID_ARG, "idx is null");
    idx->id_offset = offset;
    return VROD_OK;
}

int vrod_index_get_rows(vrod_index* idx, uint64_t first, uint64_t n, float* out_rows) {
    if (!idx || (!out_rows && n)) return fail(VROD_ERR_INVALID_ARG, "null argument");
    if (first + n > idx->count) return fail(VROD_ERR_INVALID_ARG, "rows [%llu, %llu) out of range", (unsigned long long)first, (unsigned long long)(first + n));
    if (!n) return VROD_OK;
    if (idx->composite()) return composite_get_rows(idx, first, n, out_rows);
    VROD_TRY(require_idle(idx, "vrod_index_get_rows"));
    VROD_TRY(set_device(idx));
    VROD_TRY(idx->raw_stage.ensure(n * idx->dim * 4));
    launch_rows_get((const char*)idx->corpus + first * idx->row_bytes(), idx->dtype, n, idx->dim, idx->ld, idx->raw_stage.as<float>(), idx->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_rows, idx->raw_stage.p, n * idx->dim * 4, hipMemcpyDeviceToHost, idx->stream));
    HIP_TRY(hipStreamSynchronize(idx->stream));
    return VROD_OK;
}

static int check_search_args(vrod_index* idx, const void* q, uint32_t nq, uint32_t k, const void* oi, const void* os) {
    if (!idx) return fail(VROD_ERR_INVALID_ARG, "idx is null");
    if (nq && (!q || !oi || !os)) return fail(VROD_ERR_INVALID_ARG, "null buffer");
    if (k == 0 || k > VROD_MAX_K) return fail(VROD_ERR_INVALID_ARG, "k must be in 1..%u", VROD_MAX_K);
    return VROD_OK;
}

int vrod_search_device(vrod_index* idx, const float* d_queries, uint32_t nq, uint32_t k,
                       uint64_t* d_out_ids, float* d_out_scores, void* stream) {
    VROD_TRY(check_search_args(idx, d_queries, nq, k, d_out_ids, d_out_scores));
    if (idx->composite())   // pointers on the first device of the handle
        return composite_search(idx, d_queries, false, 0, 0, nq, k, d_out_ids, d_out_scores, stream);
    VROD_TRY(require_idle(idx, "vrod_search_device"));
    VROD_TRY(set_device(idx));
    VROD_TRY(order_after_caller(idx, stream));   // the caller's inputs are ready
    return run_search(idx, d_queries, nq, k, d_out_ids, d_out_scores);
}

int vrod_search_begin_device(vrod_index* idx, const float* d_queries, uint32_t nq, uint32_t k,
                             uint64_t* d_out_ids, float* d_out_scores, void* stream) {
    VROD_TRY(check_search_args(idx, d_queries, nq, k, d_out_ids, d_out_scores));
    if (idx->composite()) return composite_begin(idx, d_queries, false, 0, 0, nq, k, d_out_ids, d_out_scores, stream);
    VROD_TRY(set_device(idx));
    VROD_TRY(order_after_caller(idx, stream));
    return search_begin(idx, d_queries, nq, k, d_out_ids, d_out_scores);
}

int vrod_search_begin_synthetic_device(vrod_index* idx, uint64_t seed, uint64_t first_row, uint32_t nq,
                                       uint32_t k, uint64_t* d_out_ids, float* d_out_scores, void* stream) {
    VROD_TRY(check_search_args(idx, (void*)1, nq, k, d_out_ids, d_out_scores));
    if (idx->composite()) return composite_begin(idx, nullptr, false, seed, first_row, nq, k, d_out_ids, d_out_scores, stream);
    if (idx->n_pending() >= 2) return fail(VROD_ERR_INVALID_ARG, "two searches are already pending: call vrod_search_end first");
    VROD_TRY(set_device(idx));
    VROD_TRY(order_after_caller(idx, stream));
    // the raw queries are consumed by the prepare launch of this same search: one shared buffer,
    // stream-ordered (a regrowth frees it through hipFree, which waits for the device)
    Pending& P = next_slot(idx);
    VROD_TRY(P.q_raw.ensure((size_t)std::max<uint32_t>(nq, 1) * idx->dim * 4));
    launch_synth_rows(seed, first_row, nq, idx->dim, P.q_raw.as<float>(), P.stream);
    return search_begin(idx, P.q_raw.as<float>(), nq, k, d_out_ids, d_out_scores);
}

int vrod_search_end(vrod_index* idx) {
    if (!idx) return fail(VROD_ERR_INVALID_ARG, "idx is null");
    if (idx->composite()) return composite_end(idx, nullptr, nullptr);
    VROD_TRY(set_device(idx));
    return search_end(idx);
}

int vrod_search_pending(const vrod_index* idx, uint32_t* out_pending) {
    if (!idx || !out_pending) return fail(VROD_ERR_INVALID_ARG, "null argument");
    *out_pending = idx->n_pending();
    return VROD_OK;
}

int vrod_search_synthetic_device(vrod_index* idx, uint64_t seed, uint64_t first_row, uint32_t nq,
                                 uint32_t k, uint64_t* d_out_ids, float* d_out_scores, void* stream) {
    VROD_TRY(check_search_args(idx, (void*)1, nq, k, d_out_ids, d_out_scores));
    if (idx->composite())   // every device generates the batch's queries itself: nothing to copy
        return composite_search(idx, nullptr, false, seed, first_row, nq, k, d_out_ids, d_out_scores, stream);
    VROD_TRY(require_idle(idx, "vrod_search_synthetic_device"));
    VROD_TRY(set_device(idx));
    VROD_TRY(order_after_caller(idx, stream));
    Pending& P = next_slot(idx);
    VROD_TRY(P.q_raw.ensure((size_t)std::max<uint32_t>(nq, 1) * idx->dim * 4));
    launch_synth_rows(seed, first_row, nq, idx->dim, P.q_raw.as<float>(), P.stream);
    return run_search(idx, P.q_raw.as<float>(), nq, k, d_out_ids, d_out_scores);
}

int vrod_search(vrod_index* idx, const float* queries, uint32_t nq, uint32_t k, uint64_t* out_ids,
                float* out_scores) {
    VROD_TRY(check_search_args(idx, queries, nq, k, out_ids, out_scores));
    if (!nq) return VROD_OK;
    if (idx->composite()) return composite_search(idx, queries, true, 0, 0, nq, k, out_ids, out_scores, nullptr);
    VROD_TRY(require_idle(idx, "vrod_search"));
    VROD_TRY(set_device(idx));
    Pending& P = next_slot(idx);
    VROD_TRY(P.q_raw.ensure((size_t)nq * idx->dim * 4));
    VROD_TRY(idx->out_ids.ensure((size_t)nq * k * 8));
    VROD_TRY(idx->out_scores.ensure((size_t)nq * k * 4));
    HIP_TRY(hipMemcpyAsync(P.q_raw.p, queries, (size_t)nq * idx->dim * 4, hipMemcpyHostToDevice, P.stream));
    VROD_TRY(run_search(idx, P.q_raw.as<float>(), nq, k, idx->out_ids.as<uint64_t>(), idx->out_scores.as<float>()));
    HIP_TRY(hipMemcpyAsync(out_ids, idx->out_ids.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost, idx->stream));
    HIP_TRY(hipMemcpyAsync(out_scores, idx->out_scores.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, idx->stream));
    HIP_TRY(hipStreamSynchronize(idx->stream));
    return VROD_OK;
}

int vrod_merge_topk_device(int device, int metric, const uint64_t* d_ids, const float* d_scores,
                           uint32_t n_lists, uint32_t nq, uint32_t k, uint64_t* d_out_ids,
                           float* d_out_scores, void* stream) {
    if ((nq && k && n_lists) && (!d_ids || !d_scores || !d_out_ids || !d_out_scores)) return fail(VROD_ERR_INVALID_ARG, "null buffer");
    if (metric != VROD_METRIC_COSINE && metric != VROD_METRIC_L2) return fail(VROD_ERR_INVALID_ARG, "bad metric %d", metric);
    HIP_TRY(hipSetDevice(device));
    launch_merge_topk(metric, d_ids, d_scores, (uint64_t)nq * k, (uint64_t)nq * k, n_lists, nq, k, d_out_ids, d_out_scores, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return VROD_OK;
}

int vrod_merge_topk_packed_device(int device, int metric, const void* d_packed, uint32_t n_lists, uint32_t nq,
                                  uint32_t k, uint64_t* d_out_ids, float* d_out_scores, void* stream) {
    if ((nq && k && n_lists) && (!d_packed || !d_out_ids || !d_out_scores)) return fail(VROD_ERR_INVALID_ARG, "null buffer");
    if (metric != VROD_METRIC_COSINE && metric != VROD_METRIC_L2) return fail(VROD_ERR_INVALID_ARG, "bad metric %d", metric);
    HIP_TRY(hipSetDevice(device));
    // one rank's block = nq*k ids (u64) followed by nq*k scores (f32): 12*nq*k bytes, 8-B aligned
    // as long as nq*k is even; the stride is given in elements of each array
    const uint64_t block_bytes = (uint64_t)nq * k * 12;
    if (block_bytes % 8 != 0) return fail(VROD_ERR_INVALID_ARG, "nq*k must be even for the packed layout");
    const uint64_t* ids = (const uint64_t*)d_packed;
    const float* scores = (const float*)((const char*)d_packed + (uint64_t)nq * k * 8);
    launch_merge_topk(metric, ids, scores, block_bytes / 8, block_bytes / 4, n_lists, nq, k, d_out_ids, d_out_scores, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return VROD_OK;
}

int vrod_index_set_path(vrod_index* idx, int path) {
    if (!idx || path < VROD_PATH_AUTO || path > VROD_PATH_EXACT) return fail(VROD_ERR_INVALID_ARG, "bad path");
    if (!idx->composite()) VROD_TRY(require_idle(idx, "vrod_index_set_path"));
    idx->path = path;
    return VROD_OK;
}

int vrod_index_set_profiling(vrod_index* idx, int on) {
    if (!idx) return fail(VROD_ERR_INVALID_ARG, "idx is null");
    idx->profiling = on < 0 ? 0 : on > 2 ? 2 : on;
    return VROD_OK;
}

int vrod_index_last_stats(const vrod_index* idx, vrod_search_stats* out) {
    if (!idx || !out) return fail(VROD_ERR_INVALID_ARG, "null argument");
    *out = idx->stats;
    return VROD_OK;
}

int vrod_index_shard_stats(const vrod_index* idx, uint32_t shard, int* out_device, vrod_search_stats* out) {
    if (!idx || !out) return fail(VROD_ERR_INVALID_ARG, "null argument");
    const size_t n = idx->composite() ? idx->shards.size() : 1;
    if (shard >= n) return fail(VROD_ERR_INVALID_ARG, "shard %u of a handle with %zu", shard, n);
    const vrod_index* sh = idx->composite() ? idx->shards[shard] : idx;
    *out = sh->stats;
    if (out_device) *out_device = sh->device;
    return VROD_OK;
}

int vrod_synth_rows_device(int device, uint64_t seed, uint64_t first_row, uint64_t n, uint32_t dim,
                           float* d_out, void* stream) {
    if (n && !d_out) return fail(VROD_ERR_INVALID_ARG, "null buffer");
    HIP_TRY(hipSetDevice(device));
    launch_synth_rows(seed, first_row, n, dim, d_out, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return VROD_OK;
}

}  // extern "C"
