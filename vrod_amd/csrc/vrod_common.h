// vrod_common.h -- shared device/host helpers for libvrod_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vrod {

constexpr int kWave = 64;             // CDNA wavefront
constexpr uint32_t kRowTile = 256;    // corpus capacity granularity (rows)
constexpr uint32_t kScoreNoneBits = 0x7FC00000u;

enum : int { DT_F32 = 0, DT_BF16 = 1 };
enum : int { M_COSINE = 0, M_L2 = 1 };

typedef uint16_t bf16_t;

// ---- splitmix64 / synthetic stream (oracle/vrod_oracle.c: orc_synth_int) ----
__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline int32_t synth_int(uint64_t key, uint64_t elem_index) {
    uint64_t h = splitmix64(key ^ elem_index);
    int32_t s = (int32_t)(h & 0xFFFF) + (int32_t)((h >> 16) & 0xFFFF) +
                (int32_t)((h >> 32) & 0xFFFF) + (int32_t)(h >> 48);
    return s - 131070;
}

// ---- bf16 <-> f32 ----
__device__ inline float bf16_to_f32(bf16_t h) { return __uint_as_float((uint32_t)h << 16); }
// round to nearest even on the bits (inputs are finite: NaN/Inf are rejected at the ABI)
__device__ inline bf16_t f32_to_bf16_rne(float x) {
    uint32_t u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// ---- order-preserving keys: LARGER key == BETTER result ----
// COSINE (higher score better): monotone map of the float.  L2 (lower better): its
// complement.  NaN ranks worst (key 0).  Ties on the score are broken by smaller id:
// the low word holds ~row so that a larger composite key means a smaller id.
__device__ inline uint32_t flip_f32(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// -0.0 and +0.0 compare equal in the spec (ties -> id), so -0 is folded onto +0 first.
template <int METRIC>
__device__ inline uint32_t score_key(float s) {
    if (s != s) return 0u;
    uint32_t k = flip_f32(s + 0.0f);
    return METRIC == M_COSINE ? k : ~k;
}
__device__ inline uint32_t score_key_rt(float s, int metric) {
    if (s != s) return 0u;
    uint32_t k = flip_f32(s + 0.0f);
    return metric == M_COSINE ? k : ~k;
}
__device__ inline float key_to_score_rt(uint32_t key, int metric) {
    uint32_t k = metric == M_COSINE ? key : ~key;
    uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}
__device__ inline uint64_t make_key(uint32_t skey, uint32_t row) {
    return ((uint64_t)skey << 32) | (uint64_t)(~row);
}
__device__ inline uint32_t key_row(uint64_t key) { return ~(uint32_t)(key & 0xFFFFFFFFu); }
__device__ inline uint32_t key_skey(uint64_t key) { return (uint32_t)(key >> 32); }

// ---- radix-select helper: a 256-thread block finds the histogram bin in which the running
// count, taken from the TOP bin downwards, first reaches kp.  h has nbins (multiple of 256)
// counters, ctl is 8 words of LDS.  On return ctl[4] = that bin (0 if the total is < kp) and
// ctl[5] = the count of elements in that bin and all better ones.  Ends with a barrier.
__device__ inline void block_find_cut_bin(const uint32_t* h, int nbins, uint32_t kp, uint32_t* ctl) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bpt = nbins >> 8;
    const int top = nbins - 1 - tid * bpt;  // thread t owns bins top, top-1, ..., top-bpt+1
    uint32_t mine = 0;
    for (int b = 0; b < bpt; ++b) mine += h[top - b];
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) ctl[wave] = incl;
    if (tid == 0) { ctl[4] = 0u; ctl[5] = 0u; }
    __syncthreads();
    uint32_t before = incl - mine;
    for (int w = 0; w < wave; ++w) before += ctl[w];
    if (before < kp && before + mine >= kp) {
        uint32_t c = before;
        int b = 0;
        for (; b < bpt; ++b) { c += h[top - b]; if (c >= kp) break; }
        ctl[4] = (uint32_t)(top - b);
        ctl[5] = c;
    }
    if (tid == 255 && before + mine < kp) ctl[5] = before + mine;  // short: everything counts
    __syncthreads();
}

// "worst possible" fast score per metric (threshold meaning: nothing is filtered)
__host__ __device__ inline float worst_score(int metric) {
    return metric == M_COSINE ? -__builtin_huge_valf() : __builtin_huge_valf();
}

}  // namespace vrod
