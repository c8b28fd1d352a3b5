// mfma_common.h -- shared by the MFMA scan kernels (kernels_mfma_phased.hip, kernels_mfma_w4.hip,
// kernels_mfma_skinny.hip) and their dispatcher (kernels_mfma.hip): tile constants, the kernel
// argument block, LDS-DMA / raw-barrier macros and the work-group -> (strip, query block) map.
#pragma once
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
// 4-wave kernel: behind the two stages the work-group's 256 thresholds / query norms and two 256-row slots of row
// norms (the tile epilogue issues no global load: a compiler-counted load there would drain the LDS-DMA pieces in
// flight), then the four wave-private hit logs up to the end of the CU's 160 KB (kernels_mfma_w4.hip, "The hit dump")
constexpr int kLdsThr = 2 * kStageBytes;          // [256] f32
constexpr int kLdsQn2 = kLdsThr + 1024;           // [256] f32
constexpr int kLdsXn2 = kLdsQn2 + 1024;           // [2][256] f32, slot = tile parity
constexpr int kLdsDump = kLdsXn2 + 2048;          // 4 logs
constexpr int kLdsTotalW4 = 160 * 1024;
// ... and per wave of the grid a region of global memory that full logs are spilled to (entries of 128 B, then one
// descriptor word per entry); read back by the wave itself at the end of the launch
constexpr uint32_t kDumpRegionCap = 512;
constexpr uint32_t kDumpRegionBytes = kDumpRegionCap * (128 + 4);

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
    char* dump;             // 4-wave kernel, filtered launches: spill regions of the hit logs, mfma_dump_bytes() (scratch)
    uint32_t* claims;       // 4-wave kernel, filtered launches: [nqb][nstrips] claim bits of the strips' tail chunks (work stealing), zeroed per launch; null: static shares only
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

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// one launcher per translation unit (kernels_mfma.hip picks among them)
void launch_mfma_phased_f32(const MfmaKernelArgs& a, int metric, bool dense, int grid, hipStream_t s, hipEvent_t start, hipEvent_t stop);
void launch_mfma_w4(const MfmaKernelArgs& a, int metric, int dense_form, bool split, int grid, hipStream_t s, hipEvent_t start, hipEvent_t stop);
void launch_mfma_skinny(const MfmaKernelArgs& a, int metric, bool dense, bool split, uint32_t nq, int num_cus, hipStream_t s, hipEvent_t start, hipEvent_t stop);

}  // namespace vrod
