// Micro-benchmark: what the instructions between the MFMAs of the 4-wave scan loop cost.  One wave per SIMD (256 threads per
// work-group, 256 work-groups), the loop's own mix per "phase" of 32 MFMAs (16 accumulator tiles, each hit twice, 16 apart):
//   V0  MFMAs only
//   V1  + 8 ds_read_b128 (two behind each of the first four MFMA groups, as the kernel places them) + s_waitcnt lgkmcnt(0)
//   V2  + 4 LDS-DMA pieces per phase (s_add_u32 m0 one group ahead of its global_load_lds_dwordx4), counted vmcnt(16) +
//         s_barrier every second phase
//   V3  V2 without the barrier            V4  V2 with the pieces' source in a 64-KB region per work-group (all L2 hits)
//   V6  V2 (rows streamed from HBM) with contiguous pieces: the corpus stored tile-major in LDS-image order
//   V5  V4 with every piece 1 KB of CONTIGUOUS memory (an operand stored in LDS-image order) instead of 8 rows x 128 B
// prints shader cycles per MFMA and the in-kernel clock.
//   hipcc --offload-arch=gfx950 -O3 -o loop_mix loop_mix.hip && ./loop_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define MF(D, A, B) "v_mfma_f32_16x16x32_bf16 a[" #D ":" #D "+3], v[" #A ":" #A "+3], v[" #B ":" #B "+3], a[" #D ":" #D "+3]\n\t"
#define G0 MF(0, 64, 80) MF(4, 64, 84) MF(8, 64, 88) MF(12, 64, 92)
#define G1 MF(16, 68, 80) MF(20, 68, 84) MF(24, 68, 88) MF(28, 68, 92)
#define G2 MF(32, 72, 80) MF(36, 72, 84) MF(40, 72, 88) MF(44, 72, 92)
#define G3 MF(48, 76, 80) MF(52, 76, 84) MF(56, 76, 88) MF(60, 76, 92)
#define VCLOB "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"
#define RD(R, O) "ds_read_b128 v[" #R ":" #R "+3], %0 offset:" #O "\n\t"
#define M0(I) "s_add_u32 m0, %1, " #I "\n\t"
#define DMA(V) "global_load_lds_dwordx4 %" #V ", %6\n\t"
__global__ void fill(uint32_t* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (h & 0x807F807Fu) | 0x3D803D80u;
    }
}
template <int V>
__global__ __launch_bounds__(256) void k(unsigned long long* out, const char* src, int iters, size_t wg_stride) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    asm volatile("" ::: "a0", "a63", VCLOB);
    {   // MFMA operands with real bit patterns (all-zero operands toggle nothing: the chip then holds 2.39 GHz and the
        // numbers mean nothing): bf16 pairs of random sign and mantissa around 2^-4, different per lane and register
        uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#define FILL(R) h = h * 1664525u + 1013904223u; asm volatile("v_mov_b32 v" #R ", %0" :: "v"((h & 0x807F807Fu) | 0x3D803D80u) : "v" #R);
        FILL(64) FILL(65) FILL(66) FILL(67) FILL(68) FILL(69) FILL(70) FILL(71) FILL(72) FILL(73) FILL(74) FILL(75) FILL(76) FILL(77) FILL(78) FILL(79)
        FILL(80) FILL(81) FILL(82) FILL(83) FILL(84) FILL(85) FILL(86) FILL(87) FILL(88) FILL(89) FILL(90) FILL(91) FILL(92) FILL(93) FILL(94) FILL(95)
#undef FILL
    }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t lds_rd = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + lane * 16u + wave * 8192u;
    const uint32_t lds_w = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + 65536u + wave * 16384u);
    const uint32_t v0 = (V == 5 || V == 6) ? lane * 16u : (lane >> 3) * 1536u + ((lane & 7) << 4);
    const uint32_t v1 = v0 + ((V == 5 || V == 6) ? 1024u : 8 * 1536u), v2 = v0 + ((V == 5 || V == 6) ? 2048u : 16 * 1536u), v3 = v0 + ((V == 5 || V == 6) ? 3072u : 24 * 1536u);
    const char* base = src + (size_t)blockIdx.x * wg_stride + wave * (V == 6 ? 4096u : 65536u);
    const uint64_t bv = (uint64_t)base;
    auto uni = [](uint64_t v) {   // (the builtin returns int: cast before widening)
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return (const char*)(((uint64_t)hi << 32) | lo);
    };
    const char* sb = uni(bv);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (V == 0) {
            asm volatile(G0 G1 G2 G3 G0 G1 G2 G3 ::: "memory", VCLOB);
            asm volatile(G0 G1 G2 G3 G0 G1 G2 G3 ::: "memory", VCLOB);
        } else if constexpr (V == 1) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\t" G0 RD(96, 0) RD(100, 1024) G1 RD(104, 2048) RD(108, 3072) G2 RD(112, 4096) RD(116, 5120) G3 RD(120, 6144) RD(124, 7168) G0 G1 G2 G3 :: "v"(lds_rd) : "memory", VCLOB);
            asm volatile("s_waitcnt lgkmcnt(0)\n\t" G0 RD(96, 0) RD(100, 1024) G1 RD(104, 2048) RD(108, 3072) G2 RD(112, 4096) RD(116, 5120) G3 RD(120, 6144) RD(124, 7168) G0 G1 G2 G3 :: "v"(lds_rd) : "memory", VCLOB);
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)\n\t" M0(0) G0 RD(96, 0) RD(100, 1024) DMA(2) G1 RD(104, 2048) RD(108, 3072) M0(1024) G2 RD(112, 4096) RD(116, 5120) DMA(3) G3 RD(120, 6144) RD(124, 7168) M0(2048)
                         G0 DMA(4) G1 M0(3072) G2 DMA(5) G3
                         :: "v"(lds_rd), "s"(lds_w), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sb) : "memory", "scc", VCLOB);
            asm volatile("s_waitcnt lgkmcnt(0)\n\t" M0(4096) G0 RD(96, 0) RD(100, 1024) DMA(2) G1 RD(104, 2048) RD(108, 3072) M0(5120) G2 RD(112, 4096) RD(116, 5120) DMA(3) G3 RD(120, 6144) RD(124, 7168) M0(6144)
                         G0 DMA(4) G1 M0(7168) G2 DMA(5) G3
                         :: "v"(lds_rd), "s"(lds_w), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sb) : "memory", "scc", VCLOB);
            if constexpr (V == 2 || V == 4 || V == 5 || V == 6) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            if constexpr (V == 6) sb += 16384;   // the work-group's four waves read 4 x 4 KB of fresh, contiguous memory per iteration (64 MB per work-group)
            else if constexpr (V != 4 && V != 5) { sb += 128; if ((i & 7) == 7) sb += 256 * 1536 - 8 * 128; }   // walk a row-major corpus K-tile by K-tile
            if ((i % 4000) == 3999) sb = base;   // (the walk wraps: every variant runs long enough for the clock to settle)
            sb = uni((uint64_t)sb);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[V * 2] = t1 - t0; out[V * 2 + 1] = r1 - r0; }
}
int main() {
    unsigned long long* d; (void)hipMalloc(&d, 256); (void)hipMemset(d, 0, 256);
    const int iters = 60000;   // x 64 MFMAs: ~35 ms per launch; the walk of the HBM variants wraps every 4000
    const size_t wg_stride = (size_t)(4000 / 8 + 2) * 256 * 1536 + (1 << 20);   // every work-group walks its own rows (HBM), V4: stays in 256 KB
    char* src; if (hipMalloc(&src, wg_stride * 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
    fill<<<4096, 256>>>((uint32_t*)src, wg_stride * 256 / 4);   // random bytes: what moves through L2 -> LDS toggles like data
    const int lds_bytes = 65536 + 4 * 16384;
#define RUN(V) (void)hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); k<V><<<256, 256, lds_bytes>>>(d, src, iters, wg_stride);
    // each variant three times in a row (the clock settles over tens of ms), the last one counts; two rounds, the second reported
    for (int r = 0; r < 2; ++r) { RUN(0) RUN(0) RUN(0) RUN(1) RUN(1) RUN(1) RUN(2) RUN(2) RUN(2) RUN(6) RUN(6) RUN(6) RUN(3) RUN(3) RUN(3) RUN(4) RUN(4) RUN(4) RUN(5) RUN(5) RUN(5) }
    (void)hipDeviceSynchronize();
    unsigned long long h[14]; (void)hipMemcpy(h, d, 112, hipMemcpyDeviceToHost);
    const char* n[7] = {"MFMAs only", "+ 8 ds_read_b128 per 32 MFMAs", "+ 4 LDS-DMA pieces per 32 MFMAs, vmcnt(16) + barrier per 64", "  ... without the barrier", "  ... pieces from a 256-KB region (L2 hits)", "  ... L2 hits, every piece 1 KB contiguous", "  ... from HBM, every piece 1 KB contiguous (with barrier)"};
    for (int v = 0; v < 7; ++v) printf("%-62s %.2f cycles per MFMA  %.1f per 128  clock %.3f GHz\n", n[v], (double)h[v * 2] / (iters * 64.0), (double)h[v * 2] / (iters * 64.0) * 128, (double)h[v * 2] / (double)h[v * 2 + 1] * 0.1);
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
