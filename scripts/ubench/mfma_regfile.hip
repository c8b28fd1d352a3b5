// Micro-benchmark: does v_mfma_f32_16x16x32_bf16 run at the same rate with its A/B operands in AGPRs and / or its
// accumulator in arch VGPRs?  (kernels_mfma_w4.hip keeps the fragments in a[0:127] and 48 of the 64 accumulator tiles in
// v[64:255] so that the tile epilogue's v_max3 reads them without v_accvgpr_read.)  One wave per SIMD, 256 CUs.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_regfile mfma_regfile.hip && ./mfma_regfile
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int V>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters) {
    asm volatile("" ::: "a0","a15","a31","a63","a127","a191","a255","v64","v127","v191","v255");
    __shared__ __attribute__((aligned(16))) char lds_buf[16384];
    const unsigned lds_off = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_buf + (threadIdx.x & 63) * 16u + (threadIdx.x >> 6) * 4096u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (V == 0) { REP16(asm volatile("v_mfma_f32_16x16x32_bf16 a[128:131], v[8:11], v[12:15], a[128:131]\n\tv_mfma_f32_16x16x32_bf16 a[132:135], v[8:11], v[16:19], a[132:135]\n\tv_mfma_f32_16x16x32_bf16 a[136:139], v[8:11], v[20:23], a[136:139]\n\tv_mfma_f32_16x16x32_bf16 a[140:143], v[8:11], v[24:27], a[140:143]" ::: "memory");) }
        if constexpr (V == 1) { REP16(asm volatile("v_mfma_f32_16x16x32_bf16 a[128:131], a[8:11], a[12:15], a[128:131]\n\tv_mfma_f32_16x16x32_bf16 a[132:135], a[8:11], a[16:19], a[132:135]\n\tv_mfma_f32_16x16x32_bf16 a[136:139], a[8:11], a[20:23], a[136:139]\n\tv_mfma_f32_16x16x32_bf16 a[140:143], a[8:11], a[24:27], a[140:143]" ::: "memory");) }
        if constexpr (V == 2) { REP16(asm volatile("v_mfma_f32_16x16x32_bf16 v[128:131], a[8:11], a[12:15], v[128:131]\n\tv_mfma_f32_16x16x32_bf16 v[132:135], a[8:11], a[16:19], v[132:135]\n\tv_mfma_f32_16x16x32_bf16 v[136:139], a[8:11], a[20:23], v[136:139]\n\tv_mfma_f32_16x16x32_bf16 v[140:143], a[8:11], a[24:27], v[140:143]" ::: "memory");) }
        if constexpr (V == 3) { REP16(asm volatile("v_mfma_f32_16x16x32_bf16 v[128:131], v[8:11], v[12:15], v[128:131]\n\tv_mfma_f32_16x16x32_bf16 v[132:135], v[8:11], v[16:19], v[132:135]\n\tv_mfma_f32_16x16x32_bf16 v[136:139], v[8:11], v[20:23], v[136:139]\n\tv_mfma_f32_16x16x32_bf16 v[140:143], v[8:11], v[24:27], v[140:143]" ::: "memory");) }
        // the scan loop's mix: one ds_read_b128 of a fragment per group of four MFMAs (32 per 128), destination in the file the
        // A / B operands live in, never a register the MFMAs of this iteration read
        if constexpr (V == 4) { REP16(asm volatile("v_mfma_f32_16x16x32_bf16 a[128:131], v[8:11], v[12:15], a[128:131]\n\tds_read_b128 v[32:35], %0\n\tv_mfma_f32_16x16x32_bf16 a[132:135], v[8:11], v[16:19], a[132:135]\n\tv_mfma_f32_16x16x32_bf16 a[136:139], v[8:11], v[20:23], a[136:139]\n\tv_mfma_f32_16x16x32_bf16 a[140:143], v[8:11], v[24:27], a[140:143]" :: "v"(lds_off) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if constexpr (V == 5) { REP16(asm volatile("v_mfma_f32_16x16x32_bf16 v[128:131], a[8:11], a[12:15], v[128:131]\n\tds_read_b128 a[32:35], %0\n\tv_mfma_f32_16x16x32_bf16 v[132:135], a[8:11], a[16:19], v[132:135]\n\tv_mfma_f32_16x16x32_bf16 v[136:139], a[8:11], a[20:23], v[136:139]\n\tv_mfma_f32_16x16x32_bf16 v[140:143], a[8:11], a[24:27], v[140:143]" :: "v"(lds_off) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if constexpr (V == 6) { REP16(asm volatile("v_mfma_f32_16x16x32_bf16 v[128:131], v[8:11], v[12:15], v[128:131]\n\tds_read_b128 v[32:35], %0\n\tv_mfma_f32_16x16x32_bf16 v[132:135], v[8:11], v[16:19], v[132:135]\n\tv_mfma_f32_16x16x32_bf16 v[136:139], v[8:11], v[20:23], v[136:139]\n\tv_mfma_f32_16x16x32_bf16 v[140:143], v[8:11], v[24:27], v[140:143]" :: "v"(lds_off) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[V] = t1 - t0;
}
int main() {
    unsigned long long* d; (void)hipMalloc(&d, 64); (void)hipMemset(d, 0, 64);
    const int iters = 2000;
    for (int r = 0; r < 2; ++r) { k<0><<<256, 256>>>(d, iters); k<1><<<256, 256>>>(d, iters); k<2><<<256, 256>>>(d, iters); k<3><<<256, 256>>>(d, iters);
                                  k<4><<<256, 256>>>(d, iters); k<5><<<256, 256>>>(d, iters); k<6><<<256, 256>>>(d, iters); }
    (void)hipDeviceSynchronize();
    unsigned long long h[7]; (void)hipMemcpy(h, d, 56, hipMemcpyDeviceToHost);
    const char* n[7] = {"A,B in VGPR, C/D in AGPR (today)", "A,B in AGPR, C/D in AGPR", "A,B in AGPR, C/D in VGPR", "A,B in VGPR, C/D in VGPR",
                        "today + ds_read_b128 -> VGPR per 4 MFMAs", "A,B AGPR, C/D VGPR + ds_read_b128 -> AGPR", "all VGPR + ds_read_b128 -> VGPR"};
    for (int v = 0; v < 7; ++v) printf("%-36s %.2f cycles per MFMA\n", n[v], (double)h[v] / (iters * 64.0));
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
