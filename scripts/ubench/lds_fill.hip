// Microbenchmark (dev tool, not part of the library): LDS fill/read rates per CU on gfx950.
//   mode 0: global_load_lds_dwordx4 (LDS-DMA) from an L2-hot 64 KB window, 8 waves
//   mode 1: global_load_dwordx4 -> ds_write_b128 (register staging), 8 waves
//   mode 2: ds_read_b128 only, 8 waves
//   mode 3: waves 0-3 LDS-DMA, waves 4-7 ds_read_b128 concurrently
//   mode 4: waves 0-3 register staging, waves 4-7 ds_read_b128
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GLDS16(g, l) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g), (__attribute__((address_space(3))) void*)(l), 16, 0, 0)

template <int MODE>
__global__ __launch_bounds__(512) void k(const char* src, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* g = src + (size_t)blockIdx.x * 65536 + lane * 16;
    f32x4 acc = {0, 0, 0, 0};
    const bool writer = MODE == 0 || MODE == 1 || ((MODE == 3 || MODE == 4) && wave < 4);
    const bool reader = MODE == 2 || ((MODE == 3 || MODE == 4) && wave >= 4);
    for (int it = 0; it < iters; ++it) {
        if (writer) {
            // each wave moves 8 KB per iteration: 8 pieces of 1 KB
            if constexpr (MODE == 0 || MODE == 3) {
#pragma unroll
                for (int i = 0; i < 8; ++i) GLDS16(g + ((wave * 8 + i) & 63) * 1024, lds + (wave * 8 + i) * 1024);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                f32x4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4*>(g + ((wave * 8 + i) & 63) * 1024);
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(lds + (wave * 8 + i) * 1024 + lane * 16) = v[i];
            }
        }
        if (reader) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(lds + 65536 + ((wave * 8 + i) & 63) * 1024 + lane * 16);
                acc += v;
            }
        }
    }
    if (acc.x == 12345.f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE> static void run(const char* name, const char* d_src, float* d_out, int iters, double bytes_w, double bytes_r) {
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 512, 131072>>>(d_src, d_out, 10);
    hipEventRecord(e0);
    k<MODE><<<256, 512, 131072>>>(d_src, d_out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms  write %7.1f GB/s/CU  read %7.1f GB/s/CU\n", name, ms, bytes_w * iters / ms / 1e6, bytes_r * iters / ms / 1e6);
}
int main() {
    char* d_src; float* d_out;
    hipMalloc(&d_src, 256 * 65536); hipMemset(d_src, 1, 256 * 65536); hipMalloc(&d_out, 4096);
    const int iters = 20000;
    run<0>("0: LDS-DMA fill, 8 waves", d_src, d_out, iters, 65536, 0);
    run<1>("1: global_load + ds_write_b128, 8 waves", d_src, d_out, iters, 65536, 0);
    run<2>("2: ds_read_b128, 8 waves", d_src, d_out, iters, 0, 65536);
    run<3>("3: 4 waves LDS-DMA + 4 waves ds_read_b128", d_src, d_out, iters, 32768, 32768);
    run<4>("4: 4 waves reg-staged + 4 waves ds_read_b128", d_src, d_out, iters, 32768, 32768);
    return 0;
}
