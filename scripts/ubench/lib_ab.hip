// Dev tool: the batched scan launch (vrod::launch_scan_mfma) of SEVERAL builds of the library, interleaved in ONE
// process on the same data (cdna_hip_programming.md rule 24: perf deltas come from interleaved rounds in one process).
// The builds are dlopen'ed by path; nothing but the launcher and its argument block is used, so the kernel is timed
// without the search flow around it.  Thresholds are one value for every query: +inf = no hit anywhere (the kernel's
// own rate), a finite value = appends at a chosen density (printed per launch).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../vrod_amd/csrc -o lib_ab lib_ab.hip -ldl
//   ./lib_ab rows reps rounds thr lib1.so [lib2.so ...]
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vrod_common.h"
#include "vrod_kernels.h"

__global__ void fill_kernel(uint16_t* p, uint64_t n, uint64_t seed) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t z = (i + seed * 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        // the library's synthetic stream in spirit: sum of four 16-bit fields ~ Gaussian, rows of norm ~1 at d = 768
        const float v = ((float)(z & 0xFFFF) + (float)((z >> 16) & 0xFFFF) + (float)((z >> 32) & 0xFFFF) + (float)(z >> 48) - 131070.f) * (1.0f / (37837.f * 27.7f));
        uint32_t u = __float_as_uint(v); u += 0x7FFFu + ((u >> 16) & 1u);
        p[i] = (uint16_t)(u >> 16);
    }
}
__global__ void fill_f32(float* p, uint32_t n, float v) { for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v; }

typedef void (*launch_fn)(const vrod::MfmaScanArgs&, int, int, hipStream_t);
typedef int (*clk_fn)(unsigned long long*, int);   // -DVROD_W4_CLK builds: in-kernel clock stamps

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: lib_ab rows reps rounds thr lib1.so [lib2.so ...]\n"); return 2; }
    const uint32_t rows = atoi(argv[1]), dim = 768, nq = 1024;
    const int reps = atoi(argv[2]), rounds = atoi(argv[3]);
    const float thr = atof(argv[4]);
    std::vector<launch_fn> fn;
    std::vector<clk_fn> clk, xcd;
    std::vector<const char*> names;
    for (int i = 5; i < argc; ++i) {
        void* h = dlopen(argv[i], RTLD_NOW | RTLD_LOCAL);
        if (!h) { fprintf(stderr, "dlopen %s: %s\n", argv[i], dlerror()); return 1; }
        void* f = dlsym(h, "_ZN4vrod16launch_scan_mfmaERKNS_12MfmaScanArgsEiiP12ihipStream_t");
        if (!f) { fprintf(stderr, "%s: no vrod::launch_scan_mfma\n", argv[i]); return 1; }
        fn.push_back((launch_fn)f); names.push_back(argv[i]);
        clk.push_back((clk_fn)dlsym(h, "vrod_debug_w4_clk"));
        xcd.push_back((clk_fn)dlsym(h, "vrod_debug_w4_xcd"));
    }
    uint16_t *d_c, *d_q; float *d_thr, *d_xn, *d_qn; uint2* d_lists; uint32_t *d_counts, *d_pace;
    (void)hipMalloc(&d_c, (size_t)rows * dim * 2); (void)hipMalloc(&d_q, (size_t)nq * dim * 2);
    (void)hipMalloc(&d_thr, nq * 4); (void)hipMalloc(&d_xn, (size_t)rows * 4); (void)hipMalloc(&d_qn, nq * 4);
    (void)hipMalloc(&d_lists, (size_t)nq * 8192 * 8); (void)hipMalloc(&d_counts, nq * 4); (void)hipMalloc(&d_pace, 8192);
    fill_kernel<<<4096, 256>>>(d_c, (uint64_t)rows * dim, 101);
    fill_kernel<<<256, 256>>>(d_q, (uint64_t)nq * dim, 102);
    fill_f32<<<64, 256>>>(d_thr, nq, thr);
    (void)hipMemset(d_counts, 0, nq * 4); (void)hipMemset(d_pace, 0, 8192); (void)hipMemset(d_xn, 0, (size_t)rows * 4); (void)hipMemset(d_qn, 0, nq * 4);
    vrod::MfmaScanArgs a{};
    a.corpus = d_c; a.queries = d_q; a.xnorm2 = d_xn; a.qnorm2 = d_qn; a.thr = d_thr; a.lists = d_lists; a.counts = d_counts; a.cap = 8192;
    a.ld = dim; a.nq_pad = nq; a.nq = nq; a.row_begin = 0; a.row_end = rows; a.metric = vrod::M_COSINE; a.pace = d_pace; a.pace_is_zero = false;
    uint32_t* d_claims; (void)hipMalloc(&d_claims, 4096); a.claims = d_claims;
    void* d_dump; (void)hipMalloc(&d_dump, (size_t)256 * 4 * 512 * 132); a.dump = d_dump;   // (builds without the field ignore it: it is the struct's last)
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::vector<std::vector<float>> ms(fn.size());
    std::vector<double> appends(fn.size(), 0.0);
    for (size_t v = 0; v < fn.size(); ++v) { fn[v](a, vrod::DT_BF16, 256, 0); }   // warm: code objects, attributes
    (void)hipDeviceSynchronize();
    unsigned long long ck[12];
    for (size_t v = 0; v < fn.size(); ++v) if (clk[v]) clk[v](ck, 1);
    unsigned long long xk[32];
    for (size_t v = 0; v < fn.size(); ++v) if (xcd[v]) xcd[v](xk, 1);
    for (int r = 0; r < rounds; ++r)
        for (size_t v = 0; v < fn.size(); ++v) {
            (void)hipMemsetAsync(d_counts, 0, nq * 4, 0);
            (void)hipEventRecord(e0);
            for (int i = 0; i < reps; ++i) fn[v](a, vrod::DT_BF16, 256, 0);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float m; (void)hipEventElapsedTime(&m, e0, e1);
            ms[v].push_back(m / reps);
            std::vector<uint32_t> cnt(nq); (void)hipMemcpy(cnt.data(), d_counts, nq * 4, hipMemcpyDeviceToHost);
            uint64_t tot = 0; for (auto c : cnt) tot += c;
            appends[v] = (double)tot / reps;
        }
    for (size_t v = 0; v < fn.size(); ++v) {
        std::vector<float> s = ms[v]; std::sort(s.begin(), s.end());
        const float med = s[s.size() / 2], mn = s[0];
        printf("rows %u thr %g  %-40s median %.4f ms (%.1f TF)  min %.4f ms (%.1f TF)  appends/launch %.0f  (%s)\n", rows, thr, names[v], med,
               2.0 * rows * nq * dim / (med * 1e-3) / 1e12, mn, 2.0 * rows * nq * dim / (mn * 1e-3) / 1e12, appends[v], hipGetErrorString(hipGetLastError()));
        if (clk[v] && clk[v](ck, 1) == 0 && ck[1])
            printf("    in-kernel clock %.3f GHz, %.1f shader cycles per K-tile, %.1f us per work-group; tile epilogue %.0f cycles + %.0f at the barrier behind it (wave 0)\n",
                   (double)ck[0] / (double)ck[1] * 0.1, (double)ck[0] / (double)ck[3], (double)ck[1] / (double)ck[2] * 0.01,
                   ck[6] ? (double)ck[4] / (double)ck[6] : 0.0, ck[6] ? (double)ck[5] / (double)ck[6] : 0.0);
        if (xcd[v] && xcd[v](xk, 1) == 0)
            for (int x = 0; x < 8; ++x) if (xk[x * 4 + 3]) printf("    XCC %d: %llu work-groups, loop avg %.1f us, min %.1f, max %.1f\n", x, xk[x * 4 + 3],
                (double)xk[x * 4] / (double)xk[x * 4 + 3] * 0.01, (double)(~xk[x * 4 + 2]) * 0.01, (double)xk[x * 4 + 1] * 0.01);
    }
    return 0;
}
