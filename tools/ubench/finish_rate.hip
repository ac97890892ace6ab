// Microbenchmark: what one finishing pass (lean_finish) and one cull (fast_cull) cost a SIMD, with no memory traffic:
// every lane loops over its own pair K times (the result feeds the next iteration's input, so nothing is hoisted).
// Reports wall time per wave-pass per SIMD for 1..8 resident waves per SIMD and the in-kernel shader clock
// (delta s_memtime / delta s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../../sph_retina_amd/csrc/sph2pob_fast.hpp"
using namespace sph2pob;

template <int WHAT>
__global__ __launch_bounds__(256, 8) void k(const float* __restrict__ b1, const float* __restrict__ b2, float* out, int iters,
                                            unsigned long long* clk) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float x[5], y[5];
    for (int c = 0; c < 4; c++) { x[c] = b1[i * 4 + c]; y[c] = b2[i * 4 + c]; }
    x[4] = 0.0f; y[4] = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.0f;
    for (int it = 0; it < iters; it++) {
        float r;
        if (WHAT == 0) r = lean_finish<0, 4>(x, y, 0, 0);
        else r = fast_cull<4>(x, y, 0) ? 1.0f : 0.0f;
        acc += r;
        x[0] += r * 1e-3f + 1e-3f;   // the next pass depends on this one
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[i] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
    const int maxwg = 256 * 8, n = maxwg * 256;
    std::vector<float> h1(n * 4), h2(n * 4);
    srand(1);
    for (int i = 0; i < n; i++) {
        // overlapping pairs (the survivors of the cull): second box near the first
        float th = rand() / (float)RAND_MAX * 300 + 20, ph = rand() / (float)RAND_MAX * 140 + 20;
        float a = rand() / (float)RAND_MAX * 60 + 5, b = rand() / (float)RAND_MAX * 60 + 5;
        h1[i * 4] = th; h1[i * 4 + 1] = ph; h1[i * 4 + 2] = a; h1[i * 4 + 3] = b;
        h2[i * 4] = th + rand() / (float)RAND_MAX * 10 - 5; h2[i * 4 + 1] = ph + rand() / (float)RAND_MAX * 10 - 5;
        h2[i * 4 + 2] = a * (0.7f + 0.6f * rand() / (float)RAND_MAX); h2[i * 4 + 3] = b * (0.7f + 0.6f * rand() / (float)RAND_MAX);
    }
    float *d1, *d2, *out; unsigned long long* clk;
    (void)hipMalloc(&d1, n * 16); (void)hipMalloc(&d2, n * 16); (void)hipMalloc(&out, n * 4); (void)hipMalloc(&clk, 16);
    (void)hipMemcpy(d1, h1.data(), n * 16, hipMemcpyHostToDevice); (void)hipMemcpy(d2, h2.data(), n * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int what = 0; what < 2; what++)
        for (int wps : {1, 2, 4, 6, 8}) {
            const int wgs = 256 * wps, iters = what == 0 ? 400 : 2000;
            for (int rep = 0; rep < 3; rep++) {
                if (what == 0) k<0><<<wgs, 256>>>(d1, d2, out, iters, clk); else k<1><<<wgs, 256>>>(d1, d2, out, iters, clk);
            }
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            for (int rep = 0; rep < 5; rep++) {
                if (what == 0) k<0><<<wgs, 256>>>(d1, d2, out, iters, clk); else k<1><<<wgs, 256>>>(d1, d2, out, iters, clk);
            }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            unsigned long long c[2]; (void)hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
            const double ghz = (double)c[0] / (double)c[1] * 0.1;
            const double ns_per_pass_per_simd = ms * 1e6 / iters / wps;   // each SIMD runs wps waves x iters passes
            printf("%s waves/SIMD %d: %.3f ms, %.1f ns per wave-pass per SIMD = %.0f cycles at the in-kernel clock %.3f GHz (lone-wave latency view: %.1f ns per pass)\n",
                   what == 0 ? "lean_finish" : "fast_cull  ", wps, ms, ns_per_pass_per_simd, ns_per_pass_per_simd * ghz, ghz, ms * 1e6 / iters);
        }
    return 0;
}
