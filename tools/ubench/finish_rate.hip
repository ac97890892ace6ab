// Microbenchmark: what one finishing pass (lean_finish) and one cull (fast_cull) cost a SIMD, with no memory traffic:
// every lane loops over its own pair K times (the result feeds the next iteration's input, so nothing is hoisted).
// Reports wall time per wave-pass per SIMD for 1..8 resident waves per SIMD and the in-kernel shader clock
// (delta s_memtime / delta s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#if defined(COUNT_SITES)
__device__ unsigned long long g_site_hits[8];
#define SPH_SITE_HIT(k) do { if ((threadIdx.x & 63) == __builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) atomicAdd(&g_site_hits[k], 1ull); } while (0)
#endif
#if defined(NO_GUARDS)
#define SPH_ANY_LANE(cond) false   // the common path alone: what the wave-uniform guards of the rare branches cost
#endif
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
        // the next pass depends on this one, through every coordinate (nothing of the pass is loop-invariant)
        const float d = r * 1e-4f + 1e-4f;
        x[0] += d; x[1] -= 0.5f * d; x[2] += 0.25f * d; x[3] -= 0.125f * d;
        y[0] -= 0.3f * d; y[1] += 0.2f * d; y[2] -= 0.1f * d; y[3] += 0.05f * d;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[i] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
    const int maxwg = 256 * 8, n = maxwg * 256;
    std::vector<float> h1(n * 4), h2(n * 4);
    srand(1);
    // survivors of the cull on the benchmark distribution (bench.py make_boxes: theta 360 u, phi 180 u, extents 1 + 99 u):
    // what the finishing pass sees in the dominant launch, rare branches at their real frequencies
    {
        auto uni = []() { return rand() / (float)RAND_MAX; };
        int i = 0;
        while (i < n) {
            float g[5] = {360 * uni(), 180 * uni(), 1 + 99 * uni(), 1 + 99 * uni(), 0.0f};
            float q[5] = {360 * uni(), 180 * uni(), 1 + 99 * uni(), 1 + 99 * uni(), 0.0f};
            if (fast_cull<4>(g, q, 0)) continue;
            for (int c = 0; c < 4; c++) { h1[i * 4 + c] = g[c]; h2[i * 4 + c] = q[c]; }
            i++;
        }
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
#if defined(COUNT_SITES)
    {
        unsigned long long z[8] = {0}, h[8];
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_site_hits), z, sizeof(z));
        const int wgs = 256 * 8, iters = 400;
        k<0><<<wgs, 256>>>(d1, d2, out, iters, clk);
        (void)hipDeviceSynchronize();
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_site_hits), sizeof(h));
        const double passes = (double)wgs * 4 * iters;
        printf("share of the passes that enter a guarded block: spherical shift %.4f  floors %.4f  rotated jitter %.4f  near-parallel %.4f\n",
               h[0] / passes, h[1] / passes, h[2] / passes, h[3] / passes);
    }
#endif
    return 0;
}
