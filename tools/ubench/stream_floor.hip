// Microbenchmark: streaming floor for the aligned-IoU traffic pattern (2 x float4 in, 1 float out per pair).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k_simple(const float4* a, const float4* b, float* o, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { float4 x = a[i], y = b[i]; o[i] = x.x + x.y + x.z + x.w + y.x * y.y + y.z * y.w; }
}
__global__ __launch_bounds__(256) void k_loop(const float4* a, const float4* b, float* o, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float4 x = a[i], y = b[i]; o[i] = x.x + x.y + x.z + x.w + y.x * y.y + y.z * y.w;
    }
}
template <int WORK>
__global__ __launch_bounds__(256) void k_work(const float4* a, const float4* b, float* o, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        float4 x = a[i], y = b[i];
        float acc = x.x;
#pragma unroll 8
        for (int k = 0; k < WORK; k++) acc = fmaf(acc, y.x, x.y);   // dependent chain of WORK fmas
        o[i] = acc;
    }
}
int main() {
    for (int n : {1000000, 16000000}) {
        float4 *a, *b; float* o;
        (void)hipMalloc(&a, (size_t)n * 16); (void)hipMalloc(&b, (size_t)n * 16); (void)hipMalloc(&o, (size_t)n * 4);
        (void)hipMemset(a, 0, (size_t)n * 16); (void)hipMemset(b, 0, (size_t)n * 16);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        auto timeit = [&](const char* name, auto launch) {
            for (int w = 0; w < 20; w++) launch();
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            for (int r = 0; r < 200; r++) launch();
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("n=%d %-22s %.2f us  %.0f GB/s\n", n, name, ms * 1000 / 200, 36.0 * n / (ms / 200 * 1e-3) / 1e9);
        };
        int blocks = (n + 255) / 256;
        timeit("simple", [&] { k_simple<<<blocks, 256>>>(a, b, o, n); });
        timeit("loop 1024 WGs", [&] { k_loop<<<1024, 256>>>(a, b, o, n); });
        timeit("loop 2048 WGs", [&] { k_loop<<<2048, 256>>>(a, b, o, n); });
        timeit("work 100 dep fma", [&] { k_work<100><<<blocks, 256>>>(a, b, o, n); });
        timeit("work 300 dep fma", [&] { k_work<300><<<blocks, 256>>>(a, b, o, n); });
        timeit("work 500 dep fma", [&] { k_work<500><<<blocks, 256>>>(a, b, o, n); });
        (void)hipFree(a); (void)hipFree(b); (void)hipFree(o);
    }
    return 0;
}
