// Microbenchmark: what a stream-ordered launch of the dominant kernel's SHAPE costs before any arithmetic — the period of
// back-to-back launches of (a) an empty kernel, (b) the same with the chunk kernel's 18 KB of LDS per workgroup, (c) one
// that only loads its two 16-byte boxes per pair and stores 4 bytes (the memory side alone), on 1 954 workgroups of 256
// threads (1 M pairs).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k_empty(float* out, int n) {
    if (n < 0) out[0] = 1.0f;
}
__global__ __launch_bounds__(256) void k_lds(float* out, int n) {
    __shared__ float q[4608];
    if (n < 0) { q[threadIdx.x] = 1.0f; out[0] = q[(threadIdx.x * 7) & 255]; }
}
__global__ __launch_bounds__(256, 8) void k_stream(const float4* __restrict__ a, const float4* __restrict__ b, float* __restrict__ out, int n) {
    const int wave = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 128, lane = threadIdx.x & 63;
    if (wave >= n) return;
    const int i0 = min(wave + lane, n - 1), i1 = min(wave + 64 + lane, n - 1);
    const float4 x0 = a[i0], y0 = b[i0], x1 = a[i1], y1 = b[i1];
    if (wave + lane < n) out[wave + lane] = x0.x + y0.y;
    if (wave + 64 + lane < n) out[wave + 64 + lane] = x1.z + y1.w;
}
int main() {
    const int n = 1000000, wgs = (n + 511) / 512, reps = 5000;
    float4 *a, *b; float* out;
    (void)hipMalloc(&a, n * 16); (void)hipMalloc(&b, n * 16); (void)hipMalloc(&out, n * 4);
    (void)hipMemset(a, 0, n * 16); (void)hipMemset(b, 0, n * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int what = 0; what < 3; what++) {
        auto launch = [&]() {
            if (what == 0) k_empty<<<wgs, 256>>>(out, n);
            else if (what == 1) k_lds<<<wgs, 256>>>(out, n);
            else k_stream<<<wgs, 256>>>(a, b, out, n);
        };
        for (int r = 0; r < 3000; r++) launch();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < reps; r++) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %.3f us per launch (back to back, one stream)\n", what == 0 ? "empty kernel" : what == 1 ? "empty kernel + 18 KB LDS" : "load 32 B + store 4 B per pair", ms * 1e3 / reps);
    }
    // the same launches replayed from a hipGraph (100 kernel nodes captured from the stream, the graph launched 50 times)
    {
        hipStream_t st; (void)hipStreamCreate(&st);
        for (int what = 0; what < 3; what += 2) {
            hipGraph_t graph; hipGraphExec_t exec;
            (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
            for (int r = 0; r < 100; r++) {
                if (what == 0) k_empty<<<wgs, 256, 0, st>>>(out, n); else k_stream<<<wgs, 256, 0, st>>>(a, b, out, n);
            }
            (void)hipStreamEndCapture(st, &graph);
            (void)hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            for (int r = 0; r < 20; r++) (void)hipGraphLaunch(exec, st);
            (void)hipStreamSynchronize(st);
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < 50; r++) (void)hipGraphLaunch(exec, st);
            (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("hipGraph of 100 nodes, %-30s %.3f us per node\n", what == 0 ? "empty kernel:" : "load 32 B + store 4 B per pair:", ms * 1e3 / 5000);
            (void)hipGraphExecDestroy(exec); (void)hipGraphDestroy(graph);
        }
    }
    // how the empty launch depends on the grid
    for (int g : {1, 256, 512, 1024, 1954, 3908, 7816}) {
        for (int r = 0; r < 2000; r++) k_empty<<<g, 256>>>(out, n);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < reps; r++) k_empty<<<g, 256>>>(out, n);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("empty kernel, %5d workgroups of 256 threads: %.3f us per launch\n", g, ms * 1e3 / reps);
    }
    return 0;
}
