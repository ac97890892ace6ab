// Microbenchmark: VALU issue rate on gfx950 for v_fma_f32 vs v_pk_fma_f32 vs v_min_f32/v_rcp_f32 (8 waves/SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
                x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
                p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                x0 = fminf(x0, a) + b; x1 = fmaxf(x1, a) * b; x2 = fminf(x2, b) + a; x3 = fmaxf(x3, b) * a;
                x4 = fminf(x4, a) + b; x5 = fmaxf(x5, a) * b; x6 = fminf(x6, b) + a; x7 = fmaxf(x7, b) * a;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                x0 = __builtin_amdgcn_rcpf(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_amdgcn_rcpf(x2); x3 = __builtin_amdgcn_rcpf(x3);
                x4 = __builtin_amdgcn_rcpf(x4); x5 = __builtin_amdgcn_rcpf(x5); x6 = __builtin_amdgcn_rcpf(x6); x7 = __builtin_amdgcn_rcpf(x7);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
void run(const char* name, int insts_per_iter, int flops_per_inst) {
    float* out;
    const int blocks = 256 * 8, iters = 2000;
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 10, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double waves = blocks * 4.0, winst = waves * iters * insts_per_iter;
    double per_simd = winst / (256.0 * 4.0);
    printf("%-12s %.3f ms  wave-instr/s %.3e  ns per wave-instr per SIMD %.3f  (cycles @2.4GHz %.2f)  TFLOP/s %.1f\n", name, ms,
           winst / (ms * 1e-3), ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, winst * 64 * flops_per_inst / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    run<0>("v_fma_f32", 64, 2);
    run<1>("v_pk_fma_f32", 64, 4);
    run<2>("minmax+addmul", 128, 1);
    run<3>("v_rcp_f32", 64, 1);
    return 0;
}
