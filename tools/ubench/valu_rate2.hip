// Microbenchmark 2: true per-instruction VALU issue cost on gfx950 (inline asm prevents SLP packing).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));
#define REP8(X) X X X X X X X X
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    float2v p0 = {x0, x1}, p1 = {x2, x3}, pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));) }
        if (MODE == 1) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3" : "+v"(p0), "+v"(p1) : "v"(pa), "v"(pb));) }
        if (MODE == 2) { REP8(asm volatile("v_min_f32 %0, %0, %4\n v_max_f32 %1, %1, %5\n v_min_f32 %2, %2, %4\n v_max_f32 %3, %3, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));) }
        if (MODE == 3) { REP8(asm volatile("v_mul_f32 %0, %0, %4\n v_add_f32 %1, %1, %5\n v_mul_f32 %2, %2, %4\n v_add_f32 %3, %3, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));) }
        if (MODE == 4) { REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));) }
        if (MODE == 5) { REP8(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %5, vcc" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "vcc");) }
        if (MODE == 6) { REP8(asm volatile("v_sqrt_f32 %0, %0\n v_rsq_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_rsq_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + p0.x + p0.y + p1.x + p1.y;
}
template <int MODE>
void run(const char* name, int blocks_per_cu) {
    float* out;
    const int blocks = 256 * blocks_per_cu, iters = 4000, insts_per_iter = 32;
    (void)hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 10, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double winst = blocks * 4.0 * iters * insts_per_iter;
    double per_simd = winst / 1024.0;
    printf("%-14s waves/SIMD %d  %.3f ms  ns per wave-instr per SIMD %.3f  (cycles @2.4GHz %.2f)\n", name, blocks_per_cu, ms,
           ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    (void)hipFree(out);
}
int main() {
    for (int b : {1, 2, 8}) {
        run<0>("v_fma_f32", b); run<1>("v_pk_fma_f32", b); run<2>("v_min/max_f32", b); run<3>("v_mul/add_f32", b);
        run<4>("v_rcp_f32", b); run<5>("v_cndmask_b32", b); run<6>("v_sqrt/rsq", b);
    }
    return 0;
}
