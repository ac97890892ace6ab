// Which SIMD does wave w of a 4-wave workgroup land on?  (HW_REG_HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8]
// sh_id[12] se_id[15:13]; XCC_ID register 20 [3:0])  Grid shaped like the dominant kernel's launch: 1 536 workgroups of
// 256 threads with 18 KB of LDS each.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned* out, int spin) {
    __shared__ float pad[4608];
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float x = pad[(threadIdx.x * 7) & 255];
    for (int i = 0; i < spin; i++) x = x * 1.0001f + 0.5f;
    unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID, offset 0, size 32
    unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    if (x == 12345.678f) out[0] = 0;
}
int main() {
    const int wgs = 1536;
    unsigned* d;
    (void)hipMalloc(&d, wgs * 4 * 2 * 4);
    k<<<wgs, 256>>>(d, 2000);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(wgs * 8);
    (void)hipMemcpy(h.data(), d, wgs * 32, hipMemcpyDeviceToHost);
    int hist[4][4] = {};
    for (int b = 0; b < wgs; b++)
        for (int w = 0; w < 4; w++) hist[w][(h[(b * 4 + w) * 2] >> 4) & 3]++;
    for (int w = 0; w < 4; w++) printf("wave %d of the workgroup: SIMD0 %d SIMD1 %d SIMD2 %d SIMD3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    for (int b : {0, 1, 2, 8, 9, 256, 257, 512, 1024, 1280, 1535}) {
        unsigned hw = h[b * 8], xcc = h[b * 8 + 1];
        printf("wg %4d: xcc %u se %u sh %u cu %2u | simd of waves 0..3: %u %u %u %u\n", b, xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15,
               (h[b * 8] >> 4) & 3, (h[b * 8 + 2] >> 4) & 3, (h[b * 8 + 4] >> 4) & 3, (h[b * 8 + 6] >> 4) & 3);
    }
    return 0;
}
