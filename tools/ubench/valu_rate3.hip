// Microbenchmark 3: per-instruction VALU issue cost on gfx950 for the instruction kinds of the finishing stage
// (several waves per SIMD; inline asm, four independent chains).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(X) X X X X X X X X
#define CHAIN4(OP) asm volatile(OP(0) "\n" OP(1) "\n" OP(2) "\n" OP(3) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "vcc");
#define S(x) #x
#define OP_FMA(i) "v_fma_f32 %" S(i) ", %" S(i) ", %4, %5"
#define OP_MUL(i) "v_mul_f32 %" S(i) ", %" S(i) ", %4"
#define OP_ADD(i) "v_add_f32 %" S(i) ", %" S(i) ", %5"
#define OP_SUB(i) "v_sub_f32 %" S(i) ", %" S(i) ", %5"
#define OP_MAX(i) "v_max_f32 %" S(i) ", %" S(i) ", %5"
#define OP_MIN(i) "v_min_f32 %" S(i) ", %" S(i) ", %4"
#define OP_MAX3(i) "v_max3_f32 %" S(i) ", %" S(i) ", %4, %5"
#define OP_MED3(i) "v_med3_f32 %" S(i) ", %" S(i) ", %4, %5"
#define OP_MINIMUM3(i) "v_minimum3_f32 %" S(i) ", %" S(i) ", %4, %5"
#define OP_CMP(i) "v_cmp_lt_f32 vcc, %" S(i) ", %4"
#define OP_CND(i) "v_cndmask_b32 %" S(i) ", %" S(i) ", %4, vcc"
#define OP_AND(i) "v_and_b32 %" S(i) ", %" S(i) ", %4"
#define OP_XOR(i) "v_xor_b32 %" S(i) ", %" S(i) ", %4"
#define OP_BFI(i) "v_bfi_b32 %" S(i) ", %" S(i) ", %4, %5"
#define OP_ADDU(i) "v_add_u32 %" S(i) ", %" S(i) ", %4"
#define OP_LSHL(i) "v_lshlrev_b32 %" S(i) ", 1, %" S(i)
#define OP_MOV(i) "v_mov_b32 %" S(i) ", %4"
#define OP_RND(i) "v_rndne_f32 %" S(i) ", %" S(i)
#define OP_CVT(i) "v_cvt_i32_f32 %" S(i) ", %" S(i)
#define OP_SIN(i) "v_sin_f32 %" S(i) ", %" S(i)
#define OP_RCP(i) "v_rcp_f32 %" S(i) ", %" S(i)
#define OP_FMAABS(i) "v_fma_f32 %" S(i) ", |%" S(i) "|, -%4, %5"
#define OP_MULLIT(i) "v_mul_f32 %" S(i) ", 0x3c8efa35, %" S(i)
#define OP_FMAAK(i) "v_fmaak_f32 %" S(i) ", %" S(i) ", %4, 0x3c8efa35"
#define OP_FMAC(i) "v_fmac_f32 %" S(i) ", %4, %5"
#define OP_CMPCND(i) "v_cmp_lt_f32 vcc, %" S(i) ", %4\n v_cndmask_b32 %" S(i) ", %" S(i) ", %5, vcc"
#define OP_CND64(i) "v_cndmask_b32 %" S(i) ", %" S(i) ", %4, s[20:21]"
#define OP_CMP64CND(i) "v_cmp_lt_f32 s[20:21], %" S(i) ", %4\n v_cndmask_b32 %" S(i) ", %" S(i) ", %5, s[20:21]"
#define OP_CMPMAX(i) "v_cmp_lt_f32 vcc, %" S(i) ", %4\n v_max_f32 %" S(i) ", %" S(i) ", %5"
#define OP_CNDFMA(i) "v_cndmask_b32 %" S(i) ", %" S(i) ", %4, vcc\n v_fma_f32 %" S(i) ", %" S(i) ", %4, %5"
#define OP_PKMUL(i) "v_pk_mul_f32 %" S(i) ", %" S(i) ", %4"

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { REP8(CHAIN4(OP_FMA)) }
        if (MODE == 1) { REP8(CHAIN4(OP_MUL)) }
        if (MODE == 2) { REP8(CHAIN4(OP_ADD)) }
        if (MODE == 3) { REP8(CHAIN4(OP_SUB)) }
        if (MODE == 4) { REP8(CHAIN4(OP_MAX)) }
        if (MODE == 5) { REP8(CHAIN4(OP_MIN)) }
        if (MODE == 6) { REP8(CHAIN4(OP_MAX3)) }
        if (MODE == 7) { REP8(CHAIN4(OP_MED3)) }
        if (MODE == 8) { REP8(CHAIN4(OP_MINIMUM3)) }
        if (MODE == 9) { REP8(CHAIN4(OP_CMP)) }
        if (MODE == 10) { REP8(CHAIN4(OP_CND)) }
        if (MODE == 11) { REP8(CHAIN4(OP_AND)) }
        if (MODE == 12) { REP8(CHAIN4(OP_XOR)) }
        if (MODE == 13) { REP8(CHAIN4(OP_BFI)) }
        if (MODE == 14) { REP8(CHAIN4(OP_ADDU)) }
        if (MODE == 15) { REP8(CHAIN4(OP_LSHL)) }
        if (MODE == 16) { REP8(CHAIN4(OP_MOV)) }
        if (MODE == 17) { REP8(CHAIN4(OP_RND)) }
        if (MODE == 18) { REP8(CHAIN4(OP_CVT)) }
        if (MODE == 19) { REP8(CHAIN4(OP_SIN)) }
        if (MODE == 20) { REP8(CHAIN4(OP_RCP)) }
        if (MODE == 21) { REP8(CHAIN4(OP_FMAABS)) }
        if (MODE == 22) { REP8(CHAIN4(OP_MULLIT)) }
        if (MODE == 23) { REP8(CHAIN4(OP_FMAAK)) }
        if (MODE == 24) { REP8(CHAIN4(OP_FMAC)) }
        if (MODE == 25) { REP8(CHAIN4(OP_CMPCND)) }
        if (MODE == 26) { REP8(asm volatile(OP_CND64(0) "\n" OP_CND64(1) "\n" OP_CND64(2) "\n" OP_CND64(3) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "vcc", "s20", "s21");) }
        if (MODE == 27) { REP8(asm volatile(OP_CMP64CND(0) "\n" OP_CMP64CND(1) "\n" OP_CMP64CND(2) "\n" OP_CMP64CND(3) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "vcc", "s20", "s21");) }
        if (MODE == 28) { REP8(CHAIN4(OP_CMPMAX)) }
        if (MODE == 29) { REP8(CHAIN4(OP_CNDFMA)) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3;
}
template <int MODE>
void run(const char* name, int blocks_per_cu) {
    float* out;
    const int blocks = 256 * blocks_per_cu, iters = 2000, insts_per_iter = 32;
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
    printf("%-16s waves/SIMD %d  %.3f ms  cycles per wave-instr per SIMD @2.4GHz %.2f\n", name, blocks_per_cu, ms,
           ms * 1e6 / per_simd * 2.4);
    (void)hipFree(out);
}
int main() {
    for (int b : {2, 6}) {
        run<0>("v_fma_f32", b); run<1>("v_mul_f32", b); run<2>("v_add_f32", b); run<3>("v_sub_f32", b);
        run<4>("v_max_f32", b); run<5>("v_min_f32", b); run<6>("v_max3_f32", b); run<7>("v_med3_f32", b);
        run<8>("v_minimum3_f32", b); run<9>("v_cmp_lt_f32", b); run<10>("v_cndmask_b32", b); run<11>("v_and_b32", b);
        run<12>("v_xor_b32", b); run<13>("v_bfi_b32", b); run<14>("v_add_u32", b); run<15>("v_lshlrev_b32", b);
        run<16>("v_mov_b32", b); run<17>("v_rndne_f32", b); run<18>("v_cvt_i32_f32", b); run<19>("v_sin_f32", b);
        run<20>("v_rcp_f32", b); run<21>("v_fma |x|,-a", b); run<22>("v_mul literal", b); run<23>("v_fmaak", b);
        run<24>("v_fmac", b);
        run<25>("cmp+cnd vcc (x2)", b); run<26>("v_cndmask sgpr", b); run<27>("cmp+cnd sgpr (x2)", b); run<28>("cmp+max (x2)", b); run<29>("cnd+fma (x2)", b);
    }
    return 0;
}
