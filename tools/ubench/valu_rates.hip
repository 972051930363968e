// Issue rates of the vector instructions the render kernel leans on, measured on the device:
// every wavefront runs a long unrolled sequence of one instruction on eight independent register
// sets; enough wavefronts to fill every SIMD.  Prints cycles per wave64 instruction per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define BODY32(INS) \
    for (int it = 0; it < iters; ++it) { \
        asm volatile(REP8(REP8(INS)) : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3) : "v"(h0), "v"(h1) : "vcc", "s20", "s21", "s22", "s23"); \
    }
#define BODYPK(INS) \
    for (int it = 0; it < iters; ++it) { \
        asm volatile(REP8(REP8(INS)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(c0), "v"(c1) : "vcc", "s20", "s21", "s22", "s23"); \
    }
#define BODYMIX(INS) \
    for (int it = 0; it < iters; ++it) { \
        asm volatile(REP8(REP8(INS)) : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3) : "v"(c0), "v"(h1) : "vcc", "s20", "s21", "s22", "s23"); \
    }
#define BODY(INS) \
    for (int it = 0; it < iters; ++it) { \
        asm volatile(REP8(REP8(INS)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(c0), "v"(c1) : "vcc", "s20", "s21", "s22", "s23"); \
    }

template <int WHICH>
__global__ void k(double* out, int iters) {
    float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, g0 = 1.5f, g1 = 2.5f, g2 = 3.5f, g3 = 4.5f, h0 = 0.999f, h1 = 1e-3f;
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b0 = 1.5, b1 = 2.5, b2 = 3.5, b3 = 4.5, c0 = 0.999, c1 = 1e-3;
    if (WHICH == 0) BODY("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n")
    if (WHICH == 1) BODY("v_min_f64 %0, %0, %8\n v_min_f64 %1, %1, %8\n v_min_f64 %2, %2, %8\n v_min_f64 %3, %3, %8\n v_max_f64 %4, %4, %9\n v_max_f64 %5, %5, %9\n v_max_f64 %6, %6, %9\n v_max_f64 %7, %7, %9\n")
    if (WHICH == 2) BODY("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %9\n v_add_f64 %5, %5, %9\n v_add_f64 %6, %6, %9\n v_add_f64 %7, %7, %9\n")
    if (WHICH == 3) BODY("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n")
    if (WHICH == 4) BODY("v_cmp_gt_f64 vcc, %0, %8\n v_cmp_gt_f64 vcc, %1, %8\n v_cmp_gt_f64 vcc, %2, %8\n v_cmp_gt_f64 vcc, %3, %8\n v_cmp_gt_f64 vcc, %4, %8\n v_cmp_gt_f64 vcc, %5, %8\n v_cmp_gt_f64 vcc, %6, %8\n v_cmp_gt_f64 vcc, %7, %8\n")
    if (WHICH == 5) BODY("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7\n")
    if (WHICH == 6) BODY32("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
    if (WHICH == 7) BODY32("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
    if (WHICH == 8) BODYPK("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
    if (WHICH == 9) BODY32("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n")
    if (WHICH == 10) BODY("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n v_rsq_f64 %4, %4\n v_rsq_f64 %5, %5\n v_rsq_f64 %6, %6\n v_rsq_f64 %7, %7\n")
    if (WHICH == 11) BODY32("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")
    if (WHICH == 12) BODYMIX("v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %8\n v_cvt_f32_f64 %2, %8\n v_cvt_f32_f64 %3, %8\n v_cvt_f32_f64 %4, %8\n v_cvt_f32_f64 %5, %8\n v_cvt_f32_f64 %6, %8\n v_cvt_f32_f64 %7, %8\n")
    if (WHICH == 13) BODY("v_div_fixup_f64 %0, %0, %8, %9\n v_div_fixup_f64 %1, %1, %8, %9\n v_div_fixup_f64 %2, %2, %8, %9\n v_div_fixup_f64 %3, %3, %8, %9\n v_div_fixup_f64 %4, %4, %8, %9\n v_div_fixup_f64 %5, %5, %8, %9\n v_div_fixup_f64 %6, %6, %8, %9\n v_div_fixup_f64 %7, %7, %8, %9\n")
    if (WHICH == 14) BODY32("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
    if (WHICH == 15) BODY32("v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %8\n v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %8\n")

    if (WHICH == 16) BODY32("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n")
    if (WHICH == 17) BODY32("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n")
    if (WHICH == 18) BODY32("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n")
    if (WHICH == 19) BODY32("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n")
    if (WHICH == 20) BODY32("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n")
    if (WHICH == 21) BODY32("v_bfi_b32 %0, %8, %0, %9\n v_bfi_b32 %1, %8, %1, %9\n v_bfi_b32 %2, %8, %2, %9\n v_bfi_b32 %3, %8, %3, %9\n v_bfi_b32 %4, %8, %4, %9\n v_bfi_b32 %5, %8, %5, %9\n v_bfi_b32 %6, %8, %6, %9\n v_bfi_b32 %7, %8, %7, %9\n")
    if (WHICH == 22) BODY32("v_cndmask_b32_e64 %0, %8, %9, s[20:21]\n v_cndmask_b32_e64 %1, %8, %9, s[20:21]\n v_cndmask_b32_e64 %2, %8, %9, s[20:21]\n v_cndmask_b32_e64 %3, %8, %9, s[20:21]\n v_cndmask_b32_e64 %4, %8, %9, s[20:21]\n v_cndmask_b32_e64 %5, %8, %9, s[20:21]\n v_cndmask_b32_e64 %6, %8, %9, s[20:21]\n v_cndmask_b32_e64 %7, %8, %9, s[20:21]\n")
    if (WHICH == 23) BODY32("v_cmp_gt_f32_e64 s[22:23], %0, %8\n v_cmp_gt_f32_e64 s[22:23], %1, %8\n v_cmp_gt_f32_e64 s[22:23], %2, %8\n v_cmp_gt_f32_e64 s[22:23], %3, %8\n v_cmp_gt_f32_e64 s[22:23], %4, %8\n v_cmp_gt_f32_e64 s[22:23], %5, %8\n v_cmp_gt_f32_e64 s[22:23], %6, %8\n v_cmp_gt_f32_e64 s[22:23], %7, %8\n")
    if (WHICH == 24) BODY("v_cmp_gt_f64_e64 s[22:23], %0, %8\n v_cmp_gt_f64_e64 s[22:23], %1, %8\n v_cmp_gt_f64_e64 s[22:23], %2, %8\n v_cmp_gt_f64_e64 s[22:23], %3, %8\n v_cmp_gt_f64_e64 s[22:23], %4, %8\n v_cmp_gt_f64_e64 s[22:23], %5, %8\n v_cmp_gt_f64_e64 s[22:23], %6, %8\n v_cmp_gt_f64_e64 s[22:23], %7, %8\n")
    if (WHICH == 25) BODY32("v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc\n")

    if (WHICH == 26) BODY32("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_gt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_gt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n v_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %9, vcc\n v_cmp_gt_f32 vcc, %5, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_gt_f32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %9, vcc\n v_cmp_gt_f32 vcc, %7, %8\n v_cndmask_b32 %7, %7, %9, vcc\n")
    if (WHICH == 27) BODY32("v_cmp_gt_f32_e64 s[20:21], %0, %8\n v_cndmask_b32_e64 %0, %0, %9, s[20:21]\n v_cmp_gt_f32_e64 s[20:21], %1, %8\n v_cndmask_b32_e64 %1, %1, %9, s[20:21]\n v_cmp_gt_f32_e64 s[20:21], %2, %8\n v_cndmask_b32_e64 %2, %2, %9, s[20:21]\n v_cmp_gt_f32_e64 s[20:21], %3, %8\n v_cndmask_b32_e64 %3, %3, %9, s[20:21]\n v_cmp_gt_f32_e64 s[20:21], %4, %8\n v_cndmask_b32_e64 %4, %4, %9, s[20:21]\n v_cmp_gt_f32_e64 s[20:21], %5, %8\n v_cndmask_b32_e64 %5, %5, %9, s[20:21]\n v_cmp_gt_f32_e64 s[20:21], %6, %8\n v_cndmask_b32_e64 %6, %6, %9, s[20:21]\n v_cmp_gt_f32_e64 s[20:21], %7, %8\n v_cndmask_b32_e64 %7, %7, %9, s[20:21]\n")
    if (WHICH == 28) BODY32("v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc\n")
    if (WHICH == 29) BODY32("v_addc_co_u32 %0, vcc, %0, %8, vcc\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_addc_co_u32 %2, vcc, %2, %8, vcc\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n v_addc_co_u32 %4, vcc, %4, %8, vcc\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n v_addc_co_u32 %6, vcc, %6, %8, vcc\n v_addc_co_u32 %7, vcc, %7, %8, vcc\n")
    if (WHICH == 30) BODY32("v_cmp_gt_f32 vcc, %0, %8\n s_nop 0\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_gt_f32 vcc, %1, %8\n s_nop 0\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_gt_f32 vcc, %2, %8\n s_nop 0\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_gt_f32 vcc, %3, %8\n s_nop 0\n v_cndmask_b32 %3, %3, %9, vcc\n v_cmp_gt_f32 vcc, %4, %8\n s_nop 0\n v_cndmask_b32 %4, %4, %9, vcc\n v_cmp_gt_f32 vcc, %5, %8\n s_nop 0\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_gt_f32 vcc, %6, %8\n s_nop 0\n v_cndmask_b32 %6, %6, %9, vcc\n v_cmp_gt_f32 vcc, %7, %8\n s_nop 0\n v_cndmask_b32 %7, %7, %9, vcc\n")
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3 + f0 + f1 + f2 + f3 + g0 + g1 + g2 + g3;
}

template <int W>
void run(const char* name, double* d, int cus) {
    const int iters = 200, blocks = cus * 16, threads = 256;  // 64 wavefronts per CU: 16 per SIMD in turn
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<W><<<blocks, threads>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<W><<<blocks, threads>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: blocks * 4 waves / (cus * 4) waves, each iters * 64 * 8 instructions
    const double inst_per_simd = (double)blocks * 4 / (cus * 4) * iters * 64.0 * 8.0;
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("%-28s %.3f ms  %.2f cycles per instruction per SIMD (at %d MHz)\n", name, ms, ms * 1e-3 * clk * 1e3 / inst_per_simd, clk / 1000);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    double* d; hipMalloc(&d, (size_t)cus * 16 * 256 * 8);
    run<0>("v_fma_f64", d, cus); run<2>("v_add_f64", d, cus); run<3>("v_mul_f64", d, cus); run<1>("v_min/max_f64", d, cus);
    run<4>("v_cmp_gt_f64", d, cus); run<5>("v_rcp_f64", d, cus); run<10>("v_rsq_f64", d, cus); run<13>("v_div_fixup_f64", d, cus);
    run<6>("v_cndmask_b32", d, cus); run<11>("v_mov_b32", d, cus); run<7>("v_fma_f32", d, cus); run<8>("v_pk_fma_f32", d, cus);
    run<9>("v_min_f32/v_max3_f32", d, cus); run<12>("v_cvt_f32_f64", d, cus); run<14>("v_rcp_f32", d, cus); run<15>("v_cmp_gt_f32", d, cus);
    run<16>("v_cndmask_b32_e64 (sgpr mask)", d, cus); run<22>("v_cndmask_e64 no self dep", d, cus); run<25>("v_cndmask vcc no self dep", d, cus);
    run<17>("v_max_f32", d, cus); run<18>("v_max3_f32", d, cus); run<19>("v_sub_f32", d, cus); run<20>("v_and_b32", d, cus); run<21>("v_bfi_b32", d, cus);
    run<23>("v_cmp_gt_f32_e64 sgpr", d, cus); run<24>("v_cmp_gt_f64_e64 sgpr", d, cus);
    run<26>("pair: v_cmp vcc + v_cndmask vcc", d, cus); run<27>("pair: v_cmp sgpr + v_cndmask sgpr", d, cus);
    run<28>("v_cndmask_b32_e64 with vcc", d, cus); run<29>("v_addc_co_u32 vcc", d, cus); run<30>("pair vcc with s_nop", d, cus);
    return 0;
}
