// Micro-benchmark of instruction issue on gfx950: cycles per instruction per SIMD for a few VALU forms,
// at 1 and 4 waves per SIMD, independent streams and dependent chains.  Development tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = 1.0001f, b1 = 0.9999f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {b0, b1};
    f2 p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 8 independent v_fma_f32
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
        } else if (MODE == 1) {  // dependent chain v_fma_f32
            asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(b0), "v"(b1));
        } else if (MODE == 2) {  // 8 independent v_pk_fma_f32
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                              "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q));
        } else if (MODE == 3) {  // 8 independent v_pk_mul_f32
            asm volatile(REP8("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                              "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q));
        } else if (MODE == 4) {  // 8 independent v_add_f32_dpp row_ror:8
            asm volatile(REP8("v_add_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %2, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %4, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %6, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
        } else if (MODE == 5) {  // 8 independent v_cndmask (vcc)
            asm volatile(REP8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "vcc");
        } else if (MODE == 6) {  // v_cmp into sgpr pair + s_and + s_cmp + not-taken branch, x16
            asm volatile(REP8("v_cmp_gt_f32 s[20:21], %0, %1\n s_and_b64 s[20:21], s[20:21], exec\n s_cmp_eq_u64 s[20:21], 0\n s_cbranch_scc1 1f\n"
                              "v_cmp_gt_f32 s[20:21], %1, %0\n s_and_b64 s[20:21], s[20:21], exec\n s_cmp_eq_u64 s[20:21], 0\n s_cbranch_scc1 1f\n")
                         "1:\n" : : "v"(a0), "v"(b0) : "s20", "s21", "scc");
        } else if (MODE == 7) {  // dependent chain of v_pk_mul_f32
            asm volatile(REP64("v_pk_mul_f32 %0, %0, %1\n") : "+v"(p0) : "v"(q));
        } else if (MODE == 8) {  // taken branches
            asm volatile(REP8("s_branch 1f\n s_nop 0\n1:\n s_branch 2f\n s_nop 0\n2:\n s_branch 3f\n s_nop 0\n3:\n s_branch 4f\n s_nop 0\n4:\n"
                              "s_branch 5f\n s_nop 0\n5:\n s_branch 6f\n s_nop 0\n6:\n s_branch 7f\n s_nop 0\n7:\n s_branch 8f\n s_nop 0\n8:\n") : :);
        } else if (MODE == 10) {  // v_cndmask e64 with an SGPR-pair mask, 8 independent
            asm volatile(REP8("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n"
                              "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "s20", "s21");
        } else if (MODE == 11) {  // v_mov_b32 8 independent
            asm volatile(REP8("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
                              "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
        } else if (MODE == 12) {  // v_cmp_gt_f32 into distinct SGPR pairs
            asm volatile(REP8("v_cmp_gt_f32 s[20:21], %0, %1\n v_cmp_gt_f32 s[22:23], %0, %1\n v_cmp_gt_f32 s[24:25], %0, %1\n v_cmp_gt_f32 s[26:27], %0, %1\n"
                              "v_cmp_gt_f32 s[20:21], %1, %0\n v_cmp_gt_f32 s[22:23], %1, %0\n v_cmp_gt_f32 s[24:25], %1, %0\n v_cmp_gt_f32 s[26:27], %1, %0\n")
                         : : "v"(a0), "v"(b0) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if (MODE == 13) {  // v_mul_f32 + v_add_f32 mix, 8 independent
            asm volatile(REP8("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                              "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
        } else if (MODE == 14) {  // s_and_b64 chain (SALU)
            asm volatile(REP64("s_and_b64 s[20:21], s[20:21], exec\n") : : : "s20", "s21", "scc");
        } else if (MODE == 15) {  // v_add_f32_dpp quad_perm, 8 independent
            asm volatile(REP8("v_add_f32_dpp %0, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %2, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %4, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %6, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
        } else if (MODE == 16) {  // v_permlane32_swap
            asm volatile(REP8("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                              "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 17) {  // v_cmp -> vcc, v_cndmask_e32 vcc (pairs)
            asm volatile(REP8("v_cmp_gt_f32 vcc, %8, %0\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_gt_f32 vcc, %8, %1\n v_cndmask_b32 %1, %1, %8, vcc\n"
                              "v_cmp_gt_f32 vcc, %8, %2\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_gt_f32 vcc, %8, %3\n v_cndmask_b32 %3, %3, %8, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "vcc");
        } else if (MODE == 18) {  // v_cmp -> s[20:21], v_cndmask_e64 (pairs)
            asm volatile(REP8("v_cmp_gt_f32 s[20:21], %8, %0\n v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cmp_gt_f32 s[22:23], %8, %1\n v_cndmask_b32_e64 %1, %1, %8, s[22:23]\n"
                              "v_cmp_gt_f32 s[20:21], %8, %2\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cmp_gt_f32 s[22:23], %8, %3\n v_cndmask_b32_e64 %3, %3, %8, s[22:23]\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "s20", "s21", "s22", "s23");
        } else if (MODE == 19) {  // s_and_b64 vcc + v_cndmask_e32 vcc
            asm volatile(REP8("s_and_b64 vcc, s[20:21], exec\n v_cndmask_b32 %0, %0, %8, vcc\n s_and_b64 vcc, s[20:21], exec\n v_cndmask_b32 %1, %1, %8, vcc\n"
                              "s_and_b64 vcc, s[20:21], exec\n v_cndmask_b32 %2, %2, %8, vcc\n s_and_b64 vcc, s[20:21], exec\n v_cndmask_b32 %3, %3, %8, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "vcc", "s20", "s21", "scc");
        } else if (MODE == 20) {  // s_and_b64 s[22:23] + v_cndmask_e64
            asm volatile(REP8("s_and_b64 s[22:23], s[20:21], exec\n v_cndmask_b32_e64 %0, %0, %8, s[22:23]\n s_and_b64 s[22:23], s[20:21], exec\n v_cndmask_b32_e64 %1, %1, %8, s[22:23]\n"
                              "s_and_b64 s[22:23], s[20:21], exec\n v_cndmask_b32_e64 %2, %2, %8, s[22:23]\n s_and_b64 s[22:23], s[20:21], exec\n v_cndmask_b32_e64 %3, %3, %8, s[22:23]\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "s22", "s23", "s20", "s21", "scc");
        } else if (MODE == 9) {  // dependent LDS broadcast reads (address from the previous read)
            __shared__ float4 sm[256];
            if (i == 0) { sm[threadIdx.x] = make_float4(0, 0, 0, 0); __syncthreads(); }
            int addr = 0;
            asm volatile(REP64("ds_read_b128 v[40:43], %0\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 s20, v40\n v_mov_b32 %0, s20\n")
                         : "+v"(addr) : : "v40", "v41", "v42", "v43", "s20");
            a0 += addr;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x;
}

template <int MODE>
void run(const char* name, int per_iter, float* out, int wg_per_cu) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    k<MODE><<<grid, 256>>>(out, 10);
    hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const double cycles = ms * 1e-3 * clk_khz * 1e3;
    // each SIMD runs wg_per_cu waves (a 256-thread workgroup puts one wave on each SIMD)
    printf("%-34s waves/SIMD %d: %.2f cycles per instruction per SIMD (%.2f per wave)\n", name, wg_per_cu,
           cycles / ((double)iters * per_iter * wg_per_cu), cycles / ((double)iters * per_iter));
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("clock %d MHz\n", clk_khz / 1000);
    for (int w : {1, 4}) {
        if (w == 1) {
            run<0>("v_fma_f32 x8 independent", 64, out, 1); run<1>("v_fma_f32 dependent chain", 64, out, 1);
            run<2>("v_pk_fma_f32 x8 independent", 64, out, 1); run<3>("v_pk_mul_f32 x8 independent", 64, out, 1);
            run<7>("v_pk_mul_f32 dependent chain", 64, out, 1);
            run<4>("v_add_f32_dpp x8 independent", 64, out, 1); run<5>("v_cndmask x8 independent", 64, out, 1);
            run<6>("v_cmp+s_and+s_cmp+branch(nt) group", 16, out, 1); run<8>("s_branch taken (+skipped nop)", 64, out, 1);
            run<9>("ds_read_b128 dependent round trip", 64, out, 1);
            run<10>("v_cndmask_e64 sgpr mask x8", 64, out, 1); run<11>("v_mov_b32 x8", 64, out, 1); run<12>("v_cmp_gt_f32 -> sgpr pairs", 64, out, 1);
            run<13>("v_mul/v_add mix x8", 64, out, 1); run<14>("s_and_b64 chain", 64, out, 1); run<15>("v_add_f32_dpp quad_perm x8", 64, out, 1);
            run<16>("v_permlane32_swap x8", 64, out, 1);
            run<17>("v_cmp vcc + v_cndmask_e32 (pair)", 64, out, 1); run<18>("v_cmp sgpr + v_cndmask_e64 (pair)", 64, out, 1);
            run<19>("s_and vcc + v_cndmask_e32 (pair)", 64, out, 1); run<20>("s_and sgpr + v_cndmask_e64 (pair)", 64, out, 1);

        } else {
            run<0>("v_fma_f32 x8 independent", 64, out, 4); run<1>("v_fma_f32 dependent chain", 64, out, 4);
            run<2>("v_pk_fma_f32 x8 independent", 64, out, 4); run<3>("v_pk_mul_f32 x8 independent", 64, out, 4);
            run<7>("v_pk_mul_f32 dependent chain", 64, out, 4);
            run<4>("v_add_f32_dpp x8 independent", 64, out, 4); run<5>("v_cndmask x8 independent", 64, out, 4);
            run<6>("v_cmp+s_and+s_cmp+branch(nt) group", 16, out, 4); run<8>("s_branch taken (+skipped nop)", 64, out, 4);
            run<9>("ds_read_b128 dependent round trip", 64, out, 4);
            run<10>("v_cndmask_e64 sgpr mask x8", 64, out, 4); run<11>("v_mov_b32 x8", 64, out, 4); run<12>("v_cmp_gt_f32 -> sgpr pairs", 64, out, 4);
            run<13>("v_mul/v_add mix x8", 64, out, 4); run<14>("s_and_b64 chain", 64, out, 4); run<15>("v_add_f32_dpp quad_perm x8", 64, out, 4);
            run<16>("v_permlane32_swap x8", 64, out, 4);
            run<17>("v_cmp vcc + v_cndmask_e32 (pair)", 64, out, 4); run<18>("v_cmp sgpr + v_cndmask_e64 (pair)", 64, out, 4);
            run<19>("s_and vcc + v_cndmask_e32 (pair)", 64, out, 4); run<20>("s_and sgpr + v_cndmask_e64 (pair)", 64, out, 4);

        }
    }
    return 0;
}
