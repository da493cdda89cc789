// valu_mix.hip -- issue cost on gfx950 of every VALU instruction class the march step uses.
// Each test runs 64 instances of one instruction per loop iteration on 4 independent register chains,
// 4 waves per SIMD, and reports cycles per wave-instruction per SIMD (2.4 GHz nominal).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_mix valu_mix.hip ; run: ./valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITERS 4096
#define REP16(x) x x x x x x x x x x x x x x x x
#define FOUR(INS) REP16(asm volatile(INS : "+v"(u0) : "v"(ua), "v"(ub)); asm volatile(INS : "+v"(u1) : "v"(ua), "v"(ub)); \
                        asm volatile(INS : "+v"(u2) : "v"(ua), "v"(ub)); asm volatile(INS : "+v"(u3) : "v"(ua), "v"(ub));)

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned ua, unsigned ub)
{
    unsigned u0 = threadIdx.x + ua, u1 = threadIdx.x * 3 + ub, u2 = ua * 5 + threadIdx.x, u3 = ub * 7 + threadIdx.x;
    for (int i = 0; i < N_ITERS; ++i) {
        if (OP == 0) { FOUR("v_fma_f32 %0, %0, %1, %2\n") }
        if (OP == 1) { FOUR("v_add_u32 %0, %0, %1\n") }
        if (OP == 2) { FOUR("v_ashrrev_i32 %0, 1, %0\n") }
        if (OP == 3) { FOUR("v_add3_u32 %0, %0, %1, %2\n") }
        if (OP == 4) { FOUR("v_lshl_add_u32 %0, %0, 4, %1\n") }
        if (OP == 5) { FOUR("v_cvt_i32_f32 %0, %0\n") }
        if (OP == 6) { FOUR("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n") }
        if (OP == 7) { FOUR("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n") }
        if (OP == 8) { FOUR("v_mad_u32_u24 %0, %0, %1, %2\n") }
        if (OP == 9) { FOUR("v_max3_f32 %0, |%0|, |%1|, |%2|\n") }
        if (OP == 10) { REP16(asm volatile("v_cmp_gt_u32 vcc, %1, %0\n v_cmp_gt_u32 vcc, %2, %0\n" : "+v"(u0) : "v"(ua), "v"(ub) : "vcc"); asm volatile("v_cmp_gt_u32 vcc, %1, %0\n v_cmp_gt_u32 vcc, %2, %0\n" : "+v"(u1) : "v"(ua), "v"(ub) : "vcc");) }
        if (OP == 11) { FOUR("v_sub_f32 %0, 0x47800000, %0\n") }
        if (OP == 12) { FOUR("v_mul_f32 %0, %0, %1\n") }
        if (OP == 13) { FOUR("v_and_b32 %0, %0, %1\n") }
        if (OP == 14) { FOUR("v_cvt_f32_u32 %0, %0\n") }
        if (OP == 15) { FOUR("v_cvt_flr_i32_f32 %0, %0\n") }
        if (OP == 16) { FOUR("v_rndne_f32 %0, %0\n") }
        if (OP == 17) { FOUR("v_add_f32 %0, %0, %1\n") }
        if (OP == 18) { FOUR("v_cndmask_b32 %0, %0, %1, vcc\n") }
        if (OP == 19) { FOUR("v_bfe_i32 %0, %0, 1, 31\n") }
        if (OP == 20) { FOUR("v_mul_lo_u32 %0, %0, %1\n") }
        if (OP == 21) { FOUR("v_mul_hi_u32 %0, %0, %1\n") }
        if (OP == 22) {
            unsigned long long q0 = u0, q1 = u1, q2 = u2, q3 = u3;
            REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n" : "+v"(q0) : "v"(ua), "v"(ub) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n" : "+v"(q1) : "v"(ua), "v"(ub) : "vcc");
                  asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n" : "+v"(q2) : "v"(ua), "v"(ub) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n" : "+v"(q3) : "v"(ua), "v"(ub) : "vcc");)
            u0 = (unsigned)q0; u1 = (unsigned)q1; u2 = (unsigned)q2; u3 = (unsigned)q3;
        }
        if (OP == 23) { FOUR("v_mul_u32_u24 %0, %0, %1\n") }
        if (OP == 24) { REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %2, vcc\n" : "+v"(u0) : "v"(ua), "v"(ub) : "vcc"); asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %2, vcc\n" : "+v"(u1) : "v"(ua), "v"(ub) : "vcc");) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = u0 + u1 + u2 + u3;
}

template <int OP> void run(const char *name)
{
    const int waves_per_simd = 4;
    unsigned *out;
    const int blocks = 256 * waves_per_simd;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 0x3f800123u, 0x3f000456u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, 0x3f800123u, 0x3f000456u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)waves_per_simd * N_ITERS * 64;
    printf("%-22s %.2f cycles per wave-instruction\n", name, ms * 1e-3 * 2.4e9 / insts_per_simd);
    hipFree(out);
}

int main()
{
    run<0>("v_fma_f32"); run<12>("v_mul_f32"); run<17>("v_add_f32"); run<11>("v_sub_f32 (literal)");
    run<1>("v_add_u32"); run<13>("v_and_b32"); run<2>("v_ashrrev_i32"); run<19>("v_bfe_i32"); run<3>("v_add3_u32");
    run<4>("v_lshl_add_u32"); run<8>("v_mad_u32_u24"); run<5>("v_cvt_i32_f32"); run<15>("v_cvt_flr_i32_f32");
    run<14>("v_cvt_f32_u32"); run<6>("v_cvt_f32_u32_sdwa"); run<7>("v_sub_u32_sdwa"); run<16>("v_rndne_f32");
    run<9>("v_max3_f32 |abs|"); run<10>("v_cmp_gt_u32 (x2)"); run<18>("v_cndmask_b32");
    run<20>("v_mul_lo_u32"); run<21>("v_mul_hi_u32"); run<22>("v_mad_u64_u32"); run<23>("v_mul_u32_u24"); run<24>("v_add_co+v_addc_co");
    return 0;
}
