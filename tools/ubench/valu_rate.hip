// valu_rate.hip -- microbenchmark: issue cost of scalar vs packed fp32 VALU ops and a few others on gfx950.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITERS 4096
#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, float a, float b)
{
    float v0 = a + threadIdx.x, v1 = b + threadIdx.x, v2 = a * 2, v3 = b * 3;
    float w0 = a + 1, w1 = b + 1, w2 = a + 2, w3 = b + 2;
    for (int i = 0; i < N_ITERS; ++i) {
        if (OP == 0) {   // 16 independent-ish v_fma_f32 (4 chains)
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n" : "+v"(v0) : "v"(a), "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %2\n" : "+v"(v1) : "v"(a), "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %2\n" : "+v"(v2) : "v"(a), "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %2\n" : "+v"(v3) : "v"(a), "v"(b));)
        } else if (OP == 1) {  // v_pk_fma_f32 on register pairs (4 chains)
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {v0, w0}, p1 = {v1, w1}, p2 = {v2, w2}, p3 = {v3, w3}, pa = {a, a}, pb = {b, b};
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n" : "+v"(p0) : "v"(pa), "v"(pb)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n" : "+v"(p1) : "v"(pa), "v"(pb)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n" : "+v"(p2) : "v"(pa), "v"(pb)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n" : "+v"(p3) : "v"(pa), "v"(pb));)
            v0 = p0.x + p0.y; v1 = p1.x + p1.y; v2 = p2.x + p2.y; v3 = p3.x + p3.y; w0 = p0.y; w1 = p1.y; w2 = p2.y; w3 = p3.y;
        } else if (OP == 2) {  // v_mul_f32
            REP16(asm volatile("v_mul_f32 %0, %0, %1\n" : "+v"(v0) : "v"(a)); asm volatile("v_mul_f32 %0, %0, %1\n" : "+v"(v1) : "v"(a)); asm volatile("v_mul_f32 %0, %0, %1\n" : "+v"(v2) : "v"(a)); asm volatile("v_mul_f32 %0, %0, %1\n" : "+v"(v3) : "v"(a));)
        } else if (OP == 3) {  // v_rcp_f32
            REP16(asm volatile("v_rcp_f32 %0, %0\n" : "+v"(v0)); asm volatile("v_rcp_f32 %0, %0\n" : "+v"(v1)); asm volatile("v_rcp_f32 %0, %0\n" : "+v"(v2)); asm volatile("v_rcp_f32 %0, %0\n" : "+v"(v3));)
        } else if (OP == 4) {  // v_cvt_i32_f32 / v_trunc mix
            REP16(asm volatile("v_trunc_f32 %0, %0\n" : "+v"(v0)); asm volatile("v_trunc_f32 %0, %0\n" : "+v"(v1)); asm volatile("v_trunc_f32 %0, %0\n" : "+v"(v2)); asm volatile("v_trunc_f32 %0, %0\n" : "+v"(v3));)
        } else if (OP == 5) {  // v_mad_u32_u24
            unsigned u0 = __float_as_uint(v0), u1 = __float_as_uint(v1), u2 = __float_as_uint(v2), u3 = __float_as_uint(v3), ua = __float_as_uint(a);
            REP16(asm volatile("v_mad_u32_u24 %0, %0, %1, %1\n" : "+v"(u0) : "v"(ua)); asm volatile("v_mad_u32_u24 %0, %0, %1, %1\n" : "+v"(u1) : "v"(ua)); asm volatile("v_mad_u32_u24 %0, %0, %1, %1\n" : "+v"(u2) : "v"(ua)); asm volatile("v_mad_u32_u24 %0, %0, %1, %1\n" : "+v"(u3) : "v"(ua));)
            v0 = __uint_as_float(u0); v1 = __uint_as_float(u1); v2 = __uint_as_float(u2); v3 = __uint_as_float(u3);
        } else if (OP == 6) {  // v_pk_mul_f32
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {v0, w0}, p1 = {v1, w1}, p2 = {v2, w2}, p3 = {v3, w3}, pa = {a, a};
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %1\n" : "+v"(p0) : "v"(pa)); asm volatile("v_pk_mul_f32 %0, %0, %1\n" : "+v"(p1) : "v"(pa)); asm volatile("v_pk_mul_f32 %0, %0, %1\n" : "+v"(p2) : "v"(pa)); asm volatile("v_pk_mul_f32 %0, %0, %1\n" : "+v"(p3) : "v"(pa));)
            v0 = p0.x + p0.y; v1 = p1.x + p1.y; v2 = p2.x + p2.y; v3 = p3.x + p3.y; w0 = p0.y; w1 = p1.y; w2 = p2.y; w3 = p3.y;
        } else if (OP == 7) {  // v_cndmask / v_cmp pair
            REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n" : "+v"(v0) : "v"(a) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n" : "+v"(v1) : "v"(a) : "vcc");)
        } else if (OP == 8) {  // v_mul_lo_u32
            unsigned u0 = __float_as_uint(v0), u1 = __float_as_uint(v1), u2 = __float_as_uint(v2), u3 = __float_as_uint(v3), ua = __float_as_uint(a);
            REP16(asm volatile("v_mul_lo_u32 %0, %0, %1\n" : "+v"(u0) : "v"(ua)); asm volatile("v_mul_lo_u32 %0, %0, %1\n" : "+v"(u1) : "v"(ua)); asm volatile("v_mul_lo_u32 %0, %0, %1\n" : "+v"(u2) : "v"(ua)); asm volatile("v_mul_lo_u32 %0, %0, %1\n" : "+v"(u3) : "v"(ua));)
            v0 = __uint_as_float(u0); v1 = __uint_as_float(u1); v2 = __uint_as_float(u2); v3 = __uint_as_float(u3);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + w0 + w1 + w2 + w3;
}

template <int OP> void run(const char *name, int ops_per_iter, int waves_per_simd)
{
    float *out;
    int blocks = 256 * waves_per_simd;   // 256 CUs, 256 threads = 4 waves = 1 per SIMD per block
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts_per_simd = (double)waves_per_simd * N_ITERS * ops_per_iter;
    double cycles = ms * 1e-3 * 2.4e9;
    printf("%-16s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz nominal)\n", name, waves_per_simd, ms, cycles / insts_per_simd);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", 64, w);
        run<1>("v_pk_fma_f32", 64, w);
        run<2>("v_mul_f32", 64, w);
        run<6>("v_pk_mul_f32", 64, w);
        run<3>("v_rcp_f32", 64, w);
        run<4>("v_trunc_f32", 64, w);
        run<5>("v_mad_u32_u24", 64, w);
        run<8>("v_mul_lo_u32", 64, w);
        run<7>("cmp+cndmask", 64, w);
    }
    return 0;
}
