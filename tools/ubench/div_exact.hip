// div_exact.hip -- exhaustive proof run for the short division used in the march step.
//
// The reference computes ilen = 0x42000000p0f / dot with an IEEE divide (cuda_volume_raytracer.cu:346).
// The compiler's correctly rounded sequence costs 11 VALU instructions.  This program checks, for EVERY
// float bit pattern of the denominator, which shorter v_rcp_f32 + fma sequences give the bit-identical
// quotient of 2*C/d (the doubled quotient is what march_step uses), and prints, per biased exponent of d,
// the number of mismatches -- the kernel takes the short path only in an exponent window this run shows
// to be clean, and the full IEEE divide everywhere else.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o div_exact div_exact.hip ; run: ./div_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define C2 0x84000000p0f   /* 2 * 0x42000000 */

__device__ __forceinline__ float cand_a(float d)   // 4 ops
{
    const float r = __builtin_amdgcn_rcpf(d);
    float q = C2 * r;
    const float e = __builtin_fmaf(-d, q, C2);
    return __builtin_fmaf(e, r, q);
}
__device__ __forceinline__ float cand_b(float d)   // 6 ops: refine the reciprocal first
{
    float r = __builtin_amdgcn_rcpf(d);
    const float e0 = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e0, r, r);
    float q = C2 * r;
    const float e = __builtin_fmaf(-d, q, C2);
    return __builtin_fmaf(e, r, q);
}
__device__ __forceinline__ float cand_c(float d)   // 6 ops: two quotient refinements with the raw reciprocal
{
    const float r = __builtin_amdgcn_rcpf(d);
    float q = C2 * r;
    float e = __builtin_fmaf(-d, q, C2);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-d, q, C2);
    return __builtin_fmaf(e, r, q);
}

__global__ __launch_bounds__(256) void k(unsigned long long *bad /* [3][256] */)
{
    // one block per (biased exponent, sign); threads sweep the 2^23 mantissas
    const uint32_t expo = blockIdx.x & 255u, sign = blockIdx.x >> 8;
    unsigned long long ba = 0, bb = 0, bc = 0;
    for (uint32_t m = threadIdx.x; m < (1u << 23); m += blockDim.x) {
        const float d = __uint_as_float((sign << 31) | (expo << 23) | m);
        const float ref0 = 0x42000000p0f / d;
        const uint32_t ref = __float_as_uint(ref0 + ref0);
        ba += __float_as_uint(cand_a(d)) != ref;
        bb += __float_as_uint(cand_b(d)) != ref;
        bc += __float_as_uint(cand_c(d)) != ref;
    }
    if (sign == 0) {
        atomicAdd(&bad[0 * 256 + expo], ba);
        atomicAdd(&bad[1 * 256 + expo], bb);
        atomicAdd(&bad[2 * 256 + expo], bc);
    }
}

int main()
{
    unsigned long long *d_bad;
    hipMalloc(&d_bad, 3 * 256 * sizeof(unsigned long long));
    hipMemset(d_bad, 0, 3 * 256 * sizeof(unsigned long long));
    k<<<256, 256>>>(d_bad);                     // positive denominators (dot is a sum of squares)
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<unsigned long long> bad(3 * 256);
    hipMemcpy(bad.data(), d_bad, bad.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    const char *name[3] = {"A rcp,mul,fma,fma", "B rcp refined + 1 step", "C raw rcp + 2 steps"};
    for (int c = 0; c < 3; ++c) {
        int lo = -1, hi = -1, best_lo = -1, best_hi = -2;
        unsigned long long total = 0;
        for (int e = 0; e < 256; ++e) {
            total += bad[c * 256 + e];
            if (bad[c * 256 + e] == 0) { if (lo < 0) lo = e; hi = e; if (hi - lo > best_hi - best_lo) { best_lo = lo; best_hi = hi; } }
            else lo = -1;
        }
        printf("%-26s mismatches over all positive d: %llu ; longest clean biased-exponent window: [%d, %d]\n",
               name[c], total, best_lo, best_hi);
        printf("  per-exponent mismatch counts (nonzero only):");
        for (int e = 0; e < 256; ++e) if (bad[c * 256 + e]) printf(" %d:%llu", e, bad[c * 256 + e]);
        printf("\n");
    }
    return 0;
}
