// step_rate.hip -- speed of light of the march step on gfx950: the production step (vrt_march.h, corners read from an
// LDS brick exactly as the slab kernel's inner loop does) run back to back with no planning, no staging and no divergence,
// at 1..8 wavefronts per SIMD.  Reports shader cycles per wave-step per SIMD at the clock the chip really holds
// (s_memtime / s_memrealtime), i.e. the floor the slab kernel's marching phase can reach with this instruction mix.
//   MODE 0  step + in-brick test (the slab kernel's loop body)
//   MODE 1  step without the in-brick test
//   MODE 2  step on corners held in registers (no LDS reads): the pure VALU cost
//   MODE 6  MODE 3 with TWO independent rays per lane in the same brick (two dependent chains interleaved): what 128-ray
//           groups would cost per PAIR of wave-steps
//   MODE 3  the production PROVEN loop body (vrt_slab.h): address by v_mad_u32_u16, eight ds_read_b128 from a prescaled brick,
//           march_step_now without window test, unit invscale -- 79 VALU instructions + the 3 of the fold-back below
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -o step_rate step_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../volumeraytracer_amd/csrc/vrt_slab.h"

using namespace vrt;

template <int MODE, int OCC>
__global__ __launch_bounds__(256, OCC) void k_step(const vox_f32 *src, uint32_t *out, int nsteps, unsigned long long *clk,
                                                   float invx, float invy, float invz, uint32_t win_lo, uint32_t win_span)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    vox_f32 *seg = (vox_f32 *)smem + (size_t)wave * 384;
    for (int j = 0; j < 6; ++j) seg[lane + 64 * j] = src[lane + 64 * j];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // brick 8 x 6 x 8; rays start in cell (1..2, 1..2, 1..2) and are folded back every step
    const uint32_t ey = 6, ez = 8, ex1 = 7, ey1 = 5, ez1 = 7;
    Ray r;
    r.px = 0x10000u + lane * 600u; r.py = 0x18000u + lane * 300u; r.pz = 0x14000u + lane * 450u;
    r.dx = 65536.0f; r.dy = 100.0f * (float)lane; r.dz = -50.0f * (float)lane;
    const StepWindow win = {win_lo, win_span};
    uint32_t it = 0x7FFFFFFFu;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    f4 creg[8];
    if (MODE == 2) gather_lds(seg, 60, ez, ey * ez, creg);
    const uint32_t sy = ez, sx = ey * ez;
    const uint64_t m_pos0 = lane_mask(it > 0);
    Ray r2 = r;
    r2.px += 0x4000u; r2.py += 0x2000u; r2.dy = -r.dy;
    for (int k = 0; k < nsteps; ++k) {
        const uint32_t cx = r.px >> 16, cy = r.py >> 16, cz = r.pz >> 16;
        if (MODE == 6) {
            const uint32_t sy16 = sy * 16u, sx16 = sx * 16u, seg_addr = lds_addr(seg, 0);
            uint32_t a0 = mad_hi16(r.pz, 16u, mad_hi16(r.py, sy16, mad_hi16(r.px, sx16, seg_addr)));
            uint32_t a1 = a0 + sy16, a2 = a0 + sx16, a3 = a2 + sy16;
            uint32_t b0 = mad_hi16(r2.pz, 16u, mad_hi16(r2.py, sy16, mad_hi16(r2.px, sx16, seg_addr)));
            uint32_t b1 = b0 + sy16, b2 = b0 + sx16, b3 = b2 + sy16;
            Weights w = lerp_weights(r), w2 = lerp_weights(r2);
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) :: "memory");
            vox_f32 v[8], u[8];
            v[0] = lds_at<vox_f32>(a0)[0]; v[1] = lds_at<vox_f32>(a0)[1]; v[2] = lds_at<vox_f32>(a1)[0]; v[3] = lds_at<vox_f32>(a1)[1];
            v[4] = lds_at<vox_f32>(a2)[0]; v[5] = lds_at<vox_f32>(a2)[1]; v[6] = lds_at<vox_f32>(a3)[0]; v[7] = lds_at<vox_f32>(a3)[1];
            u[0] = lds_at<vox_f32>(b0)[0]; u[1] = lds_at<vox_f32>(b0)[1]; u[2] = lds_at<vox_f32>(b1)[0]; u[3] = lds_at<vox_f32>(b1)[1];
            u[4] = lds_at<vox_f32>(b2)[0]; u[5] = lds_at<vox_f32>(b2)[1]; u[6] = lds_at<vox_f32>(b3)[0]; u[7] = lds_at<vox_f32>(b3)[1];
            f4 c[8], e[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { cvt_vox(v[q], c[q]); cvt_vox(u[q], e[q]); }
            march_step_now<true, false, true>(r, c, w, invx, invy, invz, win);
            march_step_now<true, false, true>(r2, e, w2, invx, invy, invz, win);
            --it;
            r.px = (r.px & 0x1FFFFu) | 0x10000u; r.py = (r.py & 0x1FFFFu) | 0x10000u; r.pz = (r.pz & 0x1FFFFu) | 0x10000u;
            r2.px = (r2.px & 0x1FFFFu) | 0x10000u; r2.py = (r2.py & 0x1FFFFu) | 0x10000u; r2.pz = (r2.pz & 0x1FFFFu) | 0x10000u;
            continue;
        }
        if (MODE == 0) {
            const uint64_t m_inb = lane_mask(cx < ex1) & lane_mask(cy < ey1) & lane_mask(cz < ez1);
            if ((m_pos0 & ~m_inb) != 0ull) {
                if (any_lane(!(cx < ex1 && cy < ey1 && cz < ez1) && r.px < 0x80000000u)) break;
            }
            if ((m_pos0 & m_inb) == 0ull) break;
            if (cx < ex1 && cy < ey1 && cz < ez1) {
                f4 c[8];
                gather_lds(seg, mad24(mad24(cx, ey, cy), ez, cz), sy, sx, c);
                (void)march_step(r, c, invx, invy, invz, win, false);
                --it;
            }
        } else if (MODE == 1) {
            f4 c[8];
            gather_lds(seg, mad24(mad24(cx, ey, cy), ez, cz), sy, sx, c);
            (void)march_step(r, c, invx, invy, invz, win, false);
            --it;
        } else if (MODE == 4 || MODE == 5) {   // scheduling variants of MODE 3: 4 = no pinning at all, 5 = addresses pinned, weights after the reads
            const uint32_t sy16 = sy * 16u, sx16 = sx * 16u, seg_addr = lds_addr(seg, 0);
            uint32_t a0 = mad_hi16(r.pz, 16u, mad_hi16(r.py, sy16, mad_hi16(r.px, sx16, seg_addr)));
            uint32_t a1 = a0 + sy16, a2 = a0 + sx16, a3 = a2 + sy16;
            if (MODE == 5) asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "memory");
            vox_f32 v[8];
            v[0] = lds_at<vox_f32>(a0)[0]; v[1] = lds_at<vox_f32>(a0)[1]; v[2] = lds_at<vox_f32>(a1)[0]; v[3] = lds_at<vox_f32>(a1)[1];
            v[4] = lds_at<vox_f32>(a2)[0]; v[5] = lds_at<vox_f32>(a2)[1]; v[6] = lds_at<vox_f32>(a3)[0]; v[7] = lds_at<vox_f32>(a3)[1];
            Weights w = lerp_weights(r);
            if (MODE == 5) asm volatile("" : "+v"(w.wrx), "+v"(w.wlx), "+v"(w.wry), "+v"(w.wly), "+v"(w.wrz), "+v"(w.wlz));
            f4 c[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) cvt_vox(v[q], c[q]);
            march_step_now<true, false, true>(r, c, w, invx, invy, invz, win);
            --it;
        } else if (MODE == 3) {
            const uint32_t sy16 = sy * 16u, sx16 = sx * 16u, seg_addr = lds_addr(seg, 0);
            uint32_t a0 = mad_hi16(r.pz, 16u, mad_hi16(r.py, sy16, mad_hi16(r.px, sx16, seg_addr)));
            uint32_t a1 = a0 + sy16, a2 = a0 + sx16, a3 = a2 + sy16;
            Weights w = lerp_weights(r);
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(w.wrx), "+v"(w.wlx), "+v"(w.wry), "+v"(w.wly), "+v"(w.wrz), "+v"(w.wlz) :: "memory");
            vox_f32 v[8];
            v[0] = lds_at<vox_f32>(a0)[0]; v[1] = lds_at<vox_f32>(a0)[1]; v[2] = lds_at<vox_f32>(a1)[0]; v[3] = lds_at<vox_f32>(a1)[1];
            v[4] = lds_at<vox_f32>(a2)[0]; v[5] = lds_at<vox_f32>(a2)[1]; v[6] = lds_at<vox_f32>(a3)[0]; v[7] = lds_at<vox_f32>(a3)[1];
            f4 c[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) cvt_vox(v[q], c[q]);
            march_step_now<true, false, true>(r, c, w, invx, invy, invz, win);
            --it;
        } else {
            (void)march_step(r, creg, invx, invy, invz, win, false);
            --it;
        }
        // fold the ray back into cells 1..2 (3 VALU the production loop does not have)
        r.px = (r.px & 0x1FFFFu) | 0x10000u; r.py = (r.py & 0x1FFFFu) | 0x10000u; r.pz = (r.pz & 0x1FFFFu) | 0x10000u;
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = r.px + r.py + r.pz + it + __float_as_uint(r.dx + r.dy + r.dz) +
                                                 (MODE == 6 ? r2.px + r2.py + r2.pz + __float_as_uint(r2.dx + r2.dy + r2.dz) : 0u);
    if (lane == 0) { atomicAdd(clk, t1 - t0); atomicAdd(clk + 1, r1 - r0); }
}

template <int MODE, int OCC> void run(const char *name, const vox_f32 *src)
{
    const int nsteps = 20000;
    const int blocks = 256 * OCC;
    uint32_t *out;
    unsigned long long *clk, h[2];
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = 4 * 384 * 16;
    k_step<MODE, OCC><<<blocks, 256, lds>>>(src, out, nsteps, clk, 1.0f, 1.0f, 1.0f, 32u << 23, (251u - 32u) << 23);
    hipDeviceSynchronize();
    hipMemset(clk, 0, 16);
    hipEventRecord(e0);
    k_step<MODE, OCC><<<blocks, 256, lds>>>(src, out, nsteps, clk, 1.0f, 1.0f, 1.0f, 32u << 23, (251u - 32u) << 23);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double clock = (double)h[0] / (double)h[1] * 100e6;
    const double wave_steps_per_simd = (double)OCC * nsteps;
    printf("%-28s waves/SIMD=%d  %.3f ms  clock %.3f GHz  %.1f cycles per wave-step per SIMD (%.1f at 2.4 GHz nominal)\n", name, OCC, ms,
           clock / 1e9, ms * 1e-3 * clock / wave_steps_per_simd, ms * 1e-3 * 2.4e9 / wave_steps_per_simd);
    hipFree(out); hipFree(clk);
}

int main()
{
    std::vector<float> h(384 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (i % 4 == 3) ? -32767.0f : (float)((int)(i * 2654435761u % 2001u) - 1000) * 1.0e-3f;
    vox_f32 *src;
    hipMalloc(&src, h.size() * 4);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0, 1>("step+test (LDS)", src); run<0, 2>("step+test (LDS)", src); run<0, 3>("step+test (LDS)", src); run<0, 4>("step+test (LDS)", src); run<0, 5>("step+test (LDS)", src); run<0, 6>("step+test (LDS)", src);
    run<1, 1>("step (LDS)", src); run<1, 2>("step (LDS)", src); run<1, 3>("step (LDS)", src); run<1, 4>("step (LDS)", src); run<1, 5>("step (LDS)", src); run<1, 6>("step (LDS)", src);
    run<3, 1>("proven loop body (LDS)", src); run<3, 2>("proven loop body (LDS)", src); run<3, 3>("proven loop body (LDS)", src); run<3, 4>("proven loop body (LDS)", src); run<3, 5>("proven loop body (LDS)", src); run<3, 6>("proven loop body (LDS)", src);
    run<4, 2>("proven body, unpinned", src); run<4, 3>("proven body, unpinned", src); run<4, 4>("proven body, unpinned", src); run<4, 6>("proven body, unpinned", src);
    run<5, 2>("proven body, reads first", src); run<5, 3>("proven body, reads first", src); run<5, 4>("proven body, reads first", src); run<5, 6>("proven body, reads first", src);
    run<6, 1>("two rays per lane (per PAIR)", src); run<6, 2>("two rays per lane (per PAIR)", src); run<6, 3>("two rays per lane (per PAIR)", src); run<6, 4>("two rays per lane (per PAIR)", src);
    run<2, 1>("step (registers)", src); run<2, 2>("step (registers)", src); run<2, 3>("step (registers)", src); run<2, 4>("step (registers)", src); run<2, 5>("step (registers)", src); run<2, 6>("step (registers)", src);
    return 0;
}
