// vrt_slab.h -- Kernel 2: SLAB.  Wave-cooperative LDS staging of the voxel brick a
// wavefront is about to walk.
//
// Same per-ray arithmetic as the DIRECT kernel (vrt_march.h, bit-identical results); what
// changes is where the 8 trilinear corners come from.  The reference's GPU path gathers them
// from global memory on every step (/root/reference/src/cuda_volume_raytracer.cu:138-143 via
// :397-414).  Here each 64-lane wavefront owns a private LDS segment and repeats:
//
//   plan   wave-reduce (DPP) the live lanes' positions and per-step displacement; pick the
//          largest step count K whose swept bounding brick (+1 voxel for the upper trilinear
//          corners), clipped to the field, fits the segment
//   stage  all 64 lanes copy the brick HBM/L2 -> LDS with 16-byte accesses (z is the
//          memory-minor axis, so brick rows are contiguous runs)
//   march  up to K steps with all corners read from LDS (ds_read_b128); a ballot ends the
//          round as soon as any live lane would leave the brick (rays bend, the plan is only
//          a prediction) and ends the wavefront when every lane has terminated
//
// Wavefronts whose rays do not fit a brick even for one step (incoherent rays) fall back to
// the global gather for a few steps and re-plan.  Waves never synchronise with each other:
// a 256-thread workgroup is four independent wavefronts with four LDS segments.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vrt_march.h"

extern "C" __device__ unsigned int __ockl_wfred_min_u32(unsigned int);
extern "C" __device__ unsigned int __ockl_wfred_max_u32(unsigned int);
extern "C" __device__ int __ockl_wfred_min_i32(int);
extern "C" __device__ int __ockl_wfred_max_i32(int);

namespace vrt {

constexpr int kSlabWavesPerBlock = 4;
constexpr int kSlabMaxK = 192;          // longest planned round, steps
constexpr int kSlabDirectSteps = 16;    // global-gather steps for a wave that fits no brick

__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

struct Brick {
    uint32_t lox, loy, loz;   // first voxel
    uint32_t ex, ey, ez;      // extents in voxels (>= 2 each when usable)
};

// Plan the brick for K steps from wave-uniform extremes.  Positions are 16.16 in uint32, the
// displacement extremes are int32 per step.  Returns the voxel count (0 if degenerate).
__device__ __forceinline__ uint32_t plan_brick(const uint32_t pmin[3], const uint32_t pmax[3], const int32_t smin[3],
                                               const int32_t smax[3], int K, const uint32_t B[3], Brick &b)
{
    uint32_t lo[3], ext[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        // positions before each of the K steps: k = 0 .. K-1
        int64_t a = (int64_t)pmin[d] + (int64_t)(K - 1) * (smin[d] < 0 ? smin[d] : 0);
        int64_t c = (int64_t)pmax[d] + (int64_t)(K - 1) * (smax[d] > 0 ? smax[d] : 0);
        if (a < 0) a = 0;
        int64_t l = a >> 16;
        int64_t h = (c >> 16) + 1;                       // +1: upper trilinear corner
        const int64_t top = (int64_t)B[d] - 1;
        if (h > top) h = top;
        if (l > h) l = h;
        lo[d] = (uint32_t)l;
        ext[d] = (uint32_t)(h - l + 1);
    }
    b.lox = lo[0]; b.loy = lo[1]; b.loz = lo[2];
    b.ex = ext[0]; b.ey = ext[1]; b.ez = ext[2];
    const uint64_t vol = (uint64_t)ext[0] * ext[1] * ext[2];
    return vol > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)vol;
}

template <typename FieldT>
__device__ __forceinline__ void gather_lds(const typename FieldTraits<FieldT>::vox *seg, uint32_t base, uint32_t sy,
                                           uint32_t sx, f4 (&c)[8])
{
    using vox = typename FieldTraits<FieldT>::vox;
    const vox *b = seg + base;
    vox v[8];
    v[0] = b[0];       v[1] = b[1];
    v[2] = b[sy];      v[3] = b[sy + 1];
    v[4] = b[sx];      v[5] = b[sx + 1];
    v[6] = b[sx + sy]; v[7] = b[sx + sy + 1];
#pragma unroll
    for (int k = 0; k < 8; ++k) cvt_vox(v[k], c[k]);
}

template <typename FieldT, typename DirT>
__global__ __launch_bounds__(64 * kSlabWavesPerBlock) void k_march_slab(const MarchParams p)
{
    using vox = typename FieldTraits<FieldT>::vox;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t wave = threadIdx.x >> 6;
    vox *seg = (vox *)(smem + (size_t)wave * p.slab_lds_bytes);
    const uint32_t cap_vox = p.slab_lds_bytes / sizeof(vox);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    const bool have = slot < p.nrays;
    const uint32_t ray = have ? (p.order ? p.order[slot] : slot) : 0u;
    Ray r;
    r.px = r.py = r.pz = 0; r.dx = r.dy = r.dz = 1.0f;
    if (have) load_ray<DirT>(p, ray, r);
    const vox *f = (const vox *)p.field;
    const uint32_t B[3] = {p.Bx, p.By, p.Bz};
    const uint32_t bx1 = (p.Bx - 1) & 0xFFFFu, by1 = (p.By - 1) & 0xFFFFu, bz1 = (p.Bz - 1) & 0xFFFFu;
    const float invx = p.invx, invy = p.invy, invz = p.invz;

    // `it` is the reference's remaining-iterations counter (.cu:333-351): it drops by one per
    // completed step and the ray stops at it == 0, on leaving the field, or on w > 0; the
    // value at that moment is raydata._iterations.
    uint32_t it = p.iterations - 1;
    bool done = !have;

    for (;;) {
        bool live = !done && it > 0 && in_field(r, bx1, by1, bz1);
        if (!live) done = true;
        if (__ballot(live) == 0ull) break;              // wave early-out: every lane has terminated

        // ---- plan -------------------------------------------------------------------------
        int32_t s[3] = {0, 0, 0};
        if (live) {                                     // per-step displacement estimate from the current direction
            const float dot = fma_(r.dz, r.dz, fma_(r.dx, r.dx, r.dy * r.dy));
            const float ilen = 0x42000000p0f / dot;
            s[0] = round_to_i32((r.dx * invx) * ilen);
            s[1] = round_to_i32((r.dy * invy) * ilen);
            s[2] = round_to_i32((r.dz * invz) * ilen);
        }
        uint32_t pmin[3], pmax[3];
        int32_t smin[3], smax[3];
        pmin[0] = sgpr(__ockl_wfred_min_u32(live ? r.px : 0xFFFFFFFFu)); pmax[0] = sgpr(__ockl_wfred_max_u32(live ? r.px : 0u));
        pmin[1] = sgpr(__ockl_wfred_min_u32(live ? r.py : 0xFFFFFFFFu)); pmax[1] = sgpr(__ockl_wfred_max_u32(live ? r.py : 0u));
        pmin[2] = sgpr(__ockl_wfred_min_u32(live ? r.pz : 0xFFFFFFFFu)); pmax[2] = sgpr(__ockl_wfred_max_u32(live ? r.pz : 0u));
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            smin[d] = (int32_t)sgpr((uint32_t)__ockl_wfred_min_i32(live ? s[d] : 0x7FFFFFFF));
            smax[d] = (int32_t)sgpr((uint32_t)__ockl_wfred_max_i32(live ? s[d] : (int32_t)0x80000000));
        }
        Brick bk;
        int K = kSlabMaxK;
        uint32_t vol = plan_brick(pmin, pmax, smin, smax, K, B, bk);
        while (vol > cap_vox && K > 1) {
            K >>= 1;
            vol = plan_brick(pmin, pmax, smin, smax, K, B, bk);
        }

        if (vol > cap_vox || bk.ex < 2 || bk.ey < 2 || bk.ez < 2) {
            // ---- incoherent wave: gather from global memory for a few steps, then re-plan ----
            for (int k = 0; k < kSlabDirectSteps; ++k) {
                live = !done && it > 0 && in_field(r, bx1, by1, bz1);
                if (!live) done = true;
                if (__ballot(live) == 0ull) break;
                if (live) {
                    f4 c[8];
                    gather_global<FieldT>(f, r, p.By, p.Bz, c);
                    if (march_step(r, c, invx, invy, invz)) done = true; else --it;
                }
            }
            continue;
        }

        // ---- stage: brick -> LDS ----------------------------------------------------------
        {
            const uint32_t ez = bk.ez, eyz = bk.ey * bk.ez;
            // exact division by the wave-uniform extents via 2^32 reciprocals (v < 2^16)
            const uint32_t mz = ez > 1 ? (uint32_t)(0x100000000ull / ez) + 1u : 0u;
            const uint32_t my = bk.ey > 1 ? (uint32_t)(0x100000000ull / bk.ey) + 1u : 0u;
            const size_t gsy = p.Bz, gsx = (size_t)p.By * p.Bz;
            const vox *g0 = f + ((size_t)bk.lox * p.By + bk.loy) * p.Bz + bk.loz;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");    // previous round's LDS reads are done
            for (uint32_t v = lane; v < vol; v += 64) {
                const uint32_t t = ez > 1 ? __umulhi(v, mz) : v;           // v / ez
                const uint32_t lz = v - t * ez;
                const uint32_t lx = bk.ey > 1 ? __umulhi(t, my) : t;       // t / ey
                const uint32_t ly = t - lx * bk.ey;
                seg[v] = g0[(size_t)lx * gsx + (size_t)ly * gsy + lz];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            (void)eyz;
        }

        // ---- march: up to K steps out of LDS ------------------------------------------------
        {
            const uint32_t sy = bk.ez, sx = bk.ey * bk.ez;
            const uint32_t ex1 = bk.ex - 1, ey1 = bk.ey - 1, ez1 = bk.ez - 1;
            for (int k = 0; k < K; ++k) {
                live = !done && it > 0 && in_field(r, bx1, by1, bz1);
                if (!live) done = true;
                const uint32_t cx = (r.px >> 16) - bk.lox, cy = (r.py >> 16) - bk.loy, cz = (r.pz >> 16) - bk.loz;
                const bool inb = cx < ex1 && cy < ey1 && cz < ez1;   // cell and its +1 neighbours are staged
                const unsigned long long lv = __ballot(live);
                if (lv == 0ull) break;
                if (__ballot(live && !inb) != 0ull) break;           // somebody left the brick: re-plan
                if (live) {
                    f4 c[8];
                    gather_lds<FieldT>(seg, (cx * bk.ey + cy) * bk.ez + cz, sy, sx, c);
                    if (march_step(r, c, invx, invy, invz)) done = true; else --it;
                }
            }
        }
    }
    if (have) store_ray<DirT>(p, ray, r, it);
}

template <typename FieldT, typename DirT>
hipError_t launch_slab(MarchParams p, bool /*path*/, int /*num_cu*/, hipStream_t s)
{
    if (p.slab_lds_bytes == 0) p.slab_lds_bytes = 8192;
    const unsigned threads = 64 * kSlabWavesPerBlock;
    const unsigned grid = (p.nrays + threads - 1) / threads;
    const size_t lds = (size_t)p.slab_lds_bytes * kSlabWavesPerBlock;
    k_march_slab<FieldT, DirT><<<grid, threads, lds, s>>>(p);
    return hipGetLastError();
}

}  // namespace vrt
