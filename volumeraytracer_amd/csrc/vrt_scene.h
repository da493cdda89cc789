// vrt_scene.h -- scene front end on the device.
//
// Replaces, bit for bit, the per-voxel and per-ray work of the reference's scene layer
//   RaytraceScene ctor: crop, derivative stencil, (then TraceRaysCu ctor)   image_util.cpp:501-643
//     convolution<T,V,W>::operator()                                         image_util.cpp:261-297
//     standart_3d_stamp                                                      image_util.cpp:421-425
//   RaytraceScene::trace_rays prologue: range check, ior interpolation,
//     direction normalisation, coordinate shifts                            image_util.cpp:645-772
//     interpolator<T>                                                        image_util.h:348-431
// log(ior) itself stays on the host (libm, the very function the reference calls) so that the
// logarithm's last bit matches; everything that scales with 18 taps x 3 axes x voxels runs here
// and writes the engine's interleaved {dx,dy,dz,w} field directly -- no intermediate diff arrays.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vrt_march.h"

namespace vrt {

// derivative stamp: value(i,j,k) = s(c[axis]) * m(other two), s = {-1,0,1},
// m = [[14,47,14],[47,162,47],[14,47,14]]; weight = 812 * 0x100 (image_util.cpp:421-441)
__device__ __forceinline__ int stamp_m(int a, int b)
{
    const int ca = a == 1 ? 1 : 0, cb = b == 1 ? 1 : 0;
    return ca && cb ? 162 : (ca || cb ? 47 : 14);
}

// Voxel sources: a plain array, or an 8-bit density volume seen through a 256-entry table
// (BASELINE config 5 / SURVEY row f-3, reading A: the table is expanded on the fly, so the stencil and
// the ior interpolation see exactly the arrays the reference would be given).
template <typename T> struct SrcArray {
    const T *a;
    __device__ __forceinline__ T operator()(size_t i) const { return a[i]; }
};
template <typename T> struct SrcLut {
    const uint8_t *d;
    const T *lut;
    __device__ __forceinline__ T operator()(size_t i) const { return lut[d[i]]; }
};

struct SceneErr {            // device -> host error words
    unsigned int diff_overflow;       // "differention overflow" (image_util.cpp:291-293)
    unsigned int first_bad_ray;       // 0xFFFFFFFF = none; else lowest ray index failing the range check
    unsigned int normalize_overflow;  // "Normalize length failed" (image_util.cpp:705-708)
};

// One thread per output voxel; terms are accumulated in the reference's order: ascending linear
// stamp index (i major, k minor) with zero coefficients skipped (image_util.cpp:401-411).
template <typename LogSrc, typename TrSrc, typename VoxT>
__global__ __launch_bounds__(256) void k_scene_build(const LogSrc lg, const TrSrc transl,
                                                     uint32_t Nx, uint32_t Ny, uint32_t Nz, VoxT *__restrict__ field,
                                                     SceneErr *err, uint32_t *__restrict__ transl_cropped /* or nullptr */)
{
    const uint32_t Ox = Nx - 2, Oy = Ny - 2, Oz = Nz - 2;
    const size_t n = (size_t)Ox * Oy * Oz;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const uint32_t z = (uint32_t)(idx % Oz);
    const size_t t = idx / Oz;
    const uint32_t y = (uint32_t)(t % Oy), x = (uint32_t)(t / Oy);
    const size_t p0 = ((size_t)x * Ny + y) * Nz + z;
    const size_t sy = Nz, sx = (size_t)Ny * Nz;

    if constexpr (sizeof(VoxT) == 16) {                   // float family: RaytraceScene<float,float,float>
        float sum[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float v = (float)lg(p0 + i * sx + j * sy + k);
                    if (i != 1) sum[0] = fma_((float)((i - 1) * stamp_m(j, k)), v, sum[0]);
                    if (j != 1) sum[1] = fma_((float)((j - 1) * stamp_m(i, k)), v, sum[1]);
                    if (k != 1) sum[2] = fma_((float)((k - 1) * stamp_m(i, j)), v, sum[2]);
                }
        const float weight = 812.0f * 256.0f;
        const uint32_t tr = transl(((size_t)(x + 1) * Ny + (y + 1)) * Nz + (z + 1));      // crop_matrix, :300-319
        VoxT o;
        o.x = sum[0] / weight; o.y = sum[1] / weight; o.z = sum[2] / weight;              // "sum /= weight", :283-285
        o.w = (float)((int32_t)(0x7FFFFFFFu - tr) / 0x10000);                             // cuda_volume_raytracer.cu:656-659
        field[idx] = o;
        if (transl_cropped) transl_cropped[idx] = tr;
    } else {                                              // integer family: RaytraceScene<ior_t,iorlog_t,diff_t>
        uint32_t sum[3] = {0u, 0u, 0u};                   // int32 arithmetic wraps in the reference binary
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const uint32_t v = (uint32_t)lg(p0 + i * sx + j * sy + k);
                    if (i != 1) sum[0] += (uint32_t)((i - 1) * stamp_m(j, k)) * v;
                    if (j != 1) sum[1] += (uint32_t)((j - 1) * stamp_m(i, k)) * v;
                    if (k != 1) sum[2] += (uint32_t)((k - 1) * stamp_m(i, j)) * v;
                }
        const int32_t weight = 812 * 0x100;
        int16_t o[3];
        bool ovf = false;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int32_t s = (int32_t)sum[a];
            const int32_t q = s < 0 ? (s - weight / 2) / weight : (s + weight / 2) / weight;   // divRoundClosest, image_util.h:34-38
            o[a] = (int16_t)q;
            ovf |= (int32_t)o[a] != q;
        }
        if (ovf) atomicOr(&err->diff_overflow, 1u);
        const uint32_t tr = transl(((size_t)(x + 1) * Ny + (y + 1)) * Nz + (z + 1));
        VoxT v;
        v.x = o[0]; v.y = o[1]; v.z = o[2];
        v.w = (int16_t)((int32_t)(0x7FFFFFFFu - tr) / 0x10000);
        field[idx] = v;
        if (transl_cropped) transl_cropped[idx] = tr;
    }
}

// ---- ray prologue (image_util.cpp:673-722) ---------------------------------------------------

// interpolator<float>::operator() (image_util.h:390-431)
template <typename Src>
__device__ __forceinline__ float interp_ior_f(const Src ior, uint32_t Ny, uint32_t Nz, const uint32_t p[3])
{
    const size_t base = ((size_t)(p[0] >> 16) * Ny + (p[1] >> 16)) * Nz + (p[2] >> 16);
    const size_t o2 = Nz, o4 = (size_t)Ny * Nz;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = ior(base + ((k & 1) ? 1 : 0) + ((k & 2) ? o2 : 0) + ((k & 4) ? o4 : 0));
    int cnt = 8;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const uint32_t mr = p[d] & 0xFFFFu, ml = 0x10000u - mr;
        const float fl = (float)ml, fr = (float)mr;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < cnt / 2) v[i] = fma_(v[i], fl, v[i + cnt / 2] * fr);
        cnt /= 2;
    }
    return v[0] * (1.0f / 0x1000000000000p0f);
}

// interpolator<ior_t>::operator() (image_util.h:348-388)
template <typename Src>
__device__ __forceinline__ uint32_t interp_ior_u(const Src ior, uint32_t Ny, uint32_t Nz, const uint32_t p[3])
{
    const size_t base = ((size_t)(p[0] >> 16) * Ny + (p[1] >> 16)) * Nz + (p[2] >> 16);
    const size_t o2 = Nz, o4 = (size_t)Ny * Nz;
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = ior(base + ((k & 1) ? 1 : 0) + ((k & 2) ? o2 : 0) + ((k & 4) ? o4 : 0));
    int cnt = 8;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const uint64_t mr = p[d] & 0xFFFFu, ml = 0x10000u - mr;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < cnt / 2) v[i] = (uint32_t)(((uint64_t)v[i] * ml + (uint64_t)v[i + cnt / 2] * mr + 0x8000u) / 0x10000u);
        cnt /= 2;
    }
    return v[0];
}

template <typename IorSrc, typename DirT>
__global__ __launch_bounds__(256) void k_ray_prologue(const IorSrc ior, uint32_t Nx, uint32_t Ny, uint32_t Nz,
                                                      const uint32_t *pos_in, const DirT *dir_in,
                                                      uint32_t *pos_out, DirT *dir_out,   // may alias the inputs: each thread reads its ray first
                                                      uint32_t n, SceneErr *err)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t p[3] = {pos_in[3 * (size_t)i], pos_in[3 * (size_t)i + 1], pos_in[3 * (size_t)i + 2]};
    const uint32_t N[3] = {Nx, Ny, Nz};
    bool bad = false;
#pragma unroll
    for (int d = 0; d < 3; ++d)
        bad |= (uint64_t)p[d] < 0x10000ull || (uint64_t)p[d] + 1ull >= (uint64_t)N[d] * 0x10000ull;   // image_util.cpp:690
    if (bad) {
        atomicMin(&err->first_bad_ray, i);
#pragma unroll
        for (int d = 0; d < 3; ++d) { pos_out[3 * (size_t)i + d] = 0xFFFF0000u; dir_out[3 * (size_t)i + d] = dir_in[3 * (size_t)i + d]; }
        return;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) p[d] -= 0x8000u;
    float sf = 0.0f;
    uint32_t su = 0u;
    if constexpr (sizeof(DirT) == 4) sf = interp_ior_f(ior, Ny, Nz, p); else su = interp_ior_u(ior, Ny, Nz, p);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if constexpr (sizeof(DirT) == 4) {
            dir_out[3 * (size_t)i + d] = dir_in[3 * (size_t)i + d] * sf;                    // "*iter *= interpolated"
        } else {
            const int64_t nn = (int64_t)dir_in[3 * (size_t)i + d] * (int64_t)su, dd = 0x10000;
            const int64_t t = nn < 0 ? (nn - dd / 2) / dd : (nn + dd / 2) / dd;             // divRoundClosest
            if (t > 32767 || t < -32768) atomicOr(&err->normalize_overflow, 1u);
            dir_out[3 * (size_t)i + d] = (DirT)t;
        }
        pos_out[3 * (size_t)i + d] = p[d] - 0x8000u;
    }
}

}  // namespace vrt
