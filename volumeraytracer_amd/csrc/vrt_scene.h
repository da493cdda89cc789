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
#include <type_traits>

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
// Two-phase form for the tiled build: raw(i) is the element as it lies in memory (issued a plane ahead), cook(raw, table)
// turns it into the value -- for a LUT source a look-up in the workgroup's LDS copy of the 256-entry table, so that no
// global load ever depends on another one.
template <typename T> struct SrcArray {
    const T *a;
    using raw_t = T;
    static constexpr bool kLut = false;
    __device__ __forceinline__ T operator()(size_t i) const { return a[i]; }
    __device__ __forceinline__ raw_t raw(size_t i) const { return a[i]; }
    __device__ __forceinline__ T cook(raw_t r, const uint32_t *) const { return r; }
    __device__ __forceinline__ uint32_t table_bits(uint32_t) const { return 0u; }
};
template <typename T> struct SrcLut {
    const uint8_t *d;
    const T *lut;
    using raw_t = uint8_t;
    static constexpr bool kLut = true;
    __device__ __forceinline__ T operator()(size_t i) const { return lut[d[i]]; }
    __device__ __forceinline__ raw_t raw(size_t i) const { return d[i]; }
    __device__ __forceinline__ T cook(raw_t r, const uint32_t *table) const
    {
        const uint32_t b = table[r];
        T v;
        __builtin_memcpy(&v, &b, 4);
        return v;
    }
    __device__ __forceinline__ uint32_t table_bits(uint32_t k) const
    {
        const T v = lut[k];
        uint32_t b;
        __builtin_memcpy(&b, &v, 4);
        return b;
    }
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

// ---- LDS-tiled scene build (round 3) ---------------------------------------------------------------------------
// The per-voxel kernel above reads its 27 taps straight from global memory: 27 poorly coalesced loads per voxel (or 27
// byte loads + table look-ups for a LUT scene) kept it at 0.37 of the HBM peak at 512^3 and 0.12 at 1024^3 + LUT
// (profiles/r02_scene_build_512.txt).  Here a workgroup owns a column of kTY x kTZ output voxels and walks it along x
// (the classic 2.5-D stencil): every input plane is staged ONCE per workgroup into LDS as a (kTY + 2) x (kTZ + 2) tile
// of log values -- rows of 66 consecutive z, i.e. coalesced 264-byte (66-byte for the 8-bit volume) runs; the LUT is applied
// while staging, one look-up per staged voxel instead of 27 per output -- each thread then reads its 3 x 3 neighbourhood of
// the new plane from LDS (9 reads per output instead of 27 loads) and keeps the two previous planes in registers.  The
// plane buffer is double-buffered: one barrier per plane, and the next plane's global loads are in flight during the
// arithmetic of the current one.  Results are written as 1 KiB (512-byte int16) runs along z, one row per wavefront.
//
// Tile shape (rows x z, 256 threads unless noted; MI355X, 512^3 float / 512^3 fixed-point / 1024^3 LUT, ms): 8 x 32 0.735 / 0.556 /
// 4.6; 8 x 64 (512 threads) 0.69 / 0.51 / 4.7; 8 x 128 (1024) 0.74 / 0.60 / 4.85; 4 x 128 (512) 0.69 / 0.556 / 4.75; 6 x 64 (384)
// 0.89 / 0.64 / 5.5; **4 x 64 0.675 / 0.487 / 4.65** -- the field is dense, its rows are not line-aligned (see the row shift below), so
// the longer a wavefront's run the fewer lines it shares with another workgroup; that beats the larger halo (1.5 x 1.03 instead of
// 1.25 x 1.06).  16 or 64 planes per workgroup instead of 32: slower (0.75 / 0.55 / 5.2, 0.755 / 0.54 / 4.8).
//
// The arithmetic is the per-voxel kernel's, term for term: three independent fma chains in ascending stamp index
// (i major, k minor), zero coefficients skipped (image_util.cpp:401-411), IEEE division by 812 * 256; the integer family
// wraps in int32 and rounds with divRoundClosest.  tests/test_gpu_scene.py holds both kernels to the reference's bits.
//
// The field scan the engine needs for its proven rounds and the prescale flag (engine_field_absmax: max |channel|, max
// squared gradient length, smallest non-zero magnitude) is fused in: each workgroup reduces its voxels and writes five
// words; k_absmax_reduce folds them.  (Atomics on five addresses from 16 384 workgroups would serialise, see k_ray_bounds.)
#ifndef VRT_SCENE_TY
#define VRT_SCENE_TY 4
#endif
#ifndef VRT_SCENE_TZ
#define VRT_SCENE_TZ 64
#endif
#ifndef VRT_SCENE_XS
#define VRT_SCENE_XS 32
#endif
constexpr int kTY = VRT_SCENE_TY, kTZ = VRT_SCENE_TZ, kXS = VRT_SCENE_XS;   // output tile of a workgroup: kXS x kTY x kTZ voxels (x walked)
constexpr int kZS = 3;                                  // largest row shift (voxels) of a tile: runs aligned to four voxels

template <typename LogSrc, typename TrSrc, typename VoxT>
__global__ __launch_bounds__(kTY * kTZ) void k_scene_build_tiled(const LogSrc lg, const TrSrc transl, uint32_t Nx, uint32_t Ny, uint32_t Nz,
                                                                  VoxT *__restrict__ field, SceneErr *err,
                                                                  uint32_t *__restrict__ transl_cropped /* or nullptr */,
                                                                  uint32_t *__restrict__ absmax_partial /* [workgroups][5] */,
                                                                  uint32_t zshift /* 0 or kZS: see below */)
{
    constexpr bool kFloat = sizeof(VoxT) == 16;
    using T = std::conditional_t<kFloat, float, uint32_t>;   // log values: float, or int32 held as uint32 (the sums wrap)
    constexpr int PY = kTY + 2, PZ = kTZ + 2 + kZS, kThreads = kTY * kTZ;
    static_assert(PY * PZ <= 2 * kThreads && kThreads <= 1024 && kThreads % 64 == 0, "two staging slots per thread must cover the tile");
    __shared__ T plane[2][PY][PZ];
    __shared__ uint32_t tab_lg[LogSrc::kLut ? 256 : 1], tab_tr[TrSrc::kLut ? 256 : 1];   // LUT scenes: the two tables, one look-up per staged voxel
    if constexpr (LogSrc::kLut) { for (uint32_t k = threadIdx.x; k < 256u; k += kThreads) tab_lg[k] = lg.table_bits(k); }
    if constexpr (TrSrc::kLut) { for (uint32_t k = threadIdx.x; k < 256u; k += kThreads) tab_tr[k] = transl.table_bits(k); }
    const uint32_t Ox = Nx - 2, Oy = Ny - 2, Oz = Nz - 2;
    const uint32_t tid = threadIdx.x, tz = tid % kTZ, ty = tid / kTZ;
    const uint32_t z0 = blockIdx.x * kTZ, y0 = blockIdx.y * kTY, x0 = blockIdx.z * kXS;
    // Row shift: the field is dense, its rows of Oz voxels are in general not a multiple of a memory sector long, so a run of
    // kTZ voxels starting at a multiple of kTZ straddles sectors it shares with the neighbouring workgroup's run (measured:
    // 3.7 -> 4.0 TB/s for int16 voxels when the rows are 32-byte aligned, 4.1 -> 4.4 for float voxels at 64 bytes).
    // With zshift = kZS = 3 the runs of row y start at z = kTZ * i - ((y * Oz) & 3), i.e. at a multiple of four voxels of the
    // field's linear index -- the same shift in every plane, because the engine only asks for it when Oy * Oz is a multiple of
    // four; the LDS tile is kZS columns wider for it.
    const uint32_t y = y0 + ty, zsh = (y * Oz) & zshift;
    const uint32_t z = z0 + tz - zsh;                     // wraps below zero for the first lanes of a shifted row's first run: idle
    const bool active = y < Oy && z < Oz;
    const uint32_t zcol = tz + zshift - zsh;              // this thread's first column of the tile (the tile starts at z0 - zshift)
    const uint32_t pw = (uint32_t)kTZ + 2u + zshift;      // staged columns
    const uint32_t nx = Ox - x0 < (uint32_t)kXS ? Ox - x0 : (uint32_t)kXS;     // output planes of this workgroup
    const uint32_t nplanes = nx + 2;

    // staging slots of this thread: the same two tile positions for every plane
    bool s_ok[2];
    uint32_t s_ly[2], s_lz[2];
    size_t s_off[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const uint32_t sidx = tid + kThreads * q;
        s_ly[q] = sidx / pw;
        s_lz[q] = sidx - s_ly[q] * pw;
        const uint32_t zin = z0 + s_lz[q] - zshift;       // wraps below zero left of the volume
        s_ok[q] = sidx < (uint32_t)PY * pw && y0 + s_ly[q] < Ny && zin < Nz;
        s_off[q] = (size_t)(y0 + s_ly[q]) * Nz + zin;
    }
    // global loads are issued a plane ahead (gload at the top of an iteration) and consumed at its bottom (lstore) or in the
    // next iteration (the translucency of the output voxel): their latency lies under the arithmetic of the current plane.
    // (Four planes ahead, with the loop unrolled by four, was measured too: 0.83 / 0.66 / 5.79 ms against 0.79 / 0.63 / 5.81
    // for 512^3 float / 512^3 fixed-point / 1024^3 LUT at the time: no gain.  The compiler guards these conditional loads with
    // vmcnt(0) -- it cannot count operations in flight across the branches -- so a wavefront waits for the plane it has just
    // requested (SQ_WAIT_ANY = 69 % of the wave-cycles of the 1024^3 LUT build); with every load and LDS write made unconditional
    // (clamped addresses, dummy slots) the waits become vmcnt(3) / (2) / (1) and the time does not move -- 4.36-4.72 against 4.35-4.66,
    // one or two planes ahead alike: eight wavefronts per SIMD already cover the wait.  Dealing the tiles of an 8-bit volume
    // to the XCDs in quads of z-neighbours, so that the four tiles sharing a 128-byte line share an L2, cuts the fetched bytes
    // (raw FETCH_SIZE) from 2.85 to 1.08 GB and makes the build slower, 5.0-5.2 ms.  Both kept as a patch: tools/r03_batches/scene_unconditional_loads.patch.)
    typename LogSrc::raw_t stage[2] = {};
    auto gload = [&](uint32_t xin) {
#pragma unroll
        for (int q = 0; q < 2; ++q) if (s_ok[q]) stage[q] = lg.raw((size_t)xin * Ny * Nz + s_off[q]);
    };
    auto lstore = [&](uint32_t b) {
#pragma unroll
        for (int q = 0; q < 2; ++q) if (s_ok[q]) plane[b][s_ly[q]][s_lz[q]] = (T)lg.cook(stage[q], tab_lg);
    };
    typename TrSrc::raw_t tr_raw = {}, tr_next = {};
    const size_t tr_off = (size_t)(y + 1) * Nz + (z + 1);                // crop_matrix, image_util.cpp:300-319

    // The three planes a thread works on live in three register sets.  Fixed-point family: their roles rotate with the plane
    // index (the loop is unrolled by three, every index a compile-time constant, no register is ever copied): 0.63 -> 0.58 ms at
    // 512^3.  Float family: the sets are shifted by 18 register moves per plane instead -- that kernel is bound by its memory
    // traffic (3.2 GB in 0.79 ms = 0.77 of the copy rate measured on these boxes), not by its instructions: the unrolled form
    // measured slower (0.85 ms), and replacing the three IEEE divisions by a reciprocal multiplication + fma correction (exact for
    // every float inside an exponent window: proven exhaustively on the device, then removed again) changed nothing (0.792 vs 0.783).
#ifdef VRT_SCENE_ROTATE_ALL
    constexpr bool kRotate = true;
#else
    constexpr bool kRotate = kFloat;
#endif
#ifdef VRT_SCENE_NO_PARTIALS
    constexpr bool kPartials = false;
#else
    constexpr bool kPartials = !kFloat;
#endif
    uint32_t part[3][5] = {};
    T p[3][3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) p[i][j][k] = T(0);
    uint32_t m[5] = {0u, 0u, 0u, 0u, 0u};                 // the field scan: see k_field_absmax (vrt_engine.hip)
    bool ovf = false;

    auto step = [&](uint32_t t, auto ph) {                // input plane t goes into set P; set (P + 1 + i) % 3 holds plane t - 2 + i
        constexpr int P = decltype(ph)::value;            // (kRotate: always 2, and plane t - 2 + i is in set i)
        if (t + 1 < nplanes) gload(x0 + t + 1);           // in flight during the arithmetic below
        if (active && t >= 1 && t - 1 < nx) tr_next = transl.raw((size_t)(x0 + t) * Ny * Nz + tr_off);   // of output plane x0 + t - 1, needed next iteration
        if constexpr (kPartials) {
            // Fixed-point family: the sums wrap (two's complement), so they form a ring and may be regrouped freely.  Each plane
            // contributes to the three sums of its three output planes only through five numbers, computed once when it arrives:
            // its stamp-weighted 2-D sum (x derivative = that of the plane ahead - that of the plane behind) and, for y and z,
            // the outer pair and the centre of its three differences across the rows / the columns (weighted 14 | 47 or 47 | 162
            // by the plane's position).  36 integer operations per voxel instead of 108, 15 live registers instead of 27
            // (56 VGPRs, 8 wavefronts per SIMD).  Measured against the term-by-term form (-DVRT_SCENE_NO_PARTIALS): 1024^3 LUT
            // 4.81-4.96 -> 4.35-4.66 ms; 512^3 from arrays unchanged within the noise (0.54-0.57 vs 0.58-0.63: that build moves
            // 16 B per voxel and sits at the memory rate).  (Regrouping inside the voxel only -- 9 multiplications instead of 54,
            // every plane still differenced three times -- was the earlier form for LUT scenes: 4.95 ms.)
            uint32_t c[3][3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) c[j][k] = plane[t & 1u][ty + j][zcol + k];
            part[P][0] = 14u * (c[0][0] + c[0][2] + c[2][0] + c[2][2]) + 47u * (c[0][1] + c[1][0] + c[1][2] + c[2][1]) + 162u * c[1][1];
            part[P][1] = (c[2][0] - c[0][0]) + (c[2][2] - c[0][2]);      // y differences, outer columns
            part[P][2] = c[2][1] - c[0][1];                              //                centre column
            part[P][3] = (c[0][2] - c[0][0]) + (c[2][2] - c[2][0]);      // z differences, outer rows
            part[P][4] = c[1][2] - c[1][0];                              //                centre row
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if constexpr (kRotate) { p[0][j][k] = p[1][j][k]; p[1][j][k] = p[2][j][k]; }
                    p[P][j][k] = plane[t & 1u][ty + j][zcol + k];
                }
        }
        if (t >= 2 && active) {
            const uint32_t x = x0 + t - 2;
            const size_t idx = ((size_t)x * Oy + y) * Oz + z;
            const uint32_t tr = transl.cook(tr_raw, tab_tr);
            VoxT o;
            float a[3];
            if constexpr (kFloat) {
                float sum[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const float v = p[(P + 1 + i) % 3][j][k];
                            if (i != 1) sum[0] = fma_((float)((i - 1) * stamp_m(j, k)), v, sum[0]);
                            if (j != 1) sum[1] = fma_((float)((j - 1) * stamp_m(i, k)), v, sum[1]);
                            if (k != 1) sum[2] = fma_((float)((k - 1) * stamp_m(i, j)), v, sum[2]);
                        }
                const float weight = 812.0f * 256.0f;
                o.x = sum[0] / weight; o.y = sum[1] / weight; o.z = sum[2] / weight;          // "sum /= weight", image_util.cpp:283-285
                o.w = (float)((int32_t)(0x7FFFFFFFu - tr) / 0x10000);                         // cuda_volume_raytracer.cu:656-659
                a[0] = o.x; a[1] = o.y; a[2] = o.z;
            } else {
                uint32_t sum[3];                                                              // int32 arithmetic wraps in the reference binary
                if constexpr (kPartials) {
                    const uint32_t *pa = part[(P + 1) % 3], *pb = part[(P + 2) % 3], *pc = part[P];   // planes x, x + 1, x + 2
                    sum[0] = pc[0] - pa[0];
                    sum[1] = 14u * (pa[1] + pc[1]) + 47u * (pa[2] + pc[2] + pb[1]) + 162u * pb[2];
                    sum[2] = 14u * (pa[3] + pc[3]) + 47u * (pa[4] + pc[4] + pb[3]) + 162u * pb[4];
                } else {                                                                      // term by term, as k_scene_build
                    sum[0] = sum[1] = sum[2] = 0u;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j)
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                const uint32_t v = p[(P + 1 + i) % 3][j][k];
                                if (i != 1) sum[0] += (uint32_t)((i - 1) * stamp_m(j, k)) * v;
                                if (j != 1) sum[1] += (uint32_t)((j - 1) * stamp_m(i, k)) * v;
                                if (k != 1) sum[2] += (uint32_t)((k - 1) * stamp_m(i, j)) * v;
                            }
                }
                const int32_t weight = 812 * 0x100;
                int16_t q16[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int32_t sv = (int32_t)sum[c];
                    const int32_t q = sv < 0 ? (sv - weight / 2) / weight : (sv + weight / 2) / weight;   // divRoundClosest, image_util.h:34-38
                    q16[c] = (int16_t)q;
                    ovf |= (int32_t)q16[c] != q;
                    a[c] = (float)q16[c];
                }
                o.x = q16[0]; o.y = q16[1]; o.z = q16[2];
                o.w = (int16_t)((int32_t)(0x7FFFFFFFu - tr) / 0x10000);
            }
            field[idx] = o;
            if (transl_cropped) transl_cropped[idx] = tr;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const uint32_t b = __float_as_uint(a[c]) & 0x7FFFFFFFu;
                m[c] = b > m[c] ? b : m[c];
                const uint32_t cc = ~(b - 1u);
                m[4] = cc > m[4] ? cc : m[4];
            }
            const uint32_t q2 = __float_as_uint(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) & 0x7FFFFFFFu;
            m[3] = q2 > m[3] ? q2 : m[3];
        }
        if (t + 1 < nplanes) lstore((t + 1) & 1u);
        tr_raw = tr_next;
        __syncthreads();
    };

    __syncthreads();                                      // the tables
    gload(x0);
    lstore(0);
    __syncthreads();
    if constexpr (kRotate) {
        for (uint32_t t = 0; t < nplanes; ++t) step(t, std::integral_constant<int, 2>{});
    } else {
        for (uint32_t t = 0; t < nplanes; t += 3) {       // nplanes is uniform over the workgroup: so are the barriers inside
            step(t, std::integral_constant<int, 0>{});
            if (t + 1 < nplanes) step(t + 1, std::integral_constant<int, 1>{});
            if (t + 2 < nplanes) step(t + 2, std::integral_constant<int, 2>{});
        }
    }
    if (ovf) atomicOr(&err->diff_overflow, 1u);
    // workgroup reduction of the field scan -> five words per workgroup
    __shared__ uint32_t s_m[kThreads / 64][5];
#pragma unroll
    for (int d = 0; d < 5; ++d) {
        for (int o = 32; o > 0; o >>= 1) { const uint32_t tt = (uint32_t)__shfl_xor((int)m[d], o); m[d] = tt > m[d] ? tt : m[d]; }
        if ((tid & 63u) == 0) s_m[tid >> 6][d] = m[d];
    }
    __syncthreads();
    if (tid < 5 && absmax_partial) {
        uint32_t v = s_m[0][tid];
        for (int w = 1; w < kThreads / 64; ++w) v = s_m[w][tid] > v ? s_m[w][tid] : v;
        const size_t wg = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        absmax_partial[wg * 5 + tid] = v;
    }
}

// folds the workgroups' five words: a few workgroups, one atomic per word each (out[] zeroed by the caller)
__global__ __launch_bounds__(256) void k_absmax_reduce(const uint32_t *partial, size_t nwg, uint32_t *out /* [5] */)
{
    __shared__ uint32_t s_m[4][5];
    uint32_t m[5] = {0u, 0u, 0u, 0u, 0u};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwg; i += (size_t)gridDim.x * blockDim.x)
#pragma unroll
        for (int d = 0; d < 5; ++d) { const uint32_t v = partial[i * 5 + d]; m[d] = v > m[d] ? v : m[d]; }
#pragma unroll
    for (int d = 0; d < 5; ++d) {
        for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)m[d], o); m[d] = t > m[d] ? t : m[d]; }
        if ((threadIdx.x & 63u) == 0) s_m[threadIdx.x >> 6][d] = m[d];
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        uint32_t v = s_m[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) v = s_m[w][threadIdx.x] > v ? s_m[w][threadIdx.x] : v;
        atomicMax(&out[threadIdx.x], v);
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
