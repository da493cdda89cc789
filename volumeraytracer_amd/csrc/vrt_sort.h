// vrt_sort.h -- ray -> wavefront assignment ("rays grouped by entry tile").
//
// The reference processes rays in caller order, one CUDA thread each
// (/root/reference/src/cuda_volume_raytracer.cu:397-414).  On CDNA4 a 64-lane wavefront gathers
// 8 voxels per lane per step, so which rays share a wavefront decides how many distinct
// cache lines / LDS slab voxels one step touches.  This pass builds a permutation that puts
// rays with the same dominant direction and neighbouring start voxels (3-D Morton order) next
// to each other; kernels process slot i -> ray order[i] and write results back to the ray's
// own index, so outputs are independent of the permutation.
#pragma once
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>

namespace vrt {

__device__ __forceinline__ uint64_t spread3(uint32_t v)   // 16 bits -> every third bit
{
    uint64_t x = v & 0xFFFFu;
    x = (x | (x << 32)) & 0x001F00000000FFFFull;
    x = (x | (x << 16)) & 0x001F0000FF0000FFull;
    x = (x | (x << 8))  & 0x100F00F00F00F00Full;
    x = (x | (x << 4))  & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2))  & 0x1249249249249249ull;
    return x;
}

template <typename DirT>
__global__ __launch_bounds__(256) void k_tile_keys(const uint32_t *pos, const DirT *dir, uint32_t n,
                                                   uint64_t *keys, uint32_t *vals)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t x = pos[3 * (size_t)i] >> 16, y = pos[3 * (size_t)i + 1] >> 16, z = pos[3 * (size_t)i + 2] >> 16;
    const float dx = (float)dir[3 * (size_t)i], dy = (float)dir[3 * (size_t)i + 1], dz = (float)dir[3 * (size_t)i + 2];
    const float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
    uint32_t dom;
    if (ax >= ay && ax >= az) dom = dx < 0 ? 1u : 0u;
    else if (ay >= az)        dom = dy < 0 ? 3u : 2u;
    else                      dom = dz < 0 ? 5u : 4u;
    // z is the memory-minor axis: give it the lowest Morton bit so lane neighbours are address neighbours
    keys[i] = ((uint64_t)dom << 48) | (spread3(x) << 2) | (spread3(y) << 1) | spread3(z);
    vals[i] = i;
}

// grow-only device buffer interface expected from the caller (see DevBuf in vrt_engine.hip)
template <typename Buf>
hipError_t build_tile_order(const uint32_t *d_pos, const void *d_dir, int dir_is_i16, uint32_t n, Buf &keys,
                            Buf &keys_alt, Buf &vals, Buf &vals_alt, Buf &tmp, hipStream_t s,
                            const uint32_t **order_out)
{
    hipError_t e;
    if ((e = keys.reserve((size_t)n * 8)) != hipSuccess) return e;
    if ((e = keys_alt.reserve((size_t)n * 8)) != hipSuccess) return e;
    if ((e = vals.reserve((size_t)n * 4)) != hipSuccess) return e;
    if ((e = vals_alt.reserve((size_t)n * 4)) != hipSuccess) return e;
    const unsigned grid = (n + 255) / 256;
    if (dir_is_i16) k_tile_keys<int16_t><<<grid, 256, 0, s>>>(d_pos, (const int16_t *)d_dir, n, (uint64_t *)keys.p, (uint32_t *)vals.p);
    else            k_tile_keys<float><<<grid, 256, 0, s>>>(d_pos, (const float *)d_dir, n, (uint64_t *)keys.p, (uint32_t *)vals.p);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    size_t tmp_bytes = 0;
    e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, (const uint64_t *)keys.p, (uint64_t *)keys_alt.p,
                                  (const uint32_t *)vals.p, (uint32_t *)vals_alt.p, n, 0, 51, s);
    if (e != hipSuccess) return e;
    if ((e = tmp.reserve(tmp_bytes ? tmp_bytes : 16)) != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(tmp.p, tmp_bytes, (const uint64_t *)keys.p, (uint64_t *)keys_alt.p,
                                  (const uint32_t *)vals.p, (uint32_t *)vals_alt.p, n, 0, 51, s);
    if (e != hipSuccess) return e;
    *order_out = (const uint32_t *)vals_alt.p;
    return hipSuccess;
}

}  // namespace vrt
