// vrt_sort.h -- ray -> wavefront assignment ("rays grouped by entry tile").
//
// The reference processes rays in caller order, one CUDA thread each
// (/root/reference/src/cuda_volume_raytracer.cu:397-414).  On CDNA4 a 64-lane wavefront gathers
// 8 voxels per lane per step, so which rays share a wavefront decides how many distinct cache
// lines one gather touches (direct kernel) and how small the voxel brick is that the wavefront
// has to stage in LDS (slab kernel).  This pass builds a permutation such that 64 consecutive
// slots hold rays that (a) share the dominant direction, (b) start inside one square tile of
// the plane orthogonal to it, the tile edge W chosen from the measured ray density so that a
// tile holds about 64 rays, and (c) are ordered row-major inside the tile with the
// memory-minor axis fastest, so the four lanes of a load quad fall into the same 128-byte line.
// Tiles are visited in serpentine order so that a wavefront straddling two tiles still has
// adjacent ones.  Kernels process slot i -> ray order[i] and write results back to the ray's
// own index, so outputs are independent of the permutation.
#pragma once
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "vrt_march.h"

namespace vrt {

struct RayBounds {          // filled on the device, consumed on the device: no host round trip
    uint32_t lo[3], hi[3];  // min / max start position per axis (16.16)
    unsigned long long work_sum;   // k_wave_work: sum of the wavefronts' work estimates (coarse steps); read by the time-sliced march
};

// ---- round 3: the same bounds without an init kernel, without atomics and with the call's memsets folded in --------------
// Every workgroup writes its own six words (`partial[block]`); whoever needs the bounds next (k_tile_hist, k_bounds_finish)
// folds the at most kBoundsBlocks partials itself -- 6 x 256 words out of L2, cheaper than 6 contended atomics per workgroup
// and one launch less (k_bounds_init).  The same kernel zeroes what the rest of the call needs zeroed: the tile bins and, for the
// engine, the scheduler's counters / queue ring / clock words (ZeroList) -- each of those was a hipMemsetAsync of its own, five
// small launches per call, as many as the ray order has kernels.
constexpr uint32_t kBoundsBlocks = 256;
struct ZeroList { uint32_t *p[4]; uint32_t words[4]; };
struct BoundsPartial { uint32_t lo[3], hi[3]; };

__global__ __launch_bounds__(256) void k_ray_bounds_partial(const uint32_t *pos, uint32_t n, BoundsPartial *partial, const ZeroList z)
{
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x, gsize = gridDim.x * blockDim.x;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        for (uint32_t i = gtid; i < z.words[k]; i += gsize) z.p[k][i] = 0u;
    __shared__ uint32_t s_lo[4][3], s_hi[4][3];
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (uint32_t i = gtid; i < n; i += gsize) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const uint32_t v = pos[3 * (size_t)i + d];
            lo[d] = v < lo[d] ? v : lo[d];
            hi[d] = v > hi[d] ? v : hi[d];
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t a = __shfl_xor(lo[d], o), c = __shfl_xor(hi[d], o);
            lo[d] = a < lo[d] ? a : lo[d];
            hi[d] = c > hi[d] ? c : hi[d];
        }
        if ((threadIdx.x & 63u) == 0) { s_lo[threadIdx.x >> 6][d] = lo[d]; s_hi[threadIdx.x >> 6][d] = hi[d]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int d = threadIdx.x;
        uint32_t l = s_lo[0][d], h = s_hi[0][d];
        for (int w = 1; w < 4; ++w) { l = s_lo[w][d] < l ? s_lo[w][d] : l; h = s_hi[w][d] > h ? s_hi[w][d] : h; }
        partial[blockIdx.x].lo[d] = l;
        partial[blockIdx.x].hi[d] = h;
    }
}

// fold the partials (one workgroup of 256 threads, nparts <= kBoundsBlocks) into `out` (LDS or global), work_sum = 0
__device__ __forceinline__ void fold_bounds(const BoundsPartial *partial, uint32_t nparts, RayBounds *out, uint32_t (*s_red)[6])
{
    const uint32_t t = threadIdx.x;
    uint32_t v[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    if (t < nparts) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { v[d] = partial[t].lo[d]; v[3 + d] = partial[t].hi[d]; }
    }
#pragma unroll
    for (int d = 0; d < 6; ++d) {
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t a = __shfl_xor(v[d], o);
            v[d] = d < 3 ? (a < v[d] ? a : v[d]) : (a > v[d] ? a : v[d]);
        }
        if ((t & 63u) == 0) s_red[t >> 6][d] = v[d];
    }
    __syncthreads();
    if (t < 6) {
        uint32_t r = s_red[0][t];
        for (int w = 1; w < 4; ++w) r = t < 3 ? (s_red[w][t] < r ? s_red[w][t] : r) : (s_red[w][t] > r ? s_red[w][t] : r);
        if (t < 3) out->lo[t] = r; else out->hi[t - 3] = r;
    }
    if (t == 6) out->work_sum = 0ull;
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_bounds_finish(const BoundsPartial *partial, uint32_t nparts, RayBounds *rb)
{
    __shared__ uint32_t s_red[4][6];
    fold_bounds(partial, nparts, rb, s_red);
}

// 32-bit key, most significant first:
//   work bucket:3 | dom:3 | depth bin:4 | tile u:8 | tile v (serpentine):8 | row inside tile:3 | v inside tile:3
// (a 64-bit key with finer fields sorted no better and cost twice the radix passes: 0.6 -> 0.3 ms per
// 10^6 rays).  The optional work bucket orders wavefronts longest-first (LPT) from the straight-line
// distance to the field boundary; off by default (measured: -2 % on config 4, +12 % on config 3).
template <typename DirT>
__device__ __forceinline__ uint32_t tile_key(const uint32_t *pos, const DirT *dir, uint32_t i, uint32_t n, const RayBounds *rb,
                                             uint32_t Bx, uint32_t By, uint32_t Bz, int lpt)
{
    const uint32_t p[3] = {pos[3 * (size_t)i], pos[3 * (size_t)i + 1], pos[3 * (size_t)i + 2]};
    const float dx = (float)dir[3 * (size_t)i], dy = (float)dir[3 * (size_t)i + 1], dz = (float)dir[3 * (size_t)i + 2];
    const float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
    // (w, u, v) = (dominant axis, slower orthogonal axis, faster orthogonal axis); z is memory-minor
    int aw, au, av;
    uint32_t dom;
    if (ax >= ay && ax >= az) { aw = 0; au = 1; av = 2; dom = dx < 0 ? 1u : 0u; }
    else if (ay >= az)        { aw = 1; au = 0; av = 2; dom = dy < 0 ? 3u : 2u; }
    else                      { aw = 2; au = 0; av = 1; dom = dz < 0 ? 5u : 4u; }
    const float ru = (float)(rb->hi[au] - rb->lo[au]) + 1.0f, rv = (float)(rb->hi[av] - rb->lo[av]) + 1.0f;
    const float rw = (float)(rb->hi[aw] - rb->lo[aw]) + 1.0f;
    // tile edge W (16.16 units): W^2 * (n / (ru*rv)) = 64 rays; if the rays fill a volume rather
    // than a sheet, depth bins of the same size thin the density by rw/W -- one fixed-point
    // refinement step is enough for a grouping heuristic.
    float W = sqrtf(64.0f * ru * rv / (float)n);
    if (rw > 2.0f * W) W = cbrtf(64.0f * ru * rv * rw / (float)n);
    W = fminf(fmaxf(W, 4096.0f), 64.0f * 65536.0f);
    W = fmaxf(W, fmaxf(ru, rv) / 254.0f);                        // at most 255 tiles per axis (bigger tiles beyond)
    const float invW = 1.0f / W;
    const float fu = (float)(p[au] - rb->lo[au]) * invW, fv = (float)(p[av] - rb->lo[av]) * invW;
    uint32_t tu = (uint32_t)fu, tv = (uint32_t)fv;
    tu = tu > 255u ? 255u : tu;
    tv = tv > 255u ? 255u : tv;
    uint32_t row = (uint32_t)((fu - (float)tu) * 8.0f), col = (uint32_t)((fv - (float)tv) * 8.0f);
    row = row > 7u ? 7u : row;
    col = col > 7u ? 7u : col;
    uint32_t wb = (uint32_t)((float)(p[aw] - rb->lo[aw]) * fminf(invW, 15.0f / rw));
    wb = wb > 15u ? 15u : wb;
    if (tu & 1u) { tv = 255u - tv; col = 7u - col; }             // serpentine: odd tile columns run backwards
    uint32_t bucket = 7u;
    if (lpt) {
        // straight-line exit distance inside [0, B-1]^3, relative to the longest axis
        const float d3[3] = {dx, dy, dz};
        const float hi3[3] = {(float)(Bx - 1) * 65536.0f, (float)(By - 1) * 65536.0f, (float)(Bz - 1) * 65536.0f};
        float t = 3.0e38f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float pd = (float)p[d];
            if (d3[d] > 0.0f) t = fminf(t, (hi3[d] - pd) / d3[d]);
            else if (d3[d] < 0.0f) t = fminf(t, -pd / d3[d]);
        }
        const float len = t < 3.0e38f ? t * sqrtf(dx * dx + dy * dy + dz * dz) : 0.0f;
        const int b = (int)(len * 6.0f / fmaxf(fmaxf(hi3[0], hi3[1]), hi3[2]));
        bucket = (uint32_t)(b < 0 ? 0 : (b > 7 ? 7 : b));
    }
    return ((7u - bucket) << 29) | (dom << 26) | (wb << 22) | (tu << 14) | (tv << 6) | (row << 3) | col;
}

template <typename DirT>
__global__ __launch_bounds__(256) void k_tile_keys(const uint32_t *pos, const DirT *dir, uint32_t n,
                                                   const RayBounds *rb, uint32_t Bx, uint32_t By, uint32_t Bz,
                                                   int lpt, uint32_t *keys, uint32_t *vals)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = tile_key<DirT>(pos, dir, i, n, rb, Bx, By, Bz, lpt);
    vals[i] = i;
}

// ---- the tile order without a general sort (round 3) -------------------------------------------------------------
// The key above is (dom | depth bin | tile u | tile v | row | column): what the march needs is that the rays of one tile
// end up in 64 consecutive slots, tiles in serpentine order, and -- a refinement -- row-major order inside a tile.  A
// radix sort of the whole key (rocPRIM, rounds 1 and 2: 21 dependent launches, about 150 us at 10^6 rays, the one library
// call on the timed path) does far more than that.  Here: count the rays per tile (tile u, tile v: 65 536 bins, global
// atomics), scan the bins in one workgroup, scatter the rays to their tile's segment, and let one wavefront per tile put
// its segment -- about 64 rays -- in order by counting ranks out of LDS.  Dominant direction and depth bin become the leading
// bits of the in-tile key, so that a tile holding rays of several kinds keeps them in separate runs.  Four launches.
constexpr uint32_t kTileBins = 65536u;
constexpr uint32_t kTileSortMax = 512u;                 // largest segment ordered in place; longer ones (the tile grid is sized
                                                        // for 64 rays per tile: only grossly uneven ray sets) stay as scattered
__device__ __forceinline__ uint32_t tile_bin(uint32_t key) { return (key >> 6) & 0xFFFFu; }
__device__ __forceinline__ uint32_t tile_sub(uint32_t key) { return ((key >> 22) & 0x7Fu) << 6 | (key & 0x3Fu); }   // dom:3 | depth:4 | row:3 | col:3

// Rays usually arrive in some raster order, so consecutive lanes of a wavefront fall into the same tile in runs (8 at a time
// for a lattice): one atomic per RUN instead of one per lane -- same-address atomics of one instruction serialise, and with
// one per ray the two atomic passes took 100 and 121 us at 10^6 rays, more than the radix sort they replace.  run_of() gives
// every lane the first lane of its run of equal bins among consecutive lanes, its place in the run and the run's length.
struct LaneRun { uint32_t first, place, length; };
__device__ __forceinline__ LaneRun run_of(uint32_t bin, bool valid)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t prev = (uint32_t)__shfl_up((int)bin, 1);
    const bool pvalid = __shfl_up((int)valid, 1) != 0;
    const bool starts = lane == 0u || prev != bin || !pvalid || !valid;
    const unsigned long long m = __ballot(starts);                        // bit l: lane l starts a run
    const unsigned long long below = m & (~0ull >> (63u - lane));         // run starts at or below this lane (lane 0 always)
    const uint32_t first = 63u - (uint32_t)__builtin_clzll(below);
    const unsigned long long above = lane == 63u ? 0ull : (m >> (lane + 1u));
    const uint32_t end = above ? lane + (uint32_t)__builtin_ctzll(above) : 63u;   // last lane of the run
    return {first, lane - first, end - first + 1u};
}

template <typename DirT>
__global__ __launch_bounds__(256) void k_tile_hist(const uint32_t *pos, const DirT *dir, uint32_t n, const BoundsPartial *partial,
                                                   uint32_t nparts, RayBounds *rb_out, uint32_t Bx, uint32_t By, uint32_t Bz, uint32_t *keys,
                                                   uint32_t *bins)
{
    __shared__ RayBounds s_rb;
    __shared__ uint32_t s_red[4][6];
    fold_bounds(partial, nparts, &s_rb, s_red);            // every workgroup folds the bounds for itself ...
    if (blockIdx.x == 0 && threadIdx.x < 7) {              // ... and the first leaves them (and a zeroed work sum) for the later kernels
        if (threadIdx.x < 3) rb_out->lo[threadIdx.x] = s_rb.lo[threadIdx.x];
        else if (threadIdx.x < 6) rb_out->hi[threadIdx.x - 3] = s_rb.hi[threadIdx.x - 3];
        else rb_out->work_sum = 0ull;
    }
    const RayBounds *rb = &s_rb;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    uint32_t key = 0;
    if (valid) { key = tile_key<DirT>(pos, dir, i, n, rb, Bx, By, Bz, 0); keys[i] = key; }
    const LaneRun r = run_of(tile_bin(key), valid);
    if (valid && r.place == 0u) atomicAdd(&bins[tile_bin(key)], r.length);
}

// Exclusive scan of the 65 536 tile counts in two levels: kScanBlocks workgroups scan 1 024 bins each in place and leave their
// totals; the two consumers (scatter, in-tile order) add the prefix of the totals themselves (64 words: one wavefront reduction).
// (One workgroup scanning all bins took 18 us: the largest kernel of the chain after the pre-march.)
constexpr uint32_t kScanBlocks = 64, kScanPer = kTileBins / kScanBlocks;      // 1 024 bins per workgroup, one per thread
__global__ __launch_bounds__(1024) void k_tile_scan(uint32_t *bins, uint32_t *totals)
{
    __shared__ uint32_t wsum[16];
    const uint32_t tid = threadIdx.x;
    uint32_t *b = bins + blockIdx.x * kScanPer;
    const uint32_t mine = b[tid];
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o); if ((tid & 63u) >= (uint32_t)o) incl += t; }
    if ((tid & 63u) == 63u) wsum[tid >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < (tid >> 6); ++w) base += wsum[w];
    b[tid] = base + incl - mine;
    if (tid == 1023u) totals[blockIdx.x] = base + incl;
}
// first slot of scan block `blk`: the sum of the totals of the blocks before it (all 64 lanes of the wavefront take part)
__device__ __forceinline__ uint32_t scan_base(const uint32_t *totals, uint32_t blk)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t v = lane < blk ? totals[lane] : 0u;
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
    return v;
}

// every ray to its tile's segment; the bin's word ends up as the segment's END relative to its scan block (the next bin's start)
__global__ __launch_bounds__(256) void k_tile_scatter(const uint32_t *keys, uint32_t n, uint32_t *bins, const uint32_t *totals,
                                                      uint32_t *sub_out, uint32_t *ray_out)
{
    __shared__ uint32_t s_base[kScanBlocks];
    if (threadIdx.x < 64u) {                               // exclusive prefix of the 64 totals, once per workgroup
        const uint32_t t = totals[threadIdx.x];
        uint32_t incl = t;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)incl, o); if (threadIdx.x >= (uint32_t)o) incl += u; }
        s_base[threadIdx.x] = incl - t;
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    const uint32_t key = valid ? keys[i] : 0u;
    const uint32_t bin = tile_bin(key);
    const LaneRun r = run_of(bin, valid);
    uint32_t base = 0;
    if (valid && r.place == 0u) base = atomicAdd(&bins[bin], r.length);       // one atomic per run of equal tiles
    base = (uint32_t)__shfl((int)base, (int)r.first);
    if (!valid) return;
    const uint32_t at = s_base[bin / kScanPer] + base + r.place;
    sub_out[at] = tile_sub(key);
    ray_out[at] = i;
}

// one wavefront per tile: rank of every ray of the segment by (in-tile key, position in the segment) -- unique, so the
// rank is the number of smaller ones -- counted against the segment's keys in LDS
__global__ __launch_bounds__(256) void k_tile_order(const uint32_t *bins, const uint32_t *totals, const uint32_t *sub, const uint32_t *ray,
                                                    uint32_t *order)
{
    __shared__ uint32_t s_key[4][kTileSortMax];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t bin = blockIdx.x * 4u + wave;
    const uint32_t blk = bin / kScanPer;
    const uint32_t end_rel = bins[bin], start_rel = (bin % kScanPer) ? bins[bin - 1u] : 0u;   // relative to the scan block
    const uint32_t cnt = end_rel - start_rel;
    if (cnt == 0u) return;                                 // (wave-uniform)
    const uint32_t start = scan_base(totals, blk) + start_rel;
    if (cnt > kTileSortMax) {                              // copy through
        for (uint32_t e = lane; e < cnt; e += 64u) order[start + e] = ray[start + e];
        return;
    }
    uint32_t *k = s_key[wave];
    for (uint32_t e = lane; e < cnt; e += 64u) k[e] = (sub[start + e] << 9) | e;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t e = lane; e < cnt; e += 64u) {
        const uint32_t mine = k[e];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < cnt; ++j) rank += k[j] < mine ? 1u : 0u;
        order[start + rank] = ray[start + e];
    }
}

// Longest-first launch order of whole wavefronts.  A launch of W wavefronts on S resident slots runs
// W/S "rounds"; with caller/tile order the last round starts long marches late and the GPU drains
// half empty (tools/timeline.py: 26 % of a config-3 launch ran below half occupancy).  Each
// wavefront (64 consecutive slots of the tile order) gets a work estimate and wavefronts are launched
// in descending order of it.  Only the launch order changes: slot -> ray stays the tile order, so a
// wavefront's rays and its brick are the same as without this pass, and so are all results.
//
// The estimate is a coarse march of four representative rays per wavefront through the real field:
// the same update as the march step (.cu:344-348) in plain float arithmetic, but one nearest-voxel
// sample stands for kCoarse consecutive steps.  (A straight-line distance estimate was tried first:
// in a strongly refracting field it is off by 2x for the rays that are bent back into the volume,
// and exactly those wavefronts then started last.)  It only has to rank wavefronts, not to be exact.
#ifndef VRT_COARSE
#define VRT_COARSE 64
#endif
constexpr int kCoarse = VRT_COARSE;  // fine steps per coarse step
constexpr int kCoarseMaxLoops = 2048;
// Work key of a wavefront: its estimate in half coarse steps (an LDS step counts 2, see VRT_WW_GLOBAL_W), clamped.
constexpr int kWaveKeyMax = 8192;
#ifndef VRT_WW_SPREAD_VOX
#define VRT_WW_SPREAD_VOX 0          // 0 = every coarse step counts the same (the shipped setting).  Development: count a coarse step
                                     // VRT_WW_GLOBAL_W / 2 times when the bounding box of the wavefront's 16 representatives outgrows a
                                     // brick of this many voxels (it will be gathering from global memory then).  Measured with 400 /
                                     // 640 / 1 000 voxels and weights 2.5 / 2: config 5 12.55 - 13.16 ms against 12.42 - 12.46, config 3
                                     // 7.07 - 7.16 against 6.90 - 6.93 -- ranking by plain step counts orders better (as in round 2)
#endif
#ifndef VRT_WW_GLOBAL_W
#define VRT_WW_GLOBAL_W 5
#endif
#ifndef VRT_LPT_SHIFT
#define VRT_LPT_SHIFT 0
#endif

template <typename FieldT, typename DirT>
__global__ __launch_bounds__(256) void k_wave_work(const typename FieldTraits<FieldT>::vox *field, const uint32_t *pos,
                                                   const DirT *dir, const uint32_t *order, uint32_t n, uint32_t Bx,
                                                   uint32_t By, uint32_t Bz, float invx, float invy, float invz,
                                                   uint32_t iterations, uint32_t *wkeys, uint32_t *wvals, RayBounds *rb)
{
    // (Round 3: sampling an 8 x coarser copy of the field that fits the L2 instead -- the pre-march is a chain of dependent
    // samples, one memory latency each -- cut this kernel from 62 to 40 us, but the launch order it produced was worse: the
    // march of config 5 ran 2 % longer (13.1 vs 12.85 ms), config 3's the same.  Not kept.)
#ifndef VRT_WW_REPS_LOG2
#define VRT_WW_REPS_LOG2 4
#endif
    // Representative rays per wavefront: 16 (rounds 1 and 2: 4).  The estimate is the maximum over them, and how well it ranks the
    // wavefronts shows in the march itself (MI355X, kernel ms, config 5 / config 3): 4 rays 13.28 / 6.95, 8 rays 12.69 / 6.91,
    // 16 rays 12.42 / 6.87, 32 rays 12.9 - 13.0 / 7.1, all 64 12.68 / 7.05 with the whole step 0.9 ms slower (the pre-march then
    // costs more than it returns).  The slice length (1.15 x the mean estimate) was swept again with 16: still the optimum.
    constexpr uint32_t kRepsLog2 = VRT_WW_REPS_LOG2, kReps = 1u << kRepsLog2;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t w = t >> kRepsLog2, nw = (n + 63u) >> 6;
    __shared__ uint32_t s_sum;
    if (threadIdx.x == 0) s_sum = 0u;
    __syncthreads();
    const bool valid = w < nw;                             // (threads past the end idle through the loop: the block meets again below)
    // spread over the tile: rows 0 / 2 / 5 / 7 of the 8 x 8 tile for four, every row for eight
    const uint32_t rep_slot = kReps == 4u ? (uint32_t[4]){5u, 23u, 40u, 58u}[t & 3u] : ((t & (kReps - 1u)) * (64u / kReps) + ((t * 5u) & (64u / kReps - 1u)));
    uint32_t slot = (valid ? w : nw - 1u) * 64u + rep_slot;
    slot = slot < n ? slot : n - 1u;
    const uint32_t i = order ? order[slot] : slot;
    float px = (float)pos[3 * (size_t)i], py = (float)pos[3 * (size_t)i + 1], pz = (float)pos[3 * (size_t)i + 2];
    const float dsc = sizeof(DirT) == 4 ? 65536.0f : 256.0f;             // internal direction scale (.cu:328-331)
    float dx = (float)dir[3 * (size_t)i] * dsc, dy = (float)dir[3 * (size_t)i + 1] * dsc, dz = (float)dir[3 * (size_t)i + 2] * dsc;
    const float hx = (float)(Bx - 1) * 65536.0f, hy = (float)(By - 1) * 65536.0f, hz = (float)(Bz - 1) * 65536.0f;
    const uint32_t cap = iterations / kCoarse + 1u;
    const uint32_t loops_max = cap < (uint32_t)kCoarseMaxLoops ? cap : (uint32_t)kCoarseMaxLoops;
    const float m = (float)kCoarse;
    bool alive = valid;
    uint32_t est = 0;                                     // coarse steps until this representative stops
    uint32_t tw = 0;                                      // the wavefront's estimated TIME, in half LDS steps (same in all lanes of a group)
    constexpr bool kSpread = VRT_WW_SPREAD_VOX > 0;
    for (uint32_t loops = 0; loops < loops_max; ++loops) {
        alive = alive && px >= 0.0f && py >= 0.0f && pz >= 0.0f && px < hx && py < hy && pz < hz;   // false on NaN
        if (alive) {
            const uint32_t cx = (uint32_t)(px * (1.0f / 65536.0f)), cy = (uint32_t)(py * (1.0f / 65536.0f)), cz = (uint32_t)(pz * (1.0f / 65536.0f));
            f4 g;
            cvt_vox(field[((size_t)cx * By + cy) * Bz + cz], g);
            if (g.w > 0.0f) alive = false;
            else {
                dx = fmaf(g.x * m, invx, dx);
                dy = fmaf(g.y * m, invy, dy);
                dz = fmaf(g.z * m, invz, dz);
                const float il = m * 0x42000000p0f / (dx * dx + dy * dy + dz * dz);
                px = fmaf(dx * invx, il, px);
                py = fmaf(dy * invy, il, py);
                pz = fmaf(dz * invz, il, pz);
                ++est;
            }
        }
        if (__ballot(alive) == 0ull) break;               // (the four groups of a wavefront of this kernel leave together)
        if constexpr (kSpread) {
            // how far apart are the group's representatives?  When their bounding box (plus the trilinear halo) outgrows a brick
            // the wavefront will be gathering from global memory, at about 2.4 x the cost of a step out of LDS
            float lo[3] = {alive ? px : 3.0e38f, alive ? py : 3.0e38f, alive ? pz : 3.0e38f};
            float hi[3] = {alive ? px : -3.0e38f, alive ? py : -3.0e38f, alive ? pz : -3.0e38f};
#pragma unroll
            for (uint32_t o = 1; o < kReps; o <<= 1)
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    lo[d] = fminf(lo[d], __shfl_xor(lo[d], (int)o));
                    hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], (int)o));
                }
            if (hi[0] >= lo[0]) {
                const float vol = ((hi[0] - lo[0]) * (1.0f / 65536.0f) + 2.0f) * ((hi[1] - lo[1]) * (1.0f / 65536.0f) + 2.0f) * ((hi[2] - lo[2]) * (1.0f / 65536.0f) + 2.0f);
                tw += vol > (float)VRT_WW_SPREAD_VOX ? (uint32_t)VRT_WW_GLOBAL_W : 2u;
            }
        }
    }
    // a wavefront marches until its last ray stops: the largest of the representatives' estimates
#pragma unroll
    for (uint32_t o = 1; o < kReps; o <<= 1) est = max(est, (uint32_t)__shfl_xor((int)est, (int)o));
    if constexpr (!kSpread) tw = 2u * est;
    if (valid && (t & (kReps - 1u)) == 0) {
        const uint32_t k = (tw >> VRT_LPT_SHIFT) < (uint32_t)kWaveKeyMax ? (tw >> VRT_LPT_SHIFT) : (uint32_t)kWaveKeyMax;
        wkeys[w] = (uint32_t)kWaveKeyMax - k;              // ascending sort = longest first
        wvals[w] = w;
        atomicAdd(&s_sum, est);                            // LDS: 64 adds per block, then one global atomic per block
    }
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&rb->work_sum, (unsigned long long)s_sum);
}

// Launch order of the wavefronts from their work keys (kWaveKeyMax - estimate, 13 bits): ONE workgroup sorts them by
// counting in LDS -- histogram with LDS atomics, scan of the 8 193 bins, scatter.  (Rounds 1 and 2 called rocPRIM's radix sort
// here: six to eight small dependent launches, about 40 us plus their gaps, for 16 384 keys.)  Ties come out in arbitrary
// order: the launch order is a scheduling hint, results never depend on it.
constexpr int kWaveSortThreads = 1024;
__global__ __launch_bounds__(kWaveSortThreads) void k_wave_sort(const uint32_t *__restrict__ wkeys, uint32_t nw, uint32_t *__restrict__ order_out)
{
    constexpr int kBins = kWaveKeyMax + 1;
    constexpr int kPer = (kBins + kWaveSortThreads - 1) / kWaveSortThreads;     // bins per thread in the scan (3)
    __shared__ uint32_t cnt[kBins];
    __shared__ uint32_t wsum[kWaveSortThreads / 64];
    const uint32_t tid = threadIdx.x;
    for (uint32_t b = tid; b < (uint32_t)kBins; b += kWaveSortThreads) cnt[b] = 0u;
    __syncthreads();
    for (uint32_t i = tid; i < nw; i += kWaveSortThreads) {
        const uint32_t k = wkeys[i];
        atomicAdd(&cnt[k < (uint32_t)kBins ? k : (uint32_t)kBins - 1u], 1u);
    }
    __syncthreads();
    // exclusive scan: thread t owns bins [t * kPer, t * kPer + kPer)
    uint32_t mine[kPer], tot = 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const uint32_t b = tid * kPer + j;
        mine[j] = b < (uint32_t)kBins ? cnt[b] : 0u;
        tot += mine[j];
    }
    uint32_t incl = tot;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o); if ((tid & 63u) >= (uint32_t)o) incl += t; }
    if ((tid & 63u) == 63u) wsum[tid >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < (tid >> 6); ++w) base += wsum[w];
    uint32_t run = base + incl - tot;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const uint32_t b = tid * kPer + j;
        if (b < (uint32_t)kBins) cnt[b] = run;
        run += mine[j];
    }
    __syncthreads();
    for (uint32_t i = tid; i < nw; i += kWaveSortThreads) {
        const uint32_t k = wkeys[i];
        order_out[atomicAdd(&cnt[k < (uint32_t)kBins ? k : (uint32_t)kBins - 1u], 1u)] = i;
    }
}

// grow-only device buffer interface expected from the caller (see DevBuf in vrt_engine.hip)
template <typename Buf>
hipError_t build_tile_order(const uint32_t *d_pos, const void *d_dir, int dir_is_i16, uint32_t n,
                            uint32_t Bx, uint32_t By, uint32_t Bz, int lpt, Buf &keys,
                            Buf &keys_alt, Buf &vals, Buf &vals_alt, Buf &tmp, Buf &bounds, hipStream_t s,
                            const uint32_t **order_out, int wave_lpt = 0, uint32_t iterations = 0,
                            const uint32_t **wave_order_out = nullptr, const void *field = nullptr, int field_is_i16 = 0,
                            const float *inv = nullptr, const ZeroList *also_zero = nullptr)
{
    hipError_t e;
    const uint32_t nw = (n + 63u) >> 6;
    const size_t kbytes = ((size_t)n > 2 * (size_t)nw ? (size_t)n : 2 * (size_t)nw) * 4;   // keys/vals are reused for the wavefront sort
    if ((e = keys.reserve(kbytes)) != hipSuccess) return e;
    if ((e = keys_alt.reserve((size_t)n * 4)) != hipSuccess) return e;
    if ((e = vals.reserve(kbytes)) != hipSuccess) return e;
    if ((e = vals_alt.reserve((size_t)n * 4)) != hipSuccess) return e;
    // bounds buffer: the RayBounds the later kernels read, then the per-workgroup partials
    constexpr size_t kPartialOff = 64;
    if ((e = bounds.reserve(kPartialOff + kBoundsBlocks * sizeof(BoundsPartial))) != hipSuccess) return e;
    RayBounds *rb = (RayBounds *)bounds.p;
    BoundsPartial *partial = (BoundsPartial *)((char *)bounds.p + kPartialOff);
    const unsigned rgrid = (unsigned)((n + 255) / 256 < kBoundsBlocks ? (n + 255) / 256 : kBoundsBlocks);
    const unsigned grid = (n + 255) / 256;
    const char *order_env = getenv("VRT_RAY_ORDER");                       // (development / tests: "radix" = the rocPRIM sort of rounds 1 and 2)
    const bool use_radix = order_env && strcmp(order_env, "radix") == 0;
    ZeroList z = {{nullptr, nullptr, nullptr, nullptr}, {0u, 0u, 0u, 0u}};
    if (also_zero) z = *also_zero;                         // slots 0..2: the engine's scheduler words; slot 3: the tile bins
    if (!lpt && !use_radix) {
        // bounds (+ the call's zeroing) -> count per tile -> scan -> scatter -> order inside the tile (see k_tile_hist above)
        if ((e = tmp.reserve((kTileBins + kScanBlocks) * sizeof(uint32_t))) != hipSuccess) return e;
        uint32_t *bins = (uint32_t *)tmp.p, *totals = bins + kTileBins;
        z.p[3] = bins; z.words[3] = kTileBins;
        k_ray_bounds_partial<<<rgrid, 256, 0, s>>>(d_pos, n, partial, z);
        if (dir_is_i16) k_tile_hist<int16_t><<<grid, 256, 0, s>>>(d_pos, (const int16_t *)d_dir, n, partial, rgrid, rb, Bx, By, Bz, (uint32_t *)keys.p, bins);
        else            k_tile_hist<float><<<grid, 256, 0, s>>>(d_pos, (const float *)d_dir, n, partial, rgrid, rb, Bx, By, Bz, (uint32_t *)keys.p, bins);
        k_tile_scan<<<kScanBlocks, 1024, 0, s>>>(bins, totals);
        k_tile_scatter<<<grid, 256, 0, s>>>((const uint32_t *)keys.p, n, bins, totals, (uint32_t *)keys_alt.p, (uint32_t *)vals.p);
        k_tile_order<<<kTileBins / 4u, 256, 0, s>>>(bins, totals, (const uint32_t *)keys_alt.p, (const uint32_t *)vals.p, (uint32_t *)vals_alt.p);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    } else {
        k_ray_bounds_partial<<<rgrid, 256, 0, s>>>(d_pos, n, partial, z);
        k_bounds_finish<<<1, 256, 0, s>>>(partial, rgrid, rb);
        if (dir_is_i16) k_tile_keys<int16_t><<<grid, 256, 0, s>>>(d_pos, (const int16_t *)d_dir, n, rb, Bx, By, Bz, lpt, (uint32_t *)keys.p, (uint32_t *)vals.p);
        else            k_tile_keys<float><<<grid, 256, 0, s>>>(d_pos, (const float *)d_dir, n, rb, Bx, By, Bz, lpt, (uint32_t *)keys.p, (uint32_t *)vals.p);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        // per-ray length buckets in the key (sort_rays = 2, opt-in): the general radix sort of the whole key
        size_t tmp_bytes = 0;
        e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, (const uint32_t *)keys.p, (uint32_t *)keys_alt.p,
                                      (const uint32_t *)vals.p, (uint32_t *)vals_alt.p, n, 0, 32, s);
        if (e != hipSuccess) return e;
        if ((e = tmp.reserve(tmp_bytes ? tmp_bytes : 16)) != hipSuccess) return e;
        e = rocprim::radix_sort_pairs(tmp.p, tmp_bytes, (const uint32_t *)keys.p, (uint32_t *)keys_alt.p,
                                      (const uint32_t *)vals.p, (uint32_t *)vals_alt.p, n, 0, lpt ? 32 : 29, s);
        if (e != hipSuccess) return e;
    }
    *order_out = (const uint32_t *)vals_alt.p;
    if (wave_lpt && wave_order_out && field && inv && nw > 1) {
        // the ray sort's input buffers are free again: [0, nw) = wavefront keys / ids, [nw, 2 nw) = sorted
        uint32_t *wk = (uint32_t *)keys.p, *wv = (uint32_t *)vals.p;
        const uint32_t *ord = (const uint32_t *)vals_alt.p;
        const unsigned wgrid = ((nw << VRT_WW_REPS_LOG2) + 255u) / 256u;
        using vf = const FieldTraits<float>::vox *;
        using vi = const FieldTraits<int16_t>::vox *;
#define VRT_WW(FT, DT, FP) k_wave_work<FT, DT><<<wgrid, 256, 0, s>>>((FP)field, d_pos, (const DT *)d_dir, ord, n, Bx, By, Bz, \
                                                                     inv[0], inv[1], inv[2], iterations, wk, wv, rb)
        if (field_is_i16) { if (dir_is_i16) VRT_WW(int16_t, int16_t, vi); else VRT_WW(int16_t, float, vi); }
        else              { if (dir_is_i16) VRT_WW(float, int16_t, vf);   else VRT_WW(float, float, vf); }
#undef VRT_WW
        if ((e = hipGetLastError()) != hipSuccess) return e;
        k_wave_sort<<<1, kWaveSortThreads, 0, s>>>(wk, nw, wv + nw);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        *wave_order_out = wv + nw;
    }
    return hipSuccess;
}

}  // namespace vrt
