// vrt_slab.h -- Kernel 2: SLAB.  Wave-cooperative LDS staging of the voxel brick a
// wavefront is about to walk.
//
// Same per-ray arithmetic as the DIRECT kernel (vrt_march.h, bit-identical results); what
// changes is where the 8 trilinear corners come from.  The reference's GPU path gathers them
// from global memory on every step (/root/reference/src/cuda_volume_raytracer.cu:138-143 via
// :397-414).  Here each 64-lane wavefront owns a private LDS segment and repeats:
//
//   plan   batched DPP wave reductions of the live lanes' positions and per-step displacement; pick the
//          largest step count K whose swept bounding brick (+1 voxel for the upper trilinear
//          corners), clipped to the field, fits the segment
//   stage  all 64 lanes copy the brick L2/HBM -> LDS as fp32 voxels carrying the interpolation's 2^-48 (prescale_ok);
//          every lane issues all of its (at most VPL) loads before the first LDS write, so one memory latency
//          covers the brick
//   march  first the steps during which no lane can PROVABLY leave the brick (safe_steps): a loop with no brick
//          test, no window test, no ballot; then the rest of the planned K with all corners read from LDS and a
//          ballot that ends the round as soon as a live lane would leave the brick (rays bend, the plan is only a
//          prediction) and ends the wavefront when every lane has terminated
//
// Wavefronts whose rays do not fit a brick for at least kSlabMinK steps (incoherent rays) fall back to
// the global gather for a few steps and re-plan.  Waves never synchronise with each other:
// a 256-thread workgroup is four independent wavefronts with four LDS segments.
// Scheduling: one wavefront per group of 64 rays (static), persistent wavefronts drawing whole groups from a
// counter, or persistent wavefronts drawing time slices of groups from a FIFO (SCHED, below).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vrt_march.h"

extern "C" __device__ unsigned int __ockl_wfred_max_u32(unsigned int);    // development builds only (statistics)

namespace vrt {

constexpr int kSlabWavesPerBlock = 4;
// A wavefront in global-gather mode is latency-bound (tools/timeline.py: 2.6 us per step against 0.65 us from LDS,
// and the launch ends with exactly these wavefronts): its few instructions go first.  Measured -2.8 % on config 3,
// +-1 % on configs 4 and 5 (round 1).  Raised priority for the 256 ... 2048 longest groups over their whole life was
// measured in round 2: no effect at any count.
#ifndef VRT_PRIO_DIRECT
#define VRT_PRIO_DIRECT 3
#endif
#ifndef VRT_SWEEP_REFRESH
#define VRT_SWEEP_REFRESH 8
#endif
#ifndef VRT_SLAB_MAXK
#define VRT_SLAB_MAXK 192
#endif
constexpr int kSlabMaxK = VRT_SLAB_MAXK;   // longest planned round, steps
#ifndef VRT_SLAB_DIRECT_STEPS
#define VRT_SLAB_DIRECT_STEPS 64
#endif
constexpr int kSlabDirectSteps = VRT_SLAB_DIRECT_STEPS;    // global-gather steps before re-planning for a wave that stages no brick
#ifndef VRT_SLAB_MINSAFE
#define VRT_SLAB_MINSAFE 6
#endif
constexpr int kSlabMinSafe = VRT_SLAB_MINSAFE;   // a proven run shorter than this is not worth a round: take the tested loop
#ifndef VRT_SLAB_MINK
#define VRT_SLAB_MINK 8
#endif
constexpr int kSlabMinK = VRT_SLAB_MINK; // rounds shorter than this are not worth staging: gather from global memory

__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// wave-uniform "some active lane has b": the compare mask itself (no 0/1 materialisation as with __ballot(int))
__device__ __forceinline__ bool any_lane(bool b) { return __builtin_amdgcn_ballot_w64(b) != 0ull; }
// a * b + c with a, b < 2^24 in one instruction (the compiler otherwise builds the brick address from
// v_mad_u64_u32 + v_mul_u32_u24 + two shifts + v_add3); b is wave-uniform
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c));
    return d;
}
// c - a * b for a, b, c < 2^23 (the remainder of a division: a = quotient, b = wave-uniform divisor) in one instruction
__device__ __forceinline__ uint32_t nmad24(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(0u - b), "v"(c));
    return d;
}
// (a * m) >> 22 for a < 2^10, m < 2^22 + 4 (stage_magic): the quotient a / d
__device__ __forceinline__ uint32_t div22(uint32_t a, uint32_t m)
{
    uint32_t d;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(d) : "s"(m), "v"(a));
    return d >> 22;
}
// a.hi16 * b + c with b < 2^16 (wave-uniform): the cell coordinate is the high half of a brick-relative position, so the
// shift is folded into the multiply (v_mad_u32_u16 with op_sel on the first source)
__device__ __forceinline__ uint32_t mad_hi16(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(a), "s"(b), "v"(c));
    return d;
}
// The persistent variant's group counter is bumped as straight-line, wave-uniform code: the atomic executes
// in lane 0 only by narrowing EXEC inside the asm block (restored before it ends).  An `if (lane == 0)`
// region around it made the compiler carry the divergence into the march that follows (lane 0 then
// marched alone and staged only its own share of the brick) -- wave-level code must not sit behind a
// divergent branch.
// (the address is wave-uniform: it goes in as a scalar base beside a zero vector offset, so no counter or queue pointer
// occupies a register pair per lane -- the 80-register kernels were spilling one)
__device__ __forceinline__ uint32_t lane0_atomic_add(uint32_t *ptr, uint32_t v)
{
    uint32_t old;
    unsigned long long saved;
    asm volatile("s_mov_b64 %1, exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "global_atomic_add %0, %2, %3, %4 sc0\n\t"
                 "s_waitcnt vmcnt(0)\n\t"
                 "s_mov_b64 exec, %1"
                 : "=&v"(old), "=&s"(saved) : "v"(0u), "v"(v), "s"(ptr) : "memory");
    return __builtin_amdgcn_readfirstlane(old);
}
// exchange by lane 0: returns the old word and leaves `v` (one atomic, so that taking a queue word and emptying it cannot
// be separated by another wavefront's read)
__device__ __forceinline__ uint32_t lane0_atomic_swap(uint32_t *ptr, uint32_t v)
{
    uint32_t old;
    unsigned long long saved;
    asm volatile("s_mov_b64 %1, exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "global_atomic_swap %0, %2, %3, %4 sc0\n\t"
                 "s_waitcnt vmcnt(0)\n\t"
                 "s_mov_b64 exec, %1"
                 : "=&v"(old), "=&s"(saved) : "v"(0u), "v"(v), "s"(ptr) : "memory");
    return __builtin_amdgcn_readfirstlane(old);
}
// Scheduler words of the persistent schedules: head (items drawn), groups finished, error, tail (items appended).  They sit
// kCtrStride words apart (32 words = a 128-byte line each; round 2 had them on one line: the two layouts measure the same within 1 %, see DESIGN).
#ifndef VRT_CTR_STRIDE
#define VRT_CTR_STRIDE 32
#endif
constexpr uint32_t kCtrStride = VRT_CTR_STRIDE;
constexpr uint32_t kCtrHead = 0, kCtrDone = 1 * kCtrStride, kCtrErr = 2 * kCtrStride, kCtrTail = 3 * kCtrStride, kCtrWords = 3 * kCtrStride + 1;
// A wavefront of the time-sliced schedule that gives up waiting: the device error word counts it, the "groups finished" word
// is pushed past any group count so that every other waiter leaves at its next look (the waiters poll that word anyway: a
// second polled word on the counters' line was measured in round 3 -- idle wavefronts at the end of a launch saturate the
// line and the whole schedule ran 17 - 34 % slower), and the host word (pinned, written through to system scope) tells the
// caller that rays were left unfinished.
__device__ __forceinline__ void raise_sched_error(uint32_t *ctr, uint32_t *host_err)
{
    (void)lane0_atomic_add(ctr + kCtrErr, 1u);
    (void)lane0_atomic_add(ctr + kCtrDone, 0x40000000u);
    if (host_err) {
        unsigned long long saved;
        uint32_t one = 1u;
        asm volatile("s_mov_b64 %0, exec\n\t"
                     "s_mov_b64 exec, 1\n\t"
                     "global_store_dword %1, %2, %3 sc0 sc1\n\t"
                     "s_waitcnt vmcnt(0)\n\t"
                     "s_mov_b64 exec, %0"
                     : "=&s"(saved) : "v"(0u), "v"(one), "s"(host_err) : "memory");
    }
}
__device__ __forceinline__ uint64_t lane_mask(bool b) { return __builtin_amdgcn_ballot_w64(b); }
// compare-and-swap by lane 0 (same EXEC narrowing): *ptr = (*ptr == expect) ? value : *ptr; returns the old word
__device__ __forceinline__ uint32_t lane0_atomic_cas(uint32_t *ptr, uint32_t expect, uint32_t value)
{
    uint32_t old;
    unsigned long long saved;
    const unsigned long long pair = ((unsigned long long)expect << 32) | value;      // {data, compare} in consecutive registers
    asm volatile("s_mov_b64 %1, exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "global_atomic_cmpswap %0, %2, %3, %4 sc0\n\t"
                 "s_waitcnt vmcnt(0)\n\t"
                 "s_mov_b64 exec, %1"
                 : "=&v"(old), "=&s"(saved) : "v"(0u), "v"(pair), "s"(ptr) : "memory");
    return __builtin_amdgcn_readfirstlane(old);
}

// ---- hand-off primitives of the time-sliced schedule (SCHED == 2) -------------------------------------------------
// A group's ray state travels between wavefronts that may sit on different XCDs, whose L2s are not coherent with each
// other.  The parked bytes live in uncached device memory (hipDeviceMallocUncached) and are moved with write-through
// (sc1) stores and sc1 loads; the storing wave drains them (s_waitcnt vmcnt(0)) before lane 0 moves the group's flag,
// and flags / counters are touched by device-scope atomics only -- no cache-wide fences (they cost microseconds each).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_sc1_b128(void *ptr, u32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(ptr), "v"(v) : "memory");
}
__device__ __forceinline__ void load_sc1_2xb128(const void *p0, const void *p1, u32x4 &v0, u32x4 &v1)   // both in flight, one wait
{
    asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\t"
                 "global_load_dwordx4 %1, %3, off sc0 sc1\n\t"
                 "s_waitcnt vmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(p0), "v"(p1) : "memory");
}

// ---- batched wave reductions -------------------------------------------------------------------------------------
// The planner needs six to seven wave-wide minima / maxima per round.  One reduction is a chain of six dependent DPP
// operations, each of which must wait two cycles for its source; the library reductions (__ockl_wfred_*) came out one
// after the other, 36 dependent operations with their wait states -- most of the 5 000 cycles a planning round took
// (tools/phases.py).  Here N reductions advance together, stage by stage: the same 6 N DPP operations, but every
// stage's N instructions are independent and fill each other's wait states, and the chain is 6 long, not 6 N.
// Unsigned minima: a maximum is taken as the minimum of the complement; signed values are biased by 2^31.
// DPP controls (GFX9): row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143; a lane without a source keeps
// `old` = the identity 0xFFFFFFFF.  After row_shr 1, 2, 4, 8 lane 15 of each row of 16 holds its row's minimum;
// row_bcast:15 (rows 1 and 3) and row_bcast:31 (rows 2 and 3) leave the wave's minimum in lane 63.
template <int N>
__device__ __forceinline__ void wave_min_u32(uint32_t (&v)[N])
{
#define VRT_DPP_STAGE(CTRL, ROWMASK)                                                                             \
    _Pragma("unroll") for (int i = 0; i < N; ++i) {                                                             \
        const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)v[i], CTRL, ROWMASK, 0xf, false); \
        v[i] = t < v[i] ? t : v[i];                                                                             \
    }
    VRT_DPP_STAGE(0x111, 0xf) VRT_DPP_STAGE(0x112, 0xf) VRT_DPP_STAGE(0x114, 0xf) VRT_DPP_STAGE(0x118, 0xf)
    VRT_DPP_STAGE(0x142, 0xa) VRT_DPP_STAGE(0x143, 0xc)
#undef VRT_DPP_STAGE
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = (uint32_t)__builtin_amdgcn_readlane((int)v[i], 63);
}

// debug counters (MarchParams::stats != nullptr): per-wave events, added by lane 0
enum { kStatRounds = 0, kStatPlannedK, kStatLdsWaveSteps, kStatDirectRounds, kStatDirectWaveSteps, kStatVoxStaged,
       kStatEarlyExit, kStatProvenRounds, kStatCount, kStatProvenSteps = 13 };
__device__ __forceinline__ void stat_add(const MarchParams &p, int which, unsigned long long v)
{
#ifdef VRT_SLAB_STATS      // development builds only (tools/build_debug.sh): keeps the branches out of the hot loop
    if (p.stats && (threadIdx.x & 63u) == 0) atomicAdd(p.stats + which, v);
#else
    (void)p; (void)which; (void)v;
#endif
}

// Development builds with -DVRT_SLAB_PHASES (tools/build_debug.sh, tools/phases.py): every wavefront sums the shader
// cycles it spends planning / staging / marching from LDS / marching from global memory (s_memtime stamps at phase
// boundaries, scalar sums) and lane 0 adds them to MarchParams::stats[16..] when the wavefront ends, together with
// the wave-step counts and a (s_memtime, s_memrealtime) pair that gives the clock the kernel really ran at.
// Stamps never touch a ray: outputs are unchanged.  Production builds contain none of this.
#ifdef VRT_SLAB_PHASES
__device__ __forceinline__ unsigned long long ph_stamp()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
__device__ __forceinline__ unsigned long long ph_realtime()
{
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
enum { kPhPlan = 0, kPhStage, kPhLds, kPhDirect, kPhOther, kPhCount };
#define VRT_PH(i) do { const unsigned long long t_ = ph_stamp(); ph_acc[i] += t_ - ph_t; ph_t = t_; } while (0)
#else
#define VRT_PH(i) ((void)0)
#endif

struct Brick {
    uint32_t lo[3];   // first voxel
    uint32_t e[3];    // extents in voxels
};

// Wave-uniform planning inputs in 1/256-voxel units (24-bit positions), rounded outwards.
struct PlanIn {
    int32_t pmin8[3], pmax8[3];   // floor(pmin/256), ceil(pmax/256)
    int32_t sneg8[3], spos8[3];   // min(floor(smin/256), 0), max(ceil(smax/256), 0), clamped to +-2^20
};

// Brick swept by the wave during K steps (positions before each of the steps k = 0..K-1), plus
// one voxel for the upper trilinear corners, clipped to the field.  Returns the voxel count,
// saturated to cap+1 when it cannot fit.  All 32-bit scalar arithmetic.
__device__ __forceinline__ uint32_t plan_brick(const PlanIn &in, int K, const uint32_t B[3], uint32_t cap, Brick &b)
{
    uint32_t vol = 1;
    bool big = false;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int32_t a = in.pmin8[d] + (K - 1) * in.sneg8[d];
        int32_t c = in.pmax8[d] + (K - 1) * in.spos8[d];
        a = a < 0 ? 0 : a;
        int32_t l = a >> 8;
        int32_t h = (c >> 8) + 1;                        // +1: upper trilinear corner
        const int32_t top = (int32_t)B[d] - 1;
        h = h > top ? top : h;
        l = l > h ? h : l;
        b.lo[d] = (uint32_t)l;
        b.e[d] = (uint32_t)(h - l + 1);
        big |= b.e[d] > cap;
        vol = big ? cap + 1 : vol * b.e[d];
        big |= vol > cap;
    }
    return big ? cap + 1 : vol;
}

// ---- guaranteed rounds ------------------------------------------------------------------------------------------
// The planned K is a prediction (rays bend), so the general round tests every lane against the brick on every step
// (three compares, a handful of scalar mask operations and branches: 13 % of a step, tools/ubench/step_rate.hip).
// safe_steps() instead PROVES, per lane, a number of steps during which the ray cannot leave the staged brick; the
// wavefront takes the minimum over its live lanes and runs that many steps with no test at all.
//
// Notation (internal units): D = dir, E_d = D_d * inv_d, dot = |D|^2, C = 0x42000000.  One step (.cu:344-347):
//   D_d += g_d * inv_d,  |g_d * inv_d| <= G_d := gstep[d] (max |field_d| * |inv_d|: the interpolation is a convex
//   combination; its three fma levels and the exact 2^-48 add a factor (1 + 2^-20) at most, and the fma of the update
//   at most |D| * 2^-23 -- both covered by `slack` below);
//   pos_d += s_d,  s_d = roundf(E_d * (C / dot)),  |s_d| <= |E_d| * C / dot * (1 + 2^-20) + 0.5, sign(s_d) = sign(E_d).
// Over j <= K steps |D_d| grows by at most K G_d and |D| shrinks by at most R := K Gn, Gn := gstep_norm >= the length of g * inv
// (the interpolated gradient is a convex combination of corner vectors, so it is no longer than the longest one; taking the
// sum Gx + Gy + Gz instead made rho about 2.5 x larger and proved 58 % of config 3's LDS steps instead of 64 %; the
// launch time did not move, as with every other saving inside the LDS step in round 2).  With rho := R / |D| <= 1/4:
//   C / dot_j <= C / (dot (1 - rho)^2),
// so every step moves the ray by at most  up_d = (max(E_d, 0) + K G_d |inv_d|) * C / (dot (1 - rho)^2) + 0.5  towards +d and by
// at most dn_d (the same with max(-E_d, 0)) towards -d (rho is evaluated 2^-9 too large and the quotient 2^-9 too large again).  The ray gathers at positions p, p + s_1, ... before steps 0, 1, ...:
// it is inside the brick (cell <= e_d - 2, relative position in [0, H_d)) before step j as long as
// j * up_d <= H_d - 1 - p_d and j * dn_d <= p_d.  The divisions are evaluated with v_rcp_f32 and scaled down by
// (1 - 2^-13), far more than the three roundings involved; a NaN or infinite direction fails `ok` and yields 1 (the
// position the lane has now is inside the brick by construction).
// The same bound proves the fast-path window of the step (StepWindow): |D| stays within [3/4, 5/4] of its present
// value for K steps, so dot stays within [0.5625, 1.5625] dot (1 +- 2^-21); if [0.55, 1.6] dot lies strictly inside
// the window now, no step of the round can leave it and the loop needs no window test (`window_ok`).
__device__ __forceinline__ uint32_t safe_steps(const Ray &r /* relative to the brick origin */, float invx, float invy, float invz,
                                               float gx, float gy, float gz, float gnorm, int K, uint32_t hx, uint32_t hy, uint32_t hz /* H_d - 1 */,
                                               StepWindow win, bool &window_ok)
{
    const float dot = fma_(r.dz, r.dz, fma_(r.dx, r.dx, r.dy * r.dy));
    const float rc = __builtin_amdgcn_rcpf(dot);
    const float len = dot * __builtin_amdgcn_rsqf(dot);                   // |D| within a few ulp
    const float Kf = (float)K;
    const float slack = len * 0x1p-21f;
    const float Gx = gx + slack, Gy = gy + slack, Gz = gz + slack;
#ifdef VRT_SLAB_L1_SHRINK
    const float R = Kf * ((Gx + Gy) + Gz);
#else
    const float R = Kf * (gnorm + 2.0f * slack);                          // |D| shrinks by at most the LENGTH of the gradient per step
#endif
    const bool ok = R * 4.0f <= len;                                      // false for NaN / inf / tiny directions
    const float rho = (R * 1.002f) * __builtin_amdgcn_rcpf(len);          // >= R / |D|
    const float shrink = __builtin_amdgcn_rcpf(1.0f - rho);               // 1 / (1 - rho), rho <= 0.2505
    const float Q = ((rc * 1.002f) * (shrink * shrink)) * 0x42000000p0f;
    const float ex = r.dx * invx, ey = r.dy * invy, ez = r.dz * invz;
    const float ax = __builtin_fabsf(invx) * (Kf * Gx), ay = __builtin_fabsf(invy) * (Kf * Gy), az = __builtin_fabsf(invz) * (Kf * Gz);
    const float dn_scale = 0.99987f;                                      // 1 - 2^-13
    auto lim = [&](float e, float a, uint32_t room_up, uint32_t room_dn) {
        const float up = fma_(fmaxf(e, 0.0f) + a, Q, 0.5f), dn = fma_(fmaxf(-e, 0.0f) + a, Q, 0.5f);
        const float nu = ((float)room_up * dn_scale) * __builtin_amdgcn_rcpf(up);
        const float nd = ((float)room_dn * dn_scale) * __builtin_amdgcn_rcpf(dn);
        return fminf(nu, nd);
    };
    float n = lim(ex, ax, hx - r.px, r.px);
    n = fminf(n, lim(ey, ay, hy - r.py, r.py));
    n = fminf(n, lim(ez, az, hz - r.pz, r.pz));
    n = fminf(n, 1.0e6f);
    const uint32_t steps = (uint32_t)n + 1u;                              // positions before steps 0 .. floor(n) are inside
    const bool inside = r.px <= hx && r.py <= hy && r.pz <= hz;           // holds by construction of the brick; cheap to assert
    // window bounds as floats: bit patterns in [lo, lo + span) are positive finite floats, ordered like their patterns
    window_ok = ok && win.span != 0u && dot * 0.55f >= __uint_as_float(win.lo) && dot * 1.6f < __uint_as_float(win.lo + win.span - 1u);
    return (ok && inside && n >= 0.0f) ? steps : 1u;
}

// Staging: q = n / d for n < 2^10 and a wave-uniform 2 <= d <= 2^10 as (n * m) >> 22 with m = 2^22 / d + (0, 3]: exact
// because n * (m - 2^22 / d) < 2^22 / d for every such pair (brute-forced on the host for every d, n and every float
// evaluation error of m up to +-0.75; the rcp + multiply + add below err by less than 0.5), and n * m < 2^32.
__device__ __forceinline__ uint32_t stage_magic(uint32_t d)
{
    return sgpr((uint32_t)(4194304.0f * __builtin_amdgcn_rcpf((float)d) + 2.0f));
}
// one voxel through a raw buffer descriptor over the brick's origin: scalar base + 32-bit byte offset per lane (no 64-bit
// address arithmetic in vector registers)
template <typename V> __device__ __forceinline__ V load_vox_buf(__amdgpu_buffer_rsrc_t rsrc, uint32_t off);
template <> __device__ __forceinline__ vox_f32 load_vox_buf<vox_f32>(__amdgpu_buffer_rsrc_t rsrc, uint32_t off)
{
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
    vox_f32 v;
    v.x = __uint_as_float(a.x); v.y = __uint_as_float(a.y); v.z = __uint_as_float(a.z); v.w = __uint_as_float(a.w);
    return v;
}
template <> __device__ __forceinline__ vox_i16 load_vox_buf<vox_i16>(__amdgpu_buffer_rsrc_t rsrc, uint32_t off)
{
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    const u2 a = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)off, 0, 0);
    vox_i16 v;
    v.x = (int16_t)(a.x & 0xFFFFu); v.y = (int16_t)(a.x >> 16); v.z = (int16_t)(a.y & 0xFFFFu); v.w = (int16_t)(a.y >> 16);
    return v;
}
__device__ __forceinline__ int32_t w_bits(const vox_f32 &v) { return __float_as_int(v.w); }
__device__ __forceinline__ int32_t w_bits(const vox_i16 &v) { return v.w; }

// The LDS brick always holds fp32 voxels: an int16 field is converted once while it is staged
// (exact: (float)int16), not 32 times per step.
__device__ __forceinline__ vox_f32 to_f32_vox(const vox_f32 &v) { return v; }
__device__ __forceinline__ vox_f32 to_f32_vox(const vox_i16 &v)
{
    vox_f32 o;
    o.x = (float)v.x; o.y = (float)v.y; o.z = (float)v.z; o.w = (float)v.w;
    return o;
}

// fp32 bricks are stored PRESCALED: x, y, z carry the interpolation's final factor 2^-48 (.cu:154), so the step
// does not multiply three results by it.  Scaling by a power of two commutes with every rounding of the nested lerps as
// long as no intermediate of the scaled computation is subnormal.  With all non-zero inputs at least 2^-55 in magnitude
// every intermediate of the unscaled computation is a multiple of 2^-78 (products with the integer weights and their
// sums keep that grid, and rounding a multiple of 2^-78 to 24 bits keeps it too), hence zero or at least 2^-78 and, scaled,
// zero or at least 2^-126: normal.  prescale_ok() is that input condition (plus an upper bound against overflow); a brick that fails it is not staged
// (the round gathers from global memory, unscaled), which is what keeps the results bit-identical.
__device__ __forceinline__ bool prescale_ok(const vox_f32 &v)
{
    // ... and below 2^78: the unscaled computation multiplies by weights of up to 2^48 in total before its final 2^-48, so
    // a larger value overflows to infinity there (and must here too); NaN and infinity fail the test as well.  Such bricks
    // take the unscaled global-gather path.
    auto ok = [](float a) { const uint32_t b = __float_as_uint(a) & 0x7FFFFFFFu; return b == 0u || (b >= ((127u - 55u) << 23) && b < ((127u + 78u) << 23)); };
    return ok(v.x) && ok(v.y) && ok(v.z);
}
__device__ __forceinline__ bool prescale_ok(const vox_i16 &) { return true; }   // |v| >= 1 or 0
__device__ __forceinline__ vox_f32 prescaled(vox_f32 v)
{
    const float s = 1.0f / 0x1000000000000p0f;
    v.x *= s; v.y *= s; v.z *= s;                          // w stays raw: only its sign is ever used
    return v;
}
// Bricks hold fp32 voxels for both field types: an int16 field is converted (exactly) while it is staged, once per
// voxel and round instead of 32 conversions per step.  Measured on config 5 (int16 field) at the 10 KiB segment:
// fp32 bricks 14.9 ms, int16 bricks of twice the voxel count 15.2 ms; at equal bytes the conversions cost more than
// the smaller brick (development builds with -DVRT_SLAB_LDS_FIELD keep the field's own type in LDS for comparison).
#ifdef VRT_SLAB_LDS_FIELD
template <typename FieldT> struct LdsVox { using type = typename FieldTraits<FieldT>::vox; };
__device__ __forceinline__ vox_f32 to_lds_vox(const vox_f32 &v) { return prescaled(v); }
__device__ __forceinline__ vox_i16 to_lds_vox(const vox_i16 &v) { return v; }
#else
template <typename FieldT> struct LdsVox { using type = vox_f32; };
template <typename V> __device__ __forceinline__ vox_f32 to_lds_vox(const V &v) { return prescaled(to_f32_vox(v)); }
#endif

// LDS addresses as 32-bit integers (so that they can be pinned in registers ahead of the reads without losing the
// address space: a generic pointer would turn ds_read into flat_load)
template <typename LV> __device__ __forceinline__ uint32_t lds_addr(const LV *seg, uint32_t idx)
{
    return (uint32_t)(size_t)(__attribute__((address_space(3))) const LV *)seg + idx * (uint32_t)sizeof(LV);
}
template <typename LV> __device__ __forceinline__ __attribute__((address_space(3))) const LV *lds_at(uint32_t a)
{
    return (__attribute__((address_space(3))) const LV *)(size_t)a;
}

template <typename LV>
__device__ __forceinline__ void gather_lds(const LV *seg, uint32_t base, uint32_t sy, uint32_t sx, f4 (&c)[8])
{
    const LV *b = seg + base;
    LV v[8];
    v[0] = b[0];       v[1] = b[1];
    v[2] = b[sy];      v[3] = b[sy + 1];
    v[4] = b[sx];      v[5] = b[sx + 1];
    v[6] = b[sx + sy]; v[7] = b[sx + sy + 1];
#pragma unroll
    for (int k = 0; k < 8; ++k) cvt_vox(v[k], c[k]);
}

// the same from a byte address inside the wavefront's segment and byte strides (mad_hi16 builds the address)
template <typename LV>
__device__ __forceinline__ void gather_lds_at(uint32_t a0, uint32_t sy16, uint32_t sx16, f4 (&c)[8])
{
    const uint32_t a1 = a0 + sy16, a2 = a0 + sx16, a3 = a2 + sy16;
    LV v[8];
    v[0] = lds_at<LV>(a0)[0]; v[1] = lds_at<LV>(a0)[1]; v[2] = lds_at<LV>(a1)[0]; v[3] = lds_at<LV>(a1)[1];
    v[4] = lds_at<LV>(a2)[0]; v[5] = lds_at<LV>(a2)[1]; v[6] = lds_at<LV>(a3)[0]; v[7] = lds_at<LV>(a3)[1];
#pragma unroll
    for (int k = 0; k < 8; ++k) cvt_vox(v[k], c[k]);
}

// VPL = voxels staged per lane at most; the wave's LDS segment holds 64*VPL fp32 voxels (16 B each).
// OCC = waves per SIMD the register allocator must allow (launch-bounds second argument).
// SCHED: 0 static (one wavefront per group of 64 slots), 1 persistent (wavefronts draw whole groups from a counter),
// 2 persistent and time-sliced (wavefronts draw (slice, group) tickets: a group marches slice_steps wave-steps, parks its
// ray state in global memory and is resumed later by whichever wavefront draws its next ticket).
// FAST: the opt-in tolerance step (vrt_march.h, ilen_fast / add_round_fast) in every loop; instantiated for the default geometry only.
template <typename FieldT, typename DirT, int VPL, int OCC, bool ABSORB, int SCHED = 0, bool FAST = false>
__global__ __launch_bounds__(64 * kSlabWavesPerBlock, OCC) void k_march_slab(const MarchParams p)
{
    constexpr bool PERSIST = SCHED != 0;
    constexpr bool SLICED = SCHED == 2;
    using vox = typename FieldTraits<FieldT>::vox;
    constexpr uint32_t kCap = 64u * VPL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t wave = threadIdx.x >> 6;
    using lvox = typename LdsVox<FieldT>::type;
    constexpr bool kPrescaled = sizeof(lvox) == 16;     // fp32 bricks carry the 2^-48 (see prescale_ok)
    lvox *seg = (lvox *)smem + (size_t)wave * kCap;

    const uint32_t lane = threadIdx.x & 63u;
    const vox *f = (const vox *)p.field;
    const uint32_t B[3] = {p.Bx, p.By, p.Bz};
    const uint32_t bx1 = (p.Bx - 1) & 0xFFFFu, by1 = (p.By - 1) & 0xFFFFu, bz1 = (p.Bz - 1) & 0xFFFFu;
    const float invx = p.invx, invy = p.invy, invz = p.invz;
    const StepWindow win = {p.win_lo, p.win_span};

    // Per-ray state.  `it` is the reference's remaining-iterations counter (.cu:333-351): it drops by one per
    // completed step and the ray stops at it == 0, on leaving the field, or on w > 0; the value at
    // that moment is raydata._iterations.  A ray stopped by the w > 0 test parks that value in
    // `rem_w` and zeroes `it`, so "it > 0 and inside the field" is the only liveness predicate.
    Ray r;
    uint32_t it, rem_w, light;                          // light: brightness, ABSORB only (.cu:332)
    // (the ray's index is not kept across the march: it is looked up again from the group's first slot when the ray is
    // stored -- three registers fewer live through the loops, which the 80-register instantiations were spilling)
#if defined(VRT_SLAB_TIMELINE) || defined(VRT_SLAB_PHASES)   // development builds (tools/timeline.py, tools/phases.py)
    uint32_t tl_lds = 0, tl_dir = 0, tl_rounds = 0;
#define VRT_TL(x) (x)
#else
#define VRT_TL(x) ((void)0)
#endif
#ifdef VRT_SLAB_PHASES
    unsigned long long ph_acc[kPhCount] = {0, 0, 0, 0, 0};
    const unsigned long long ph_t0 = ph_stamp(), ph_r0 = ph_realtime();
    unsigned long long ph_t = ph_t0;
    unsigned long long tl_lds_total = 0, tl_dir_total = 0, tl_rounds_total = 0;
    uint32_t sl_tickets = 0, sl_wasted = 0, sl_polls = 0, sl_waits = 0;
#define VRT_SL(x) (x)
#else
#define VRT_SL(x) ((void)0)
#endif

    // march(): advances the wavefront's rays until all of them have stopped
    uint32_t budget = 0xFFFFFFFFu;                      // wave-steps the current slice may still run (SLICED; wave-uniform)
    auto march = [&]() {
    uint32_t plan_round = 0;                            // wave-uniform
    int last_rung = kSlabMaxK >> 3;                     // rung of the K ladder the previous round ended on (wave-uniform)
    int32_t sweep_neg[3] = {0, 0, 0}, sweep_pos[3] = {0, 0, 0};
    for (;;) {
        const bool live = it > 0 && in_field(r, bx1, by1, bz1);
        VRT_PH(kPhOther);
        if (!any_lane(live)) return;             // wave early-out: every lane has terminated
        if constexpr (SLICED) { if (budget == 0u) return; }      // slice used up: the caller parks the group

        // ---- plan -------------------------------------------------------------------------
        // A wave reduction costs about 50 cycles (five DPP stages with their hazard slots, two readlanes), twelve
        // of them more than a whole march step.  The six position bounds are needed for every brick; the six
        // displacement bounds change slowly (the direction moves by g * inv per step), so they are refreshed every
        // VRT_SWEEP_REFRESH-th round only -- a stale bound can only make the brick-exit test end a round early.
        PlanIn in;
        if ((plan_round++ % VRT_SWEEP_REFRESH) == 0) {
            int32_t s[3];
            float ilen_lane;
            {                                           // per-step displacement estimate from the current direction
                const float dot = fma_(r.dz, r.dz, fma_(r.dx, r.dx, r.dy * r.dy));
                const float ilen = 0x42000000p0f * __builtin_amdgcn_rcpf(dot);   // estimate only
                ilen_lane = ilen;
                const float lim = 268435456.0f;         // 2^28
                s[0] = (int32_t)fminf(fmaxf((r.dx * invx) * ilen, -lim), lim);
                s[1] = (int32_t)fminf(fmaxf((r.dy * invy) * ilen, -lim), lim);
                s[2] = (int32_t)fminf(fmaxf((r.dz * invz) * ilen, -lim), lim);
            }
            uint32_t red[7];                            // min(s_d) and max(s_d) as unsigned minima: bias by 2^31, complement for max
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                red[2 * d] = live ? (uint32_t)s[d] ^ 0x80000000u : 0xFFFFFFFFu;
                red[2 * d + 1] = live ? ~((uint32_t)s[d] ^ 0x80000000u) : 0xFFFFFFFFu;
            }
            // the largest C / |dir|^2 of the wave (positive floats order like their bit patterns; NaN / negative -> ignored)
            red[6] = live && ilen_lane > 0.0f && ilen_lane < 3.0e38f ? ~__float_as_uint(ilen_lane) : 0xFFFFFFFFu;
            wave_min_u32(red);
            // Bending allowance: the direction changes by at most gstep_d per step, so after j steps the displacement per
            // step differs from today's by up to j * gstep_d * C / |dir|^2.  Averaged over a round of about 2 * last_rung
            // steps that is last_rung * gstep_d * ilen per step, on either side.  It makes the planned brick wide enough for
            // more of the round to be PROVEN (safe_steps): config 3, proven share of the LDS steps 50 % -> 58 %, rounds cut short by
            // a ray leaving its brick 316 k -> 156 k per launch; run time unchanged within 0.5 % (the tested steps it saves cost
            // about what the 4 % extra rounds of the slightly wider bricks do).
            const float ilen_max = red[6] == 0xFFFFFFFFu ? 0.0f : __uint_as_float(~red[6]);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
#ifdef VRT_SLAB_NO_BEND
                const int32_t bend = 0;
#else
                const float bf = fminf((float)last_rung * p.gstep[d] * ilen_max * 1.25f, 1048576.0f);
                const int32_t bend = bf > 0.0f ? (int32_t)bf : 0;         // NaN -> 0
#endif
                const int32_t smn = (int32_t)(red[2 * d] ^ 0x80000000u) - bend;
                const int32_t smx = (int32_t)(~red[2 * d + 1] ^ 0x80000000u) + bend;
                int32_t a = smn >> 8, c = (smx >> 8) + 1;          // floor / ceil (+1 keeps it conservative)
                a = a > 0 ? 0 : (a < -(1 << 20) ? -(1 << 20) : a);
                c = c < 0 ? 0 : (c > (1 << 20) ? (1 << 20) : c);
                sweep_neg[d] = a;
                sweep_pos[d] = c;
            }
        }
        {
            const uint32_t px[3] = {r.px, r.py, r.pz};
            uint32_t red[6];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                red[2 * d] = live ? px[d] : 0xFFFFFFFFu;
                red[2 * d + 1] = live ? ~px[d] : 0xFFFFFFFFu;
            }
            wave_min_u32(red);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const uint32_t lo = red[2 * d];
                const uint32_t hi = ~red[2 * d + 1];
                in.pmin8[d] = (int32_t)(lo >> 8);
                in.pmax8[d] = (int32_t)((hi >> 8) + 1);
                in.sneg8[d] = sweep_neg[d];
                in.spos8[d] = sweep_pos[d];
            }
        }
        // largest K of the ladder kSlabMaxK, /2, /4 ... 1 whose brick fits (the volume grows with K).  The search starts
        // at the rung the previous round of this group ended on -- it rarely moves by more than one -- and walks up or
        // down from there: the same K as a search from the top, in two or three evaluations instead of four or five.
        Brick bk;
        int K = last_rung;
        uint32_t vol = plan_brick(in, K, B, kCap, bk);
        if (vol <= kCap) {
            while (K < kSlabMaxK) {
                Brick b2;
                const int Kn = (K << 1) > kSlabMaxK ? kSlabMaxK : (K << 1);
                const uint32_t v2 = plan_brick(in, Kn, B, kCap, b2);
                if (v2 > kCap) break;
                K = Kn; vol = v2; bk = b2;
            }
        } else {
            while (vol > kCap && K > 1) {
                K >>= 1;
                vol = plan_brick(in, K, B, kCap, bk);
            }
        }
        last_rung = K;
        if (vol <= kCap && K >= 2 && K < kSlabMaxK) {   // the halving search is coarse: try 1.5 K
            Brick b2;
            const int K2 = K + (K >> 1);
            const uint32_t v2 = plan_brick(in, K2, B, kCap, b2);
            if (v2 <= kCap) { K = K2; vol = v2; bk = b2; }
        }

        VRT_PH(kPhPlan);
        const uint32_t ey = bk.e[1], ez = bk.e[2];
        bool brick_has_w = false;
        bool staged = false;
        // ---- stage: brick -> LDS ----------------------------------------------------------
        if (!(vol > kCap || K < kSlabMinK || bk.e[0] < 2 || bk.e[1] < 2 || bk.e[2] < 2)) {
            // voxel v of the brick -> (lx, ly, lz): exact division of v < 2^10 by the wave-uniform extents through
            // (v * m) >> 22, two full-rate instructions (stage_magic); the voxel's offset from the brick origin as a
            // 32-bit byte offset beside a scalar base whenever the field is small enough for that (off32)
            const uint32_t mz = stage_magic(ez), my = stage_magic(ey);
            const uint32_t gsy = p.Bz, gsx = p.By * p.Bz;             // gsx is only used when off32 holds (then < 2^24)
            const vox *g0 = f + ((size_t)bk.lo[0] * p.By + bk.lo[1]) * p.Bz + bk.lo[2];
            const bool off32 = (unsigned long long)(kCap / 4u) * p.By * p.Bz * sizeof(vox) < (1ull << 32);   // lx < kCap / 4
            vox tmp[VPL];
            const uint32_t vlast = vol - 1u;
            if (off32) {
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)g0, 0, 0xFFFFFFFF, 0x00020000);   // raw, no stride
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    const uint32_t v = lane + 64u * j;
                    const uint32_t vc = v < vlast ? v : vlast;            // clamp: lanes past the end re-read the last voxel
                    const uint32_t t = div22(vc, mz);                     // vc / ez
                    const uint32_t lz = nmad24(t, ez, vc);
                    const uint32_t lx = div22(t, my);                     // t / ey
                    const uint32_t ly = nmad24(lx, ey, t);
                    tmp[j] = load_vox_buf<vox>(rsrc, mad24(lx, gsx, mad24(ly, gsy, lz)) * (uint32_t)sizeof(vox));
                }
            } else {
                const size_t gsy64 = p.Bz, gsx64 = (size_t)p.By * p.Bz;
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    const uint32_t v = lane + 64u * j;
                    const uint32_t vc = v < vlast ? v : vlast;
                    const uint32_t t = div22(vc, mz);
                    const uint32_t lz = nmad24(t, ez, vc);
                    const uint32_t lx = div22(t, my);
                    const uint32_t ly = nmad24(lx, ey, t);
                    tmp[j] = g0[(size_t)lx * gsx64 + (size_t)ly * gsy64 + lz];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");    // previous round's LDS reads are done
            // w > 0 somewhere in the brick: w values are integers (never NaN, never -0.0), so the sign test is an integer
            // maximum of the bit patterns; lanes past the end re-read the last voxel: harmless
            int32_t wmax = 0;
            bool tiny = false;
#pragma unroll
            for (int j = 0; j < VPL; ++j) wmax = max(wmax, w_bits(tmp[j]));
            // fp32 bricks carry the interpolation's 2^-48 (three multiplications less per step); a brick holding a value
            // too small for that to be exact (prescale_ok) is not staged at all: the round gathers from global memory.
            // The engine scans the field once (engine_field_absmax): when every voxel of it passes, no brick is tested.
            if constexpr (kPrescaled) {
                if (!(p.field_flags & kFieldPrescaleOk)) {
#pragma unroll
                    for (int j = 0; j < VPL; ++j) tiny |= !prescale_ok(tmp[j]);
                }
            }
            staged = !kPrescaled || !any_lane(tiny);
            if (staged) {
#pragma unroll
                for (int j = 0; j < VPL; ++j) seg[lane + 64u * j] = to_lds_vox(tmp[j]);   // past the brick's end: copies of its last
                                                                                        // voxel into the segment's unused rest
                brick_has_w = any_lane(wmax > 0);                 // wave-uniform: any opaque-ish voxel staged?
                stat_add(p, kStatRounds, 1);
                VRT_TL(++tl_rounds);
                stat_add(p, kStatPlannedK, (unsigned long long)K);
                stat_add(p, kStatVoxStaged, vol);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        VRT_PH(kPhStage);

        if (!staged) {
            // ---- incoherent wave (or a brick that cannot be staged exactly): gather from global memory for a few
            // steps, then re-plan ----
            stat_add(p, kStatDirectRounds, 1);
            VRT_TL(++tl_rounds);
#ifdef VRT_SLAB_STATS
            stat_add(p, 8, bk.e[0]); stat_add(p, 9, bk.e[1]); stat_add(p, 10, bk.e[2]);      // fallback brick extents (K = 1)
            stat_add(p, 11, (unsigned long long)__popcll(__ballot(live)));                    // live lanes at fallback
            stat_add(p, 12, (unsigned long long)(p.iterations - 1 - sgpr(__ockl_wfred_max_u32(live ? it : 0u))));  // steps done (min over lanes)
            {   // would the live lanes fit TWO (or four) bricks?  split at the middle of the widest (two widest) bounding-box axes
                int ext[3] = {in.pmax8[0] - in.pmin8[0], in.pmax8[1] - in.pmin8[1], in.pmax8[2] - in.pmin8[2]};
                int a0 = ext[0] >= ext[1] ? (ext[0] >= ext[2] ? 0 : 2) : (ext[1] >= ext[2] ? 1 : 2);
                int a1 = a0 == 0 ? (ext[1] >= ext[2] ? 1 : 2) : (a0 == 1 ? (ext[0] >= ext[2] ? 0 : 2) : (ext[0] >= ext[1] ? 0 : 1));
                const uint32_t pp[3] = {r.px, r.py, r.pz};
                const uint32_t mid0 = (uint32_t)(in.pmin8[a0] + in.pmax8[a0]) << 7, mid1 = (uint32_t)(in.pmin8[a1] + in.pmax8[a1]) << 7;
                const uint32_t pa0 = a0 == 0 ? pp[0] : (a0 == 1 ? pp[1] : pp[2]), pa1 = a1 == 0 ? pp[0] : (a1 == 1 ? pp[1] : pp[2]);
                const uint32_t cl = (pa0 >= mid0 ? 1u : 0u) | (pa1 >= mid1 ? 2u : 0u);
                for (int K2 = 8; K2 <= 16; K2 += 8) {
                    uint32_t v2 = 0, v4 = 0;
                    for (uint32_t c = 0; c < 4; ++c) {
                        for (int pass = 0; pass < 2; ++pass) {           // pass 0: 2-split cluster c (c < 2), pass 1: 4-split cluster c
                            if (pass == 0 && c >= 2) continue;
                            const bool mine = live && (pass == 0 ? (cl & 1u) == c : cl == c);
                            uint32_t red[6];
                            for (int d = 0; d < 3; ++d) { red[2 * d] = mine ? pp[d] : 0xFFFFFFFFu; red[2 * d + 1] = mine ? ~pp[d] : 0xFFFFFFFFu; }
                            wave_min_u32(red);
                            if (red[0] == 0xFFFFFFFFu) continue;     // empty cluster
                            PlanIn in2 = in;
                            for (int d = 0; d < 3; ++d) { in2.pmin8[d] = (int32_t)(red[2 * d] >> 8); in2.pmax8[d] = (int32_t)((~red[2 * d + 1] >> 8) + 1); }
                            Brick b2;
                            const uint32_t vv = plan_brick(in2, K2, B, kCap, b2);
                            if (pass == 0) v2 += vv; else v4 += vv;
                        }
                    }
                    stat_add(p, K2 == 8 ? 14 : 16, v2 <= kCap ? 1 : 0);
                    stat_add(p, K2 == 8 ? 15 : 17, v4 <= kCap ? 1 : 0);
                }
            }
#endif
            // Tried in round 3 and dropped: LANE WINDOWS for int16 fields -- every lane stages a private window of 5 layers along
            // the wavefront's dominant axis x the 2 x 2 voxels around its ray (20 raw voxels = its 160 bytes of the segment, laid
            // out so that the reads are conflict-free) and marches out of it: one memory round trip per window instead of one
            // per step, and private windows cannot blow up the way a common brick does.  Bit-exact, but config 5 13.0 -> 28.4 -
            // 32.5 ms: the rays that end up in these rounds are the strongly deflected ones, they cross a lateral cell boundary
            // every 3 - 4 steps, so a window lasts a few steps and costs 20 scattered loads per lane (global-gather step 2 190 ->
            // 5 750 wall cycles, and the staging of everybody else's bricks slowed from 3 980 to 9 750 per round behind them).
#if VRT_PRIO_DIRECT
            __builtin_amdgcn_s_setprio(VRT_PRIO_DIRECT);
#endif
            // A ray advances about a quarter voxel per step, so it reads the same eight corners for several
            // steps in a row: the corners stay in registers and only lanes that entered a new cell gather again
            // (a quarter of the lanes per step).  At 4 waves/SIMD (128 registers) the eight CONVERTED corners
            // (32 registers) stay live for either field type: config 3 (fp32 field) 8.15 -> 7.31 ms -- its global-gather
            // rounds were 8 % of the steps but 16 % of a wavefront's life.  With 80 registers (6 waves/SIMD) they do not
            // fit (fp32 corners spilled: +27 % on config 3, and still at 5 waves/SIMD), so those kernels gather every step.
            // (Rounds 1 and 2 kept an int16 field's RAW corners, 16 registers, in the 80-register kernels: it cost 20 - 44
            // bytes of scratch per lane and, measured in round 3 on 512^3 int16 with 2048^2 rays, was 0.9 % slower than
            // gathering every step -- 21.97 vs 21.78 ms; 96 registers at 5 waves/SIMD: 22.46 -- so it is gone.)
#ifdef VRT_NO_KEEP_F32
            constexpr bool kKeepF32 = OCC < 0;                  // (development: A/B)
#else
            constexpr bool kKeepF32 = OCC <= 4;
#endif
            // Tried in round 3 and dropped (MI355X, kernel ms, config 5 / config 3; base 13.3 / 7.0): (a) FACE SHIFT -- a lane that
            // moves one cell forward along the wavefront's dominant axis keeps the far face as the new near face and gathers only
            // four corners (half the cache lines; six compile-time copies of this loop, one per axis and sign): 13.9 - 14.3 /
            // 7.29, global-gather step 2 190 -> 2 510 wall cycles -- fewer lines do not help; (b) the same with the face BEYOND
            // the cell prefetched into 16 (fp32) / 8 (int16 raw) more registers, so that a forward move needs no memory round
            // trip at all: 16.6 / 7.9, step 4 420 cycles, and 172 - 200 B of scratch in the fp32 kernels.
            f4 keptf[kKeepF32 ? 8 : 1];
            uint32_t have_xy = 0xFFFFFFFFu, have_z = 0xFFFFFFFFu;   // cell of the corners in keptf[]: (x | y << 16, z); none yet
            const int direct_steps = SLICED ? (int)(budget < (uint32_t)kSlabDirectSteps ? budget : (uint32_t)kSlabDirectSteps) : kSlabDirectSteps;
            for (int k = 0; k < direct_steps; ++k) {
                const bool alive = it > 0 && in_field(r, bx1, by1, bz1);
                if (!any_lane(alive)) break;
                if constexpr (SLICED) --budget;
                stat_add(p, kStatDirectWaveSteps, 1);
                VRT_TL(++tl_dir);
                if (alive) {
                    bool dark = false;
                    if constexpr (ABSORB) dark = absorb_step(p, r, light);
                    if (dark) { rem_w = it; it = 0; }
                    else {
                        const uint32_t xy = (r.px >> 16) | (r.py & 0xFFFF0000u), z = r.pz >> 16;
                        if constexpr (kKeepF32) {
                            if (xy != have_xy || z != have_z) {
                                gather_global<FieldT>(f, r, p.By, p.Bz, keptf);
                                have_xy = xy;
                                have_z = z;
                            }
                            if (march_step<false, FAST>(r, keptf, invx, invy, invz, win)) { rem_w = it; it = 0; } else --it;
                        } else {
                            f4 c[8];
                            gather_global<FieldT>(f, r, p.By, p.Bz, c);
                            if (march_step<false, FAST>(r, c, invx, invy, invz, win)) { rem_w = it; it = 0; } else --it;
                        }
                    }
                }
            }
#if VRT_PRIO_DIRECT
            __builtin_amdgcn_s_setprio(0);
#endif
            VRT_PH(kPhDirect);
            continue;
        }

        // ---- march: up to K steps out of LDS ------------------------------------------------
        {
            const uint32_t sy = ez, sx = ey * ez;
            const uint32_t ex1 = bk.e[0] - 1, ey1 = ey - 1, ez1 = ez - 1;
            const uint32_t lox = bk.lo[0], loy = bk.lo[1], loz = bk.lo[2];
            // two copies of the loop: in a brick without any positive w (the common case) the opacity test
            // cannot fire, so that copy carries neither the test nor the rem_w bookkeeping
            // (a slice that ends inside the round simply runs fewer steps of it: the brick planned for K steps holds them)
            const int Ku = (int)sgpr(SLICED && budget < (uint32_t)K ? budget : (uint32_t)K);   // keep the trip count in an SGPR
            // inside the round the positions are kept relative to the brick origin (its low 16 bits are zero, so
            // the weights and every position update are unchanged): the cell is then one shift, not an SDWA subtract
            const uint32_t ox = lox << 16, oy = loy << 16, oz = loz << 16;
            r.px -= ox; r.py -= oy; r.pz -= oz;
            // byte strides and base of the brick in LDS: a cell's address is three v_mad_u32_u16 on the positions' high halves
            const uint32_t sy16 = sy * (uint32_t)sizeof(lvox), sx16 = sx * (uint32_t)sizeof(lvox), seg_addr = lds_addr(seg, 0);
            auto run = [&](auto has_w) {
                constexpr bool kHasW = decltype(has_w)::value;
                constexpr bool kDies = kHasW || ABSORB;              // a ray can stop inside the round for another reason
                // Without the opacity test a ray can stop inside a round only by leaving the brick or by using up its
                // iterations; the round is capped at the smallest remaining count, so "has iterations left" is a
                // per-round constant there and costs no compare per step (absorption mode can darken a ray mid-round: kDies).
                // (lanes that have left the field hold it > 0 but never march again: they are outside the clipped brick.
                // They are excluded here, or a departed lane with a small count would cap every later round.)
                const bool pos0 = live;
                int Kr = Ku;
                if constexpr (!kDies) {
#ifdef VRT_SLAB_NO_PROOF
                    uint32_t red1[1] = {pos0 ? it : 0xFFFFFFFFu};
                    wave_min_u32(red1);
                    const uint32_t min_it = red1[0];
                    Kr = (uint32_t)Ku < min_it ? Ku : (int)min_it;
#else
                    // proven part of the round: no lane can leave the brick during Kg steps (safe_steps), so the loop
                    // carries no brick test, no ballot and no exit -- lanes not live at the start of the round stay masked
                    bool win_mine = true;
                    const uint32_t mine = pos0 ? safe_steps(r, invx, invy, invz, p.gstep[0], p.gstep[1], p.gstep[2], p.gstep_norm, Ku,
                                                             (ex1 << 16) - 1u, (ey1 << 16) - 1u, (ez1 << 16) - 1u, win, win_mine)
                                               : 0xFFFFFFFFu;
                    const bool win_all = !any_lane(pos0 && !win_mine);
                    uint32_t red2[2] = {pos0 ? it : 0xFFFFFFFFu, mine};
                    wave_min_u32(red2);                    // smallest remaining iteration count, smallest proven run
                    const uint32_t min_it = red2[0];
                    Kr = (uint32_t)Ku < min_it ? Ku : (int)min_it;
                    uint32_t Kg = red2[1];
                    Kg = Kg < (uint32_t)Kr ? Kg : (uint32_t)Kr;
                    if (Kg >= (uint32_t)kSlabMinSafe) {
                        stat_add(p, kStatProvenRounds, 1);
                        stat_add(p, kStatProvenSteps, Kg);
                        stat_add(p, kStatLdsWaveSteps, Kg);
                        VRT_TL(tl_lds += Kg);
                        if (pos0) {
                            auto proven = [&](auto window_test, auto unit_inv) {
                                for (uint32_t k = 0; k < Kg; ++k) {
                                    // everything the eight reads need (four addresses) and everything that would otherwise be
                                    // computed into a register one of them is still writing (the weights) comes first, so the
                                    // reads issue back to back and nothing waits on a write-after-read of a load in flight
                                    // (issuing the reads before the weights, which then cover part of the LDS latency, was
                                    // measured too: the compiler then waits for all eight reads at once; +2 % on config 4)
                                    // brick-relative cells are the positions' high halves; byte strides of a brick fit 16 bits
                                    uint32_t a0 = mad_hi16(r.pz, (uint32_t)sizeof(lvox), mad_hi16(r.py, sy16, mad_hi16(r.px, sx16, seg_addr)));
                                    uint32_t a1 = a0 + sy16, a2 = a0 + sx16, a3 = a2 + sy16;
                                    Weights w = lerp_weights(r);
                                    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(w.wrx), "+v"(w.wlx), "+v"(w.wry),
                                                 "+v"(w.wly), "+v"(w.wrz), "+v"(w.wlz) :: "memory");
                                    lvox v[8];
                                    v[0] = lds_at<lvox>(a0)[0]; v[1] = lds_at<lvox>(a0)[1]; v[2] = lds_at<lvox>(a1)[0]; v[3] = lds_at<lvox>(a1)[1];
                                    v[4] = lds_at<lvox>(a2)[0]; v[5] = lds_at<lvox>(a2)[1]; v[6] = lds_at<lvox>(a3)[0]; v[7] = lds_at<lvox>(a3)[1];
                                    f4 c[8];
#pragma unroll
                                    for (int q = 0; q < 8; ++q) cvt_vox(v[q], c[q]);
                                    march_step_now<kPrescaled, decltype(window_test)::value, decltype(unit_inv)::value, FAST>(r, c, w, invx, invy, invz, win);
                                }
                            };
#ifdef VRT_SLAB_NO_UNIT_INV
                            const bool unit_inv = false;
#else
                            const bool unit_inv = invx == 1.0f && invy == 1.0f && invz == 1.0f;      // wave-uniform (kernel arguments)
#endif
                            if (win_all) { if (unit_inv) proven(std::false_type{}, std::true_type{}); else proven(std::false_type{}, std::false_type{}); }
                            else proven(std::true_type{}, std::false_type{});
                            it -= Kg;
                        }
                        Kr -= (int)Kg;                   // whatever the plan allowed beyond the proof runs with the tests below
                        if constexpr (SLICED) budget -= Kg;
                    }
#endif
                }
                const uint64_t m_pos0 = lane_mask(pos0);
                // brick-relative positions: "cell < extent - 1" is a compare of the position itself, the cell index never forms
                const uint32_t ex1s = ex1 << 16, ey1s = ey1 << 16, ez1s = ez1 << 16;
                for (int k = 0; k < Kr; ++k) {
                    const bool inb = r.px < ex1s && r.py < ey1s && r.pz < ez1s;   // cell and its +1 neighbours are staged
                    const bool pos = kDies ? it > 0 : pos0;
                    // wave-level tests on the compare masks themselves (scalar ALU): a ballot of a compound
                    // bool would be materialised through v_cndmask + v_cmp
                    const uint64_t m_pos = kDies ? lane_mask(it > 0) : m_pos0;
                    const uint64_t m_inb = lane_mask(r.px < ex1s) & lane_mask(r.py < ey1s) & lane_mask(r.pz < ez1s);
                    if ((m_pos & ~m_inb) != 0ull) {                      // rare: some lane is outside the brick
                        // the brick is clipped to the field, so a ray that left the field is outside the brick too
                        // and simply stops marching; a ray still inside the field forces a new plan
                        const bool inf = ((r.px + ox) >> 16) < bx1 && ((r.py + oy) >> 16) < by1 && ((r.pz + oz) >> 16) < bz1;
                        if (any_lane(pos && !inb && inf)) { stat_add(p, kStatEarlyExit, 1); break; }
                    }
                    const bool alive = pos && inb;                       // inside the brick => inside the field
                    if ((m_pos & m_inb) == 0ull) break;
                    stat_add(p, kStatLdsWaveSteps, 1);
                    VRT_TL(++tl_lds);
                    if constexpr (SLICED) --budget;
                        if (alive) {
                        bool dark = false;
                        if constexpr (ABSORB) {
                            Ray ra = r;
                            ra.px += ox; ra.py += oy; ra.pz += oz;
                            dark = absorb_step(p, ra, light);
                        }
                        if (dark) { rem_w = it; it = 0; }
                        else {
                            f4 c[8];
                            gather_lds_at<lvox>(mad_hi16(r.pz, (uint32_t)sizeof(lvox), mad_hi16(r.py, sy16, mad_hi16(r.px, sx16, seg_addr))), sy16, sx16, c);
                            if constexpr (kHasW) {
                                if (march_step<kPrescaled, FAST>(r, c, invx, invy, invz, win, true)) { rem_w = it; it = 0; } else --it;
                            } else {
                                (void)march_step<kPrescaled, FAST>(r, c, invx, invy, invz, win, false);
                                --it;
                            }
                        }
                    }
                }
            };
            if (brick_has_w) run(std::true_type{}); else run(std::false_type{});
            r.px += ox; r.py += oy; r.pz += oz;
        }
        VRT_PH(kPhLds);
    }
    };  // march

#ifdef VRT_SLAB_PHASES
    auto ph_flush = [&]() {                              // once per wavefront, after its last group
        VRT_PH(kPhOther);
        const unsigned long long r1 = ph_realtime();
        if (p.stats && lane == 0) {
            for (int i = 0; i < kPhCount; ++i) atomicAdd(p.stats + 16 + i, ph_acc[i]);
            atomicAdd(p.stats + 24, 1ull);                                    // wavefronts
            atomicAdd(p.stats + 25, ph_t - ph_t0);                            // shader cycles (s_memtime)
            atomicAdd(p.stats + 26, r1 - ph_r0);                              // 100 MHz ticks (s_memrealtime)
            atomicAdd(p.stats + 27, (unsigned long long)tl_lds_total);
            atomicAdd(p.stats + 28, (unsigned long long)tl_dir_total);
            atomicAdd(p.stats + 29, (unsigned long long)tl_rounds_total);
            atomicAdd(p.stats + 30, (unsigned long long)sl_tickets);
            atomicAdd(p.stats + 31, (unsigned long long)sl_wasted);
            atomicAdd(p.stats + 32, (unsigned long long)sl_polls);
            atomicAdd(p.stats + 33, (unsigned long long)sl_waits);
        }
    };
#endif
    auto begin_group = [&](uint32_t first_slot) {        // fill_struct (.cu:468-487): fresh rays from the caller's arrays
        const uint32_t slot = first_slot + lane;
        const bool have = slot < p.nrays;
        const uint32_t ray = have ? (p.order ? p.order[slot] : slot) : 0u;
        r.px = r.py = r.pz = 0; r.dx = r.dy = r.dz = 1.0f;
        if (have) load_ray<DirT>(p, ray, r);
        it = have ? p.iterations - 1 : 0u;
        rem_w = 0xFFFFFFFFu;
        light = 0xFFFFFFFFu;
    };
    auto end_group = [&](uint32_t first_slot) {          // read_struct (.cu:490-516, 952-956): results to the ray's own index
        const uint32_t slot = first_slot + lane;
        if (slot < p.nrays) store_ray<DirT>(p, p.order ? p.order[slot] : slot, r, rem_w != 0xFFFFFFFFu ? rem_w : it, light);
    };

    // wave 0 of block 0 notes the shader clock (s_memtime) against the 100 MHz wall clock (s_memrealtime) around its
    // whole life: the clock the kernel really ran at (bench.py prices the VALU-issue roofline with it)
    // (the begin stamps are stored at once: kept in registers until the end they were four VGPRs live across the whole
    // kernel, which the 80-register instantiations spilled to scratch)
    const bool stamps = p.clock_out != nullptr && blockIdx.x == 0 && threadIdx.x < 64;
    if (stamps) {
        unsigned long long t0, r0;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
        if (lane == 0) { p.clock_out[0] = t0; p.clock_out[1] = r0; }
    }
    auto clock_stamp_end = [&]() {
        if (stamps) {
            unsigned long long t1, r1;
            asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
            if (lane == 0) { p.clock_out[2] = t1; p.clock_out[3] = r1; }
        }
    };

    if constexpr (!PERSIST) {
        // one wavefront per group of 64 slots; the hardware dispatcher hands out groups in launch order
        const uint32_t first = sgpr(launch_slot(p) & ~63u);
        const uint32_t slot = first + lane;
        const bool have = slot < p.nrays;
        begin_group(first);
#ifdef VRT_SLAB_TIMELINE
        if (p.stats && lane == 0) p.stats[64 + 4 * (size_t)(slot >> 6)] = wall_clock64();
#endif
        march();
        end_group(first);
        (void)have;
#ifdef VRT_SLAB_PHASES
        tl_lds_total += tl_lds; tl_dir_total += tl_dir; tl_rounds_total += tl_rounds;
        ph_flush();
#endif
#ifdef VRT_SLAB_TIMELINE
        {
            const uint32_t done = have ? p.iterations - (rem_w != 0xFFFFFFFFu ? rem_w : it) : 0u;
            const uint32_t mx = sgpr(__ockl_wfred_max_u32(done));
            if (p.stats && lane == 0) {
                unsigned long long *q = p.stats + 64 + 4 * (size_t)(slot >> 6);
                q[1] = wall_clock64();
                q[2] = (unsigned long long)tl_lds | ((unsigned long long)tl_dir << 32);
                q[3] = (unsigned long long)tl_rounds | ((unsigned long long)mx << 32);
            }
        }
#endif
    } else if constexpr (SLICED) {
        // Time-sliced persistent wavefronts.  With whole groups as the unit of work a launch of G groups on S slots runs
        // G / S "rounds" of unequal groups and ends with the GPU draining for as long as the last-started groups take
        // (tools/timeline.py: 7 .. 13 % of configs 3 and 5 with less than half of the slots busy).  Here the unit is one
        // SLICE of a group: slice_steps wave-steps.  Work items come from one FIFO: items 0 .. G-1 are the groups in launch
        // order (fresh rays); a group that still has live rays after its slice parks their state in global memory and
        // appends itself (ring[tail++ mod R] = group + 1, R >= 2 G: a group is running or queued, never queued twice, so at
        // most G words are occupied); a wavefront that draws item i >= G waits until its ring word has been written, empties
        // it, takes over the parked state and marches the next slice.  (A producer that finds its word still occupied -- the
        // consumer of R items ago has drawn its item but not emptied the word yet -- retries; compare-and-swap makes that exact.)  All groups advance together, so they end
        // together, and a group resumes on whichever SIMD is free.  An item is only ever appended by a wavefront that is
        // running, so whoever waits, waits for somebody who makes progress; the waits are bounded all the same and raise
        // ctr[2] instead of hanging.  ctr[0] head, ctr[1] groups finished, ctr[2] error, ctr[3] tail (items appended).
        // Counters and queue words are touched by device-scope atomics only: they execute at the coherence point, whereas
        // a load -- sc1 or not -- may be served a stale line by this XCD's L2 for as long as the line stays there.
        const uint32_t ngroups = (p.nrays + 63u) >> 6;
        uint32_t *ctr = p.group_counter;
        // Slice length.  Cutting every group into equal slices pays for the balance with a park / resume and a cold start of
        // the planner per cut; what the tail of the launch needs is only SOME short items to run last.  A slice a little
        // longer than the mean group (the pre-march's estimate, summed by k_wave_work) cuts just the long groups, and their
        // remainders are those items: config 5 13.3 ms at a fixed 1024, 13.5 at 3500, 12.9 at 4096 (mean group 3730 wave-steps),
        // 13.4 at 5000, 13.55 with whole groups.
        uint32_t slice = p.slice_steps;
        if (slice == 0u) {
            slice = 1024u;
            if (p.work_sum) {
                const unsigned long long mean = *p.work_sum * (unsigned long long)p.work_unit / (ngroups ? ngroups : 1u);
                const unsigned long long want = mean * p.slice_percent / 100u;
                slice = (uint32_t)(want < 256ull ? 256ull : (want > (1ull << 24) ? (1ull << 24) : want));
            }
            slice = sgpr(slice);
        }
        for (;;) {
            const uint32_t item = lane0_atomic_add(ctr + kCtrHead, 1u);
            VRT_SL(++sl_tickets);
            uint32_t g;
            if (item < ngroups) {
                g = p.wave_order ? sgpr(p.wave_order[item]) : item;
                begin_group(g * 64u);
            } else {
                uint32_t *qe = p.slice_flags + ((item - ngroups) & (p.ticket_limit - 1u));     // ring of ticket_limit = 2^k words
                // take the word and empty it for its next lap in ONE exchange: with a read followed by a second atomic, the
                // holder of the ticket one lap later could read the same word in between and resume the group a second time
#ifdef VRT_OLD_CONSUMER                                             // (development: A/B of the round-2 read-then-add consumer)
#define VRT_TAKE(qe) lane0_atomic_add(qe, 0u)
#else
#define VRT_TAKE(qe) lane0_atomic_swap(qe, 0u)
#endif
                uint32_t q = VRT_TAKE(qe);
                VRT_SL(sl_waits += q == 0u ? 1u : 0u);
                bool finished = false;
                for (uint32_t spin = 0; q == 0u; ++spin) {
                    // nothing appended yet: either every group has finished (leave) or a running group will append
                    if ((spin & 15u) == 0u) {
                        if (lane0_atomic_add(ctr + kCtrDone, 0u) >= ngroups) { finished = true; break; }   // (or somebody gave up: raise_sched_error)
                    }
                    if (spin >= p.spin_limit) { raise_sched_error(ctr, p.sched_err); finished = true; break; }   // never hang
                    __builtin_amdgcn_s_sleep(127);
                    q = VRT_TAKE(qe);
                    VRT_SL(++sl_polls);
                }
                if (finished) break;
#ifdef VRT_OLD_CONSUMER
                (void)lane0_atomic_add(qe, 0u - q);
#endif
#undef VRT_TAKE
                g = q - 1u;
                const u32x4 *park = (const u32x4 *)p.park + (size_t)g * 128u + lane;      // [group][2][64 lanes] x 16 B
                u32x4 a, b;
                load_sc1_2xb128(park, park + 64, a, b);
                r.px = a.x; r.py = a.y; r.pz = a.z; it = a.w;
                r.dx = __uint_as_float(b.x); r.dy = __uint_as_float(b.y); r.dz = __uint_as_float(b.z); rem_w = b.w;
                light = 0xFFFFFFFFu;
            }
            budget = slice;
            march();
            const bool live_left = any_lane(it > 0 && in_field(r, bx1, by1, bz1));
            if (live_left) {
                u32x4 *park = (u32x4 *)p.park + (size_t)g * 128u + lane;
                u32x4 a, b;
                a.x = r.px; a.y = r.py; a.z = r.pz; a.w = it;
                b.x = __float_as_uint(r.dx); b.y = __float_as_uint(r.dy); b.z = __float_as_uint(r.dz); b.w = rem_w;
                store_sc1_b128(park, a);
                store_sc1_b128(park + 64, b);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the state is out before the group is queued
                const uint32_t at = lane0_atomic_add(ctr + kCtrTail, 1u);
                uint32_t *qw = p.slice_flags + (at & (p.ticket_limit - 1u));
                uint32_t spin = 0;
                while (lane0_atomic_cas(qw, 0u, g + 1u) != 0u) {             // 0 -> g + 1; occupied only in the race described above
                    if (++spin >= p.spin_limit) { raise_sched_error(ctr, p.sched_err); break; }   // never hang
                    __builtin_amdgcn_s_sleep(16);
                }
            } else {
                end_group(g * 64u);
                (void)lane0_atomic_add(ctr + kCtrDone, 1u);
            }
#ifdef VRT_SLAB_PHASES
            tl_lds_total += tl_lds; tl_dir_total += tl_dir; tl_rounds_total += tl_rounds;
            tl_lds = tl_dir = tl_rounds = 0;
#endif
        }
#ifdef VRT_SLAB_PHASES
        ph_flush();
#endif
    } else {
        // Persistent wavefronts: the grid is what the GPU holds at once and every wavefront draws the next
        // group from a counter as soon as it has finished one -- no slot idles until the other three
        // wavefronts of a workgroup are done, and the launch order (longest first) is followed exactly.
        // Every wavefront leaves as soon as the counter runs out; nobody waits for anybody.
        const uint32_t ngroups = (p.nrays + 63u) >> 6;
        for (;;) {
            const uint32_t fr = lane0_atomic_add(p.group_counter, 1u);
            if (fr >= ngroups) break;
            const uint32_t first = (p.wave_order ? sgpr(p.wave_order[fr]) : fr) * 64u;
            const uint32_t slot = first + lane;
            const bool have = slot < p.nrays;
            begin_group(first);
#ifdef VRT_SLAB_TIMELINE
            if (p.stats && lane == 0) p.stats[64 + 4 * (size_t)(slot >> 6)] = wall_clock64();
            tl_lds = tl_dir = tl_rounds = 0;
#endif
            march();
            end_group(first);
            (void)have; (void)slot;
#ifdef VRT_SLAB_PHASES
            tl_lds_total += tl_lds; tl_dir_total += tl_dir; tl_rounds_total += tl_rounds;
            tl_lds = tl_dir = tl_rounds = 0;
#endif
#ifdef VRT_SLAB_TIMELINE
            {
                const uint32_t done = have ? p.iterations - (rem_w != 0xFFFFFFFFu ? rem_w : it) : 0u;
                const uint32_t mx = sgpr(__ockl_wfred_max_u32(done));
                if (p.stats && lane == 0) {
                    unsigned long long *q = p.stats + 64 + 4 * (size_t)(slot >> 6);
                    q[1] = wall_clock64();
                    q[2] = (unsigned long long)tl_lds | ((unsigned long long)tl_dir << 32);
                    q[3] = (unsigned long long)tl_rounds | ((unsigned long long)mx << 32);
                }
            }
#endif
        }
#ifdef VRT_SLAB_PHASES
        ph_flush();
#endif
    }
    clock_stamp_end();
#undef VRT_TL
}

// ---- launch ----------------------------------------------------------------------------------------------------
// The LDS segment per wavefront fixes the brick capacity, the wavefronts a CU holds (160 KiB LDS) and with them the
// register budget.  Two sizes are complete (persistent and static schedule, absorption mode):
//   10 KiB  640 voxels, 4 wavefronts / SIMD, 128 VGPRs   the default: longer rounds, fewer rounds that fit no brick
//    6 KiB  384 voxels, 6 wavefronts / SIMD,  80 VGPRs   dense ray sets (more than kDenseRaysPerColumn rays per voxel column of the
//                                                        largest face): tiles are narrow, bricks deep, occupancy wins
// measured (kernel ms, 6 / 8 / 10 KiB): config 3 (4 rays per column) 8.58 / 8.41 / 8.32, config 4 (16 per column)
// 25.9 / 26.8 / 26.9, config 5 (1 per column) 15.98 / 15.37 / 14.96.  The other sizes exist for vrt_options.slab_lds_bytes
// (static schedule only).
struct SlabGeometry { uint32_t seg_bytes; int occ; bool complete; };
constexpr uint32_t kDenseRaysPerColumn = 8;
inline SlabGeometry slab_geometry(uint32_t want, uint32_t nrays, uint32_t Bx, uint32_t By, uint32_t Bz)
{
    if (want == 0) {
        const uint64_t bmax = Bx > By ? (Bx > Bz ? Bx : Bz) : (By > Bz ? By : Bz);
        want = (uint64_t)nrays > (uint64_t)kDenseRaysPerColumn * bmax * bmax ? 6144u : 10240u;
    }
#ifdef VRT_SLAB_EXPERIMENT
    const bool all = true;
#else
    const bool all = false;
#endif
    if (want <= 6144) return {6144, 6, true};
    if (want <= 8192) return {8192, 5, all};
    if (want <= 10240) return {10240, 4, true};
    if (want <= 12288) return {12288, 3, all};
    return {16384, 2, all};
}

template <typename FieldT, typename DirT, int BYTES, int OCC, bool ABSORB, bool COMPLETE, bool FAST = false>
hipError_t launch_slab_bytes(const MarchParams &p, int num_cu, hipStream_t s)
{
    constexpr int VPL = BYTES / (64 * (int)sizeof(typename LdsVox<FieldT>::type));
    const unsigned threads = 64 * kSlabWavesPerBlock;
    const unsigned groups_blocks = (p.nrays + threads - 1) / threads;
    const size_t lds = (size_t)BYTES * kSlabWavesPerBlock;
    if constexpr (COMPLETE) if (p.group_counter) {
        // persistent grid: the wavefronts the GPU holds at once (OCC per SIMD = OCC blocks of 4 wavefronts per CU)
        const unsigned resident = (unsigned)(num_cu > 0 ? num_cu : 256) * (unsigned)OCC;
        const unsigned grid = groups_blocks < resident ? groups_blocks : resident;
        if constexpr (!ABSORB) if (p.park) {
            k_march_slab<FieldT, DirT, VPL, OCC, ABSORB, 2, FAST><<<grid, threads, lds, s>>>(p);
            return hipGetLastError();
        }
        k_march_slab<FieldT, DirT, VPL, OCC, ABSORB, 1, FAST><<<grid, threads, lds, s>>>(p);
        return hipGetLastError();
    }
    k_march_slab<FieldT, DirT, VPL, OCC, ABSORB, 0, FAST><<<groups_blocks, threads, lds, s>>>(p);
    return hipGetLastError();
}

template <typename FieldT, typename DirT>
hipError_t launch_slab(const MarchParams &p, SlabGeometry g, int num_cu, hipStream_t s)
{
#ifdef VRT_SLAB_EXPERIMENT
    constexpr bool kAll = true;
#else
    constexpr bool kAll = false;
#endif
    if (p.transl) {                                    // opt-in absorption mode: the two complete sizes only
        if (g.seg_bytes <= 6144) return launch_slab_bytes<FieldT, DirT, 6144, 6, true, true>(p, num_cu, s);
        return launch_slab_bytes<FieldT, DirT, 10240, 4, true, true>(p, num_cu, s);
    }
    // opt-in tolerance mode: the default geometry has a FAST instantiation; every other kernel computes exactly
    if (p.fast_math && g.seg_bytes == 10240) return launch_slab_bytes<FieldT, DirT, 10240, 4, false, true, true>(p, num_cu, s);
    switch (g.seg_bytes) {
    case 6144:  return launch_slab_bytes<FieldT, DirT, 6144, 6, false, true>(p, num_cu, s);
    case 8192:  return launch_slab_bytes<FieldT, DirT, 8192, 5, false, kAll>(p, num_cu, s);
    case 10240: return launch_slab_bytes<FieldT, DirT, 10240, 4, false, true>(p, num_cu, s);
    case 12288: return launch_slab_bytes<FieldT, DirT, 12288, 3, false, kAll>(p, num_cu, s);
    default:    return launch_slab_bytes<FieldT, DirT, 16384, 2, false, kAll>(p, num_cu, s);
    }
}

}  // namespace vrt
