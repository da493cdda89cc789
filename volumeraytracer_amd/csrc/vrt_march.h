// vrt_march.h -- device-side ray march for gfx950 (CDNA4), exact-arithmetic core.
//
// Restates, for the GPU, the live semantics of the reference's per-ray loop
//   trace_ray_function   /root/reference/src/cuda_volume_raytracer.cu:317-374
//   interpolatef (3-D)   /root/reference/src/cuda_volume_raytracer.cu:130-155
//   __float2int_rn2      /root/reference/src/tuple_math.h:270-278  (host: std::round)
// with the floating-point operation order of the reference's *CPU* build (g++ -O2 -mfma
// contracts a*b + c*d to fma(a, b, c*d)); every fma below is explicit and this file is
// compiled with -ffp-contract=off, so results are bit-identical to that build.
//
// Not a translation of the CUDA kernel (.cu:397-414, one thread per ray over AoS raydata_t):
// rays are consumed from the caller's SoA arrays through an optional tile-order
// permutation, the field is read as 16-byte (fp32) / 8-byte (int16) voxels, and the LDS
// slab kernel below lets a 64-lane wavefront stage the voxel brick it is about to walk.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vrt {

struct MarchParams {
    const void *field;        // AoS {dx,dy,dz,w} per voxel, index ((x*By)+y)*Bz+z
    uint32_t Bx, By, Bz;      // field extents
    float invx, invy, invz;   // invscale
    uint32_t win_lo, win_span; // fast-path window of |dir|^2 as float bit patterns, see StepWindow / step_window()
    const uint32_t *start_pos;
    const void *start_dir;
    uint32_t *end_pos;
    void *end_dir;
    uint32_t *end_iter;
    uint32_t *remaining_light;
    uint32_t *path;           // nullptr unless PATH
    const uint32_t *order;    // nullptr = identity; else ray processed by slot i is order[i]
    const uint32_t *wave_order; // nullptr = identity; else the g-th launched wavefront takes slots 64*wave_order[g] ..+63
    uint32_t iterations;
    uint32_t nrays;
    uint32_t slab_lds_bytes;  // slab kernel: LDS bytes per wave
    unsigned long long *stats; // optional debug counters (nullptr in production), see vrt_slab.h
    const uint32_t *transl;   // ABSORB kernels only: translucency per field voxel (opt-in mode, .cu:337-341)
    uint32_t min_brightness;  // ABSORB kernels only
    uint32_t *group_counter;  // persistent slab kernel only: next group of 64 slots to start (zeroed per launch);
                              // time-sliced schedule: [0] items drawn, [1] groups finished, [2] error word, [3] items appended
    void *park;               // time-sliced schedule only: parked ray state, 2 KiB per group ([group][2][64 lanes] x 16 B)
    uint32_t *slice_flags;    // time-sliced schedule only: the FIFO of re-queued groups (group + 1; 0 = not written yet)
    uint32_t slice_steps;     // time-sliced schedule only: wave-steps per slice; 0 = derived in the kernel from the mean work
                              // estimate of the launch's wavefronts (work_sum, work_unit, slice_percent)
    const unsigned long long *work_sum;   // sum over the launch's wavefronts of the pre-march's estimate (vrt_sort.h, k_wave_work), or null
    uint32_t work_unit;       // fine steps per unit of that estimate (kCoarse)
    uint32_t slice_percent;   // slice = mean estimated wave-steps per wavefront x this / 100
    uint32_t ticket_limit;    // time-sliced schedule only: words in the FIFO ring (a power of two >= 2 x groups)
    uint32_t spin_limit;      // time-sliced schedule only: polls after which a waiting wavefront gives up and raises the error words
    uint32_t *sched_err;      // time-sliced schedule only: error word in pinned host memory (device address), written on that path only
    uint32_t out_pos_bias;    // added to end_pos / path on output: 0 at the seam, 0x10000 for the scene
                              // layer's "+0x10000" epilogue (image_util.cpp:770-771)
    unsigned long long *clock_out; // optional: 4 words {s_memtime, s_memrealtime} x {begin, end} of one wavefront (wave 0 of block 0)
    uint32_t fast_math;       // opt-in tolerance mode (vrt_options.fast_math): the 10 KiB slab kernel takes the FAST step (below)
    uint32_t field_flags;     // kFieldPrescaleOk: every voxel of the field passes prescale_ok (vrt_slab.h), bricks need no test
    float gstep_norm;         // slab kernel: upper bound of the LENGTH of the change of dir in one step (longest gradient vector x max |inv_d|)
    float gstep[3];           // slab kernel: upper bound of |g_d * inv_d|, the change of dir_d in one step (.cu:344-345):
                              // max |field channel d| over the whole field x |inv_d|, rounded up (see safe_steps, vrt_slab.h)
};

constexpr uint32_t kFieldPrescaleOk = 1u;

struct f4 { float x, y, z, w; };

// ---- exact helpers -----------------------------------------------------------------------

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// a*wl + b*wr as the reference object computes it: fma(a, wl, b*wr)
__device__ __forceinline__ float lerp_(float a, float wl, float b, float wr) { return fma_(a, wl, b * wr); }

// std::round (half away from zero) followed by x86 cvttss2si (out of range / NaN -> INT_MIN)
__device__ __forceinline__ int32_t round_to_i32(float x)
{
    float t = __builtin_truncf(x);
    float d = x - t;                                   // exact
    if (__builtin_fabsf(d) >= 0.5f) t += __builtin_copysignf(1.0f, x);
    int32_t r = (int32_t)t;
    if (!(__builtin_fabsf(t) < 2147483648.0f)) r = (int32_t)0x80000000;
    return r;
}

// 0x42000000p0f / d without the 11-instruction IEEE sequence: v_rcp_f32 (1 ulp) and one fma
// refinement of the quotient.  tools/ubench/div_exact.hip and vrt_selftest_ilen2() compare this against
// the IEEE divide for EVERY mantissa of every exponent: it is bit-identical for biased exponents 31..252
// of d (below, the quotient overflows; above, the reciprocal is denormal).  div_window() admits 32..250.
__device__ __forceinline__ bool div_window(float d)
{
    return (__float_as_uint(d) - (32u << 23)) < ((251u - 32u) << 23);   // sign bit set / NaN / inf wrap out of range
}
__device__ __forceinline__ float ilen_short(float d)
{
    const float r = __builtin_amdgcn_rcpf(d);
    const float q = 0x42000000p0f * r;
    return fma_(fma_(-d, q, 0x42000000p0f), r, q);
}

// pos + roundf(s) for |s| < 2^31: v_cvt_rpi_i32_f32 is floor(x + 0.5) evaluated exactly (vrt_selftest_round() checks every
// float below 2^31 on the caller's GPU), which on |s| is round-half-away-from-zero of the magnitude; the sign goes back
// in as a two's-complement negate folded into the add: (m ^ sg) + pos - sg with sg = 0 or -1.
__device__ __forceinline__ uint32_t rpi_abs(float s)
{
    uint32_t m;
    asm("v_cvt_rpi_i32_f32 %0, |%1|" : "=v"(m) : "v"(s));
    return m;
}
__device__ __forceinline__ uint32_t add_round(uint32_t pos, float s)
{
    const uint32_t m = rpi_abs(s);
    const uint32_t sg = (uint32_t)((int32_t)__float_as_uint(s) >> 31);
    uint32_t t;
    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(t) : "v"(m), "v"(sg), "v"(pos));
    return t - sg;
}

// Opt-in tolerance mode (vrt_options.fast_math; north_star allows 1e-5 relative): inside the step window the step length comes
// from v_rcp_f32 alone (1 ulp, no refinement: -2 instructions) and the position update is pos + floor(s + 1/2) in one
// conversion (v_cvt_rpi_i32_f32 on the signed value: it differs from the reference's round-half-away only on exact negative
// ties; -2 instructions per component).  8 of the proven step's 79 vector instructions.  The interpolation itself is NOT
// touched: fma(a, wl, b * wr) and fma(t, b - a, a) are two instructions each, there is nothing to save there.
__device__ __forceinline__ float ilen_fast(float d) { return 0x42000000p0f * __builtin_amdgcn_rcpf(d); }
__device__ __forceinline__ uint32_t add_round_fast(uint32_t pos, float s)
{
    uint32_t m;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(m) : "v"(s));
    return pos + m;
}

// ---- voxel pair loads: voxel z and z+1 at one (x,y), as 2 x f4 --------------------------

typedef float   vox_f32 __attribute__((ext_vector_type(4)));   // 16 B: {dx, dy, dz, w}
typedef int16_t vox_i16 __attribute__((ext_vector_type(4)));   //  8 B

template <typename FieldT> struct FieldTraits;
template <> struct FieldTraits<float>   { using vox = vox_f32; static constexpr int kVoxBytes = 16; };
template <> struct FieldTraits<int16_t> { using vox = vox_i16; static constexpr int kVoxBytes = 8; };

__device__ __forceinline__ void cvt_vox(const vox_f32 &v, f4 &o) { o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w; }
__device__ __forceinline__ void cvt_vox(const vox_i16 &v, f4 &o)
{
    o.x = (float)v.x; o.y = (float)v.y; o.z = (float)v.z; o.w = (float)v.w;
}

// ---- one march step given the 8 corners ------------------------------------------------
// c[k]: k = 4*hi_x + 2*hi_y + hi_z.  Returns true if the ray terminates on the w>0 test.

// Fast-path window of dot = |dir|^2, as float bit patterns: the step takes the short division and the integer
// rounding iff (bits(dot) - lo) < span (unsigned; NaN, inf, negative values wrap out).  Two conditions are folded into it:
//   (1) the exhaustively verified range of the short division, biased exponents 32..250 (div_window);
//   (2) |s| < 2^30 for the three step components s_d = (dir_d * inv_d) * (C / dot), inside the range add_round is exact
//       on (|s| < 2^31): |dir_d| <= sqrt(dot), so 2 |s_d| <= 2C * max|inv| / sqrt(dot) < 2^31  <=>  dot > (2C * max|inv| / 2^31)^2.
//       The host (step_window) rounds that bound up by 2^-16, far more than the few 2^-24 the float evaluation of
//       dot, C/dot and the two products can add, so no per-step range test remains.
struct StepWindow { uint32_t lo, span; };

struct Ray {
    uint32_t px, py, pz;
    float dx, dy, dz;        // internal scale, unit 1.0 == 65536 (.cu:328-331)
};

// check_w == false promises that none of the eight corners has w > 0 (the slab kernel knows this for a
// whole brick from a ballot taken while staging it), so the opacity test cannot fire and is skipped.
// PRESCALED: the x, y, z channels of the corners already carry the interpolation's factor 2^-48 (the slab kernel
// scales them once while staging a brick, exactly -- see stage_scale_ok in vrt_slab.h); the w channel is always raw.
template <bool PRESCALED = false, bool FAST = false>
__device__ __forceinline__ bool march_step(Ray &r, const f4 (&c)[8], float invx, float invy, float invz,
                                           StepWindow win, bool check_w = true)
{
    // wl = 0x10000 - wr is formed in float: both are integers below 2^17, so the subtraction is exact
    // and equals the reference's (float)(0x10000 - multr) (.cu:145-153)
    const float wrx = (float)(r.px & 0xFFFFu), wlx = 65536.0f - wrx;
    const float wry = (float)(r.py & 0xFFFFu), wly = 65536.0f - wry;
    const float wrz = (float)(r.pz & 0xFFFFu), wlz = 65536.0f - wrz;

    // opacity channel (.cu:343): only the sign of the interpolated w matters.  If all eight
    // corner values are <= 0 every product and fma below is <= 0; if all are > 0 the result is
    // > 0 (the (0,0,0) corner has weight >= 1).  Only mixed neighbourhoods need the arithmetic.
    // The w values are integers stored as floats (engine-built: never NaN, never -0.0), so a value is
    // > 0.0f exactly when its bit pattern is > 0 as int32: "some corner positive" is an integer max > 0
    // and "all corners positive" an integer min > 0 -- no float canonicalisation needed.
    const int32_t wi[8] = {__float_as_int(c[0].w), __float_as_int(c[1].w), __float_as_int(c[2].w), __float_as_int(c[3].w),
                           __float_as_int(c[4].w), __float_as_int(c[5].w), __float_as_int(c[6].w), __float_as_int(c[7].w)};
    const int32_t wmax = check_w ? max(max(max(wi[0], wi[1]), max(wi[2], wi[3])), max(max(wi[4], wi[5]), max(wi[6], wi[7]))) : 0;
    if (wmax > 0) {                                   // some corner has w > 0.0f (bit pattern of +0.0 is 0)
        const int32_t wmin = min(min(min(wi[0], wi[1]), min(wi[2], wi[3])), min(min(wi[4], wi[5]), min(wi[6], wi[7])));
        if (wmin > 0) return true;
        float a00 = lerp_(c[0].w, wlx, c[4].w, wrx), a01 = lerp_(c[1].w, wlx, c[5].w, wrx);
        float a10 = lerp_(c[2].w, wlx, c[6].w, wrx), a11 = lerp_(c[3].w, wlx, c[7].w, wrx);
        float b0 = lerp_(a00, wly, a10, wry), b1 = lerp_(a01, wly, a11, wry);
        float w = lerp_(b0, wlz, b1, wrz) * (1.0f / 0x1000000000000p0f);
        if (w > 0.0f) return true;
    }

    float g[3];
    {
        float a00, a01, a10, a11, b0, b1;
#define VRT_CH(ch, out)                                                                         \
        a00 = lerp_(c[0].ch, wlx, c[4].ch, wrx); a01 = lerp_(c[1].ch, wlx, c[5].ch, wrx);        \
        a10 = lerp_(c[2].ch, wlx, c[6].ch, wrx); a11 = lerp_(c[3].ch, wlx, c[7].ch, wrx);        \
        b0 = lerp_(a00, wly, a10, wry); b1 = lerp_(a01, wly, a11, wry);                          \
        out = lerp_(b0, wlz, b1, wrz);                                                           \
        if constexpr (!PRESCALED) out *= (1.0f / 0x1000000000000p0f);
        VRT_CH(x, g[0]) VRT_CH(y, g[1]) VRT_CH(z, g[2])
#undef VRT_CH
    }
    r.dx = fma_(g[0], invx, r.dx);                          // .cu:344-345
    r.dy = fma_(g[1], invy, r.dy);
    r.dz = fma_(g[2], invz, r.dz);
    const float dot = fma_(r.dz, r.dz, fma_(r.dx, r.dx, r.dy * r.dy));
    // The step s = (dir*inv)*ilen, ilen = 0x42000000p0f / dot (IEEE divide, .cu:346-347), is needed only
    // through roundf(s): inside the window the short division and add_round (v_cvt_rpi_i32_f32 on |s|, both verified
    // exhaustively on the device: vrt_selftest_ilen2, vrt_selftest_round) replace the divide and std::round.
    if (!((__float_as_uint(dot) - win.lo) < win.span)) {    // outside the window: everything the long way
        const float ilen = 0x42000000p0f / dot;             // NaN / inf / out of range: x86 cvttss2si semantics
        r.px += (uint32_t)round_to_i32((r.dx * invx) * ilen);
        r.py += (uint32_t)round_to_i32((r.dy * invy) * ilen);
        r.pz += (uint32_t)round_to_i32((r.dz * invz) * ilen);
        return false;
    }
    if constexpr (FAST) {
        const float ilen = ilen_fast(dot);
        r.px = add_round_fast(r.px, (r.dx * invx) * ilen);
        r.py = add_round_fast(r.py, (r.dy * invy) * ilen);
        r.pz = add_round_fast(r.pz, (r.dz * invz) * ilen);
        return false;
    }
    const float ilen = ilen_short(dot);                     // == 0x42000000p0f / dot, proven exhaustively
    r.px = add_round(r.px, (r.dx * invx) * ilen);           // |s| < 2^30 inside the window (StepWindow, condition 2)
    r.py = add_round(r.py, (r.dy * invy) * ilen);
    r.pz = add_round(r.pz, (r.dz * invz) * ilen);
    return false;
}

// The same step for loops that carry no opacity test (check_w == false), written so that the common case has no
// two-sided divergent region: every lane takes the short division and the integer rounding, and lanes whose |dir|^2
// lies outside the fast-path window (none, normally: one untaken-in-effect scalar branch) are repaired the long way.
// Results are identical to march_step(..., check_w = false) lane by lane.
struct Weights { float wrx, wlx, wry, wly, wrz, wlz; };
__device__ __forceinline__ Weights lerp_weights(const Ray &r)
{
    Weights w;
    w.wrx = (float)(r.px & 0xFFFFu); w.wlx = 65536.0f - w.wrx;
    w.wry = (float)(r.py & 0xFFFFu); w.wly = 65536.0f - w.wry;
    w.wrz = (float)(r.pz & 0xFFFFu); w.wlz = 65536.0f - w.wrz;
    return w;
}
// WINDOW_TEST = false: the caller has proven that |dir|^2 stays inside the fast-path window for the whole round
// (safe_steps, vrt_slab.h), so the step carries no test at all.
// UNIT_INV = true: invscale is exactly (1, 1, 1) (every BASELINE config): fma(g, 1, d) == g + d and d * 1 == d bit for bit,
// so the three products with invscale are not computed.
template <bool PRESCALED, bool WINDOW_TEST, bool UNIT_INV = false, bool FAST = false>
__device__ __forceinline__ void march_step_now(Ray &r, const f4 (&c)[8], const Weights &w, float invx, float invy, float invz, StepWindow win)
{
    const float wrx = w.wrx, wlx = w.wlx, wry = w.wry, wly = w.wly, wrz = w.wrz, wlz = w.wlz;
    float g[3];
    {
        float a00, a01, a10, a11, b0, b1;
#define VRT_CH(ch, out)                                                                         \
        a00 = lerp_(c[0].ch, wlx, c[4].ch, wrx); a01 = lerp_(c[1].ch, wlx, c[5].ch, wrx);        \
        a10 = lerp_(c[2].ch, wlx, c[6].ch, wrx); a11 = lerp_(c[3].ch, wlx, c[7].ch, wrx);        \
        b0 = lerp_(a00, wly, a10, wry); b1 = lerp_(a01, wly, a11, wry);                          \
        out = lerp_(b0, wlz, b1, wrz);                                                           \
        if constexpr (!PRESCALED) out *= (1.0f / 0x1000000000000p0f);
        VRT_CH(x, g[0]) VRT_CH(y, g[1]) VRT_CH(z, g[2])
#undef VRT_CH
    }
    if constexpr (UNIT_INV) { r.dx = g[0] + r.dx; r.dy = g[1] + r.dy; r.dz = g[2] + r.dz; }
    else { r.dx = fma_(g[0], invx, r.dx); r.dy = fma_(g[1], invy, r.dy); r.dz = fma_(g[2], invz, r.dz); }
    const float dot = fma_(r.dz, r.dz, fma_(r.dx, r.dx, r.dy * r.dy));
    const float ex = UNIT_INV ? r.dx : r.dx * invx, ey = UNIT_INV ? r.dy : r.dy * invy, ez = UNIT_INV ? r.dz : r.dz * invz;
    const float ilen = FAST ? ilen_fast(dot) : ilen_short(dot);                     // garbage outside the window, replaced below
    uint32_t nx, ny, nz;
    if constexpr (FAST) { nx = add_round_fast(r.px, ex * ilen); ny = add_round_fast(r.py, ey * ilen); nz = add_round_fast(r.pz, ez * ilen); }
    else                { nx = add_round(r.px, ex * ilen); ny = add_round(r.py, ey * ilen); nz = add_round(r.pz, ez * ilen); }
    const bool outside = WINDOW_TEST && !((__float_as_uint(dot) - win.lo) < win.span);
    if (outside) {                                          // rare: skipped by one scalar branch when no lane needs it
        const float il = 0x42000000p0f / dot;
        nx = r.px + (uint32_t)round_to_i32(ex * il);
        ny = r.py + (uint32_t)round_to_i32(ey * il);
        nz = r.pz + (uint32_t)round_to_i32(ez * il);
    }
    r.px = nx; r.py = ny; r.pz = nz;
}

__device__ __forceinline__ bool in_field(const Ray &r, uint32_t bx1, uint32_t by1, uint32_t bz1)
{
    // uint16(pos>>16) < uint16(bounds-1), all components (.cu:335)
    return (r.px >> 16) < bx1 && (r.py >> 16) < by1 && (r.pz >> 16) < bz1;
}

// ---- ray load / store (fill_struct / read_struct, .cu:468-516, 952-956) -----------------

template <typename DirT>
__device__ __forceinline__ void load_ray(const MarchParams &p, uint32_t ray, Ray &r)
{
    const uint32_t *sp = p.start_pos + 3 * (size_t)ray;
    r.px = sp[0]; r.py = sp[1]; r.pz = sp[2];
    const DirT *sd = (const DirT *)p.start_dir + 3 * (size_t)ray;
    if constexpr (sizeof(DirT) == 4) {
        r.dx = sd[0] * 65536.0f; r.dy = sd[1] * 65536.0f; r.dz = sd[2] * 65536.0f;
    } else {
        r.dx = (float)sd[0] * 256.0f; r.dy = (float)sd[1] * 256.0f; r.dz = (float)sd[2] * 256.0f;
    }
}

// Opt-in absorption (the reference's compiled-out branch, .cu:337-341), evaluated at the ray's cell
// before the gather.  Returns true when the ray falls below the brightness cut-off.
__device__ __forceinline__ bool absorb_step(const MarchParams &p, const Ray &r, uint32_t &brightness)
{
    const size_t idx = ((size_t)((r.px >> 16) * p.By + (r.py >> 16))) * p.Bz + (r.pz >> 16);   // get_index, .cu:113
    const uint32_t dec = 0xFFFFFFFFu - p.transl[idx];
    brightness -= brightness < dec ? brightness : dec;
    return brightness < p.min_brightness;
}

template <typename DirT>
__device__ __forceinline__ void store_ray(const MarchParams &p, uint32_t ray, const Ray &r, uint32_t remaining,
                                          uint32_t light = 0xFFFFFFFFu)
{
    uint32_t *ep = p.end_pos + 3 * (size_t)ray;
    ep[0] = r.px + p.out_pos_bias; ep[1] = r.py + p.out_pos_bias; ep[2] = r.pz + p.out_pos_bias;
    DirT *ed = (DirT *)p.end_dir + 3 * (size_t)ray;
    if constexpr (sizeof(DirT) == 4) {
        ed[0] = r.dx * (1.0f / 65536.0f); ed[1] = r.dy * (1.0f / 65536.0f); ed[2] = r.dz * (1.0f / 65536.0f);
    } else {
        ed[0] = (int16_t)round_to_i32(r.dx * (1.0f / 256.0f));
        ed[1] = (int16_t)round_to_i32(r.dy * (1.0f / 256.0f));
        ed[2] = (int16_t)round_to_i32(r.dz * (1.0f / 256.0f));
    }
    p.end_iter[ray] = p.iterations - remaining;
    p.remaining_light[ray] = light;
}

__device__ __forceinline__ void path_put(uint32_t *pb, uint32_t it, const Ray &r, uint32_t bias)
{
    uint32_t *q = pb + 3 * (size_t)it;
    q[0] = r.px + bias; q[1] = r.py + bias; q[2] = r.pz + bias;
}

// slot processed by this thread: launch order of wavefronts (longest first, vrt_sort.h) x lane
__device__ __forceinline__ uint32_t launch_slot(const MarchParams &p)
{
    const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    const uint32_t nw = (p.nrays + 63u) >> 6;
    const uint32_t w = (p.wave_order && g < nw) ? p.wave_order[g] : g;
    return w * 64u + lane;
}

// =========================================================================================
// Kernel 1: DIRECT -- one lane per ray, corners gathered straight from the vector L1/L2.
// =========================================================================================

template <typename FieldT>
__device__ __forceinline__ void gather_global_raw(const typename FieldTraits<FieldT>::vox *f, const Ray &r, uint32_t By,
                                                  uint32_t Bz, typename FieldTraits<FieldT>::vox (&v)[8])
{
    const size_t idx = ((size_t)((r.px >> 16) * By + (r.py >> 16))) * Bz + (r.pz >> 16);
    const typename FieldTraits<FieldT>::vox *b = f + idx;
    const size_t sy = Bz, sx = (size_t)By * Bz;
    v[0] = b[0];       v[1] = b[1];
    v[2] = b[sy];      v[3] = b[sy + 1];
    v[4] = b[sx];      v[5] = b[sx + 1];
    v[6] = b[sx + sy]; v[7] = b[sx + sy + 1];
}

template <typename FieldT>
__device__ __forceinline__ void gather_global(const typename FieldTraits<FieldT>::vox *f, const Ray &r,
                                              uint32_t By, uint32_t Bz, f4 (&c)[8])
{
    typename FieldTraits<FieldT>::vox v[8];
    gather_global_raw<FieldT>(f, r, By, Bz, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) cvt_vox(v[k], c[k]);
}

template <typename FieldT, typename DirT, bool PATH, bool ABSORB>
__global__ __launch_bounds__(256) void k_march_direct(const MarchParams p)
{
    using vox = typename FieldTraits<FieldT>::vox;
    const uint32_t slot = launch_slot(p);
    if (slot >= p.nrays) return;
    const uint32_t ray = p.order ? p.order[slot] : slot;
    Ray r;
    load_ray<DirT>(p, ray, r);
    const vox *f = (const vox *)p.field;
    const uint32_t bx1 = (p.Bx - 1) & 0xFFFFu, by1 = (p.By - 1) & 0xFFFFu, bz1 = (p.Bz - 1) & 0xFFFFu;
    uint32_t *pb = PATH ? p.path + (size_t)p.iterations * 3 * ray : nullptr;

    uint32_t it = p.iterations - 1;                     // .cu:333
    if (PATH) path_put(pb, it, r, p.out_pos_bias);
    uint32_t light = 0xFFFFFFFFu;                       // .cu:332
    while (it-- > 0 && in_field(r, bx1, by1, bz1)) {    // .cu:335
        if constexpr (ABSORB) { if (absorb_step(p, r, light)) break; }   // .cu:337-341
        f4 c[8];
        gather_global<FieldT>(f, r, p.By, p.Bz, c);
        if (march_step(r, c, p.invx, p.invy, p.invz, StepWindow{p.win_lo, p.win_span})) break;
        if (PATH) path_put(pb, it, r, p.out_pos_bias);
    }
    ++it;                                               // .cu:350
    const uint32_t remaining = it;
    if (PATH) { while (it-- > 0) path_put(pb, it, r, p.out_pos_bias); } // .cu:352-358
    store_ray<DirT>(p, ray, r, remaining, light);
}

}  // namespace vrt
