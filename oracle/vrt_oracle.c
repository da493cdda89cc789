/*
 * vrt_oracle.c -- CPU restatement of the reference's trace_rays hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker / reported baseline.
 * The product path (volumeraytracer_amd/, include/vrt.h) never links or calls it.
 *
 * Parity status: PINNED.  This restatement is checked bit-for-bit against the
 * reference's own CPU build (oracle/_ref, compiled from /root/reference/src by
 * oracle/Makefile) on random scenes and on the reference's scaling_test
 * known answers (src/cuda_volume_raytracer_test.h:10-52); the golden vectors
 * under tests/golden/ were produced by that reference build
 * (tests/golden/make_golden.py).
 *
 * What is restated (all file:line relative to /root/reference/src):
 *   - TraceRaysCu ctor: w-channel + AoS interleave   cuda_volume_raytracer.cu:644-669
 *   - fill_struct / read_struct / end_iteration      cuda_volume_raytracer.cu:468-516, 952-956
 *   - trace_ray_function (the march loop)            cuda_volume_raytracer.cu:317-374
 *   - interpolatef 3-D (trilinear, lerp order x,y,z) cuda_volume_raytracer.cu:130-155
 *   - __float2int_rn == std::round                   tuple_math.h:270-278
 *   - scene front end (log, stencil, crop)           image_util.cpp:239-298, 421-463, 501-643
 *   - ray front end (range check, ior normalise)     image_util.cpp:645-772, image_util.h:348-431
 *
 * Floating-point contract: the reference object is built by g++ -O2 -mfma with
 * the default -ffp-contract=fast, so "a*b + c*d" becomes fma(a, b, c*d).  The
 * contraction pattern below was read off the disassembly of that object and
 * is written out with explicit fmaf(); this file must be compiled with
 * -ffp-contract=off so the compiler adds none of its own.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define VRT_F32 0
#define VRT_I16 1

/* x86 cvttss2si semantics for float -> int32 (what the reference binary does;
 * a C cast is UB out of range): NaN / out-of-range -> INT32_MIN. */
static inline int32_t cvt_i32(float r)
{
    if (!(r >= -2147483648.0f && r < 2147483648.0f)) return INT32_MIN;
    return (int32_t)r;
}

/* tuple_math.h:270 -- __float2int_rn on the host is std::round (half away from zero) */
static inline int32_t float2int_round(float x) { return cvt_i32(roundf(x)); }

/* ---- TraceRaysCu ctor: cuda_volume_raytracer.cu:654-669 ------------------ */

/* w = (0x7FFFFFFF - translucency) / 0x10000, int64 arithmetic, truncating division,
 * then narrowed to DiffType (.cu:656-659). */
static inline int64_t w_channel(uint32_t tr) { return ((int64_t)0x7FFFFFFF - (int64_t)tr) / 0x10000; }

void vrt_oracle_interleave_f32(const float *dx, const float *dy, const float *dz,
                               const uint32_t *transl, size_t nvox, float *out)
{
    for (size_t i = 0; i < nvox; ++i) {
        out[4 * i + 0] = dx[i];
        out[4 * i + 1] = dy[i];
        out[4 * i + 2] = dz[i];
        out[4 * i + 3] = (float)w_channel(transl[i]);
    }
}

void vrt_oracle_interleave_i16(const int16_t *dx, const int16_t *dy, const int16_t *dz,
                               const uint32_t *transl, size_t nvox, int16_t *out)
{
    for (size_t i = 0; i < nvox; ++i) {
        out[4 * i + 0] = dx[i];
        out[4 * i + 1] = dy[i];
        out[4 * i + 2] = dz[i];
        out[4 * i + 3] = (int16_t)w_channel(transl[i]);
    }
}

/* ---- interpolatef 3-D: cuda_volume_raytracer.cu:130-155 ------------------- */

static inline void load8_f32(const float *p, float v[8]) { memcpy(v, p, 32); }
static inline void load8_i16(const int16_t *p, float v[8])
{
    for (int k = 0; k < 8; ++k) v[k] = (float)p[k];
}

/* g[0..3] = trilinear({dx,dy,dz,w}) * 2^-48.  v0..v3 are the four (x,y) corners,
 * each 8 floats = voxel z (4 channels) then voxel z+1 (4 channels). */
static inline void trilinear(const float v0[8], const float v1[8], const float v2[8], const float v3[8],
                             uint32_t px, uint32_t py, uint32_t pz, float g[4])
{
    float a0[8], a1[8], b[8];
    float wr = (float)(px & 0xFFFFu), wl = (float)(0x10000u - (px & 0xFFFFu));
    for (int k = 0; k < 8; ++k) {
        a0[k] = fmaf(v0[k], wl, v2[k] * wr);     /* .cu:147 */
        a1[k] = fmaf(v1[k], wl, v3[k] * wr);     /* .cu:148 */
    }
    wr = (float)(py & 0xFFFFu); wl = (float)(0x10000u - (py & 0xFFFFu));
    for (int k = 0; k < 8; ++k) b[k] = fmaf(a0[k], wl, a1[k] * wr);   /* .cu:151 */
    wr = (float)(pz & 0xFFFFu); wl = (float)(0x10000u - (pz & 0xFFFFu));
    for (int k = 0; k < 4; ++k)
        g[k] = fmaf(b[k], wl, b[k + 4] * wr) * (1.0f / 0x1000000000000p0f);   /* .cu:154 */
}

/* ---- trace_ray_function: cuda_volume_raytracer.cu:317-374 ------------------ */

typedef struct {
    uint32_t pos[3];
    float    dir[3];     /* internal scale: unit 1.0 == 65536 */
    uint32_t remaining;  /* raydata._iterations on return */
} ray_state;

/* transl == NULL reproduces every reference call site (translucency is a DummyArray, .cu:882-938).
 * transl != NULL is the OPT-IN absorption mode: the branch at .cu:337-341 that the reference compiles
 * out.  PARITY UNPINNED for that mode -- the reference cannot execute it; this is its restatement. */
static void trace_one(int field_type, const void *field, const uint16_t bounds[3], const float inv[3],
                      ray_state *r, uint32_t iterations, uint32_t *path /* iterations*3 or NULL */,
                      const uint32_t *transl, uint32_t min_brightness, uint32_t *brightness_out)
{
    const uint32_t By = bounds[1], Bz = bounds[2];
    const uint16_t bx1 = (uint16_t)(bounds[0] - 1), by1 = (uint16_t)(bounds[1] - 1), bz1 = (uint16_t)(bounds[2] - 1);
    uint32_t px = r->pos[0], py = r->pos[1], pz = r->pos[2];
    float dx = r->dir[0], dy = r->dir[1], dz = r->dir[2];
    uint32_t it = iterations;
    uint32_t brightness = 0xFFFFFFFFu;                        /* .cu:332 */
    --it;                                                     /* .cu:333 path[--iterations] = pos */
    if (path) { path[3 * (size_t)it] = px; path[3 * (size_t)it + 1] = py; path[3 * (size_t)it + 2] = pz; }
    /* .cu:335 -- "iterations-- > 0 && all(uint16(pos>>16) < uint16(bounds-1))" */
    while (it-- > 0 && (uint16_t)(px >> 16) < bx1 && (uint16_t)(py >> 16) < by1 && (uint16_t)(pz >> 16) < bz1) {
        size_t idx = ((size_t)((px >> 16) * By + (py >> 16))) * Bz + (pz >> 16);   /* .cu:113 */
        if (transl) {                                          /* .cu:337-341 */
            uint32_t dec = 0xFFFFFFFFu - transl[idx];
            brightness -= brightness < dec ? brightness : dec;
            if (brightness < min_brightness) break;
        }
        float v0[8], v1[8], v2[8], v3[8], g[4];
        if (field_type == VRT_F32) {
            const float *f = (const float *)field + 4 * idx;
            load8_f32(f, v0);
            load8_f32(f + 4 * (size_t)Bz, v1);
            load8_f32(f + 4 * (size_t)By * Bz, v2);
            load8_f32(f + 4 * ((size_t)By * Bz + Bz), v3);
        } else {
            const int16_t *f = (const int16_t *)field + 4 * idx;
            load8_i16(f, v0);
            load8_i16(f + 4 * (size_t)Bz, v1);
            load8_i16(f + 4 * (size_t)By * Bz, v2);
            load8_i16(f + 4 * ((size_t)By * Bz + Bz), v3);
        }
        trilinear(v0, v1, v2, v3, px, py, pz, g);              /* .cu:342 */
        if (g[3] > 0.0f) break;                                /* .cu:343 */
        dx = fmaf(g[0], inv[0], dx);                           /* .cu:344-345 */
        dy = fmaf(g[1], inv[1], dy);
        dz = fmaf(g[2], inv[2], dz);
        float dot = fmaf(dz, dz, fmaf(dx, dx, dy * dy));       /* tuple_math.h:266 as contracted */
        float ilen = 0x42000000p0f / dot;                      /* .cu:346 */
        px += (uint32_t)float2int_round((dx * inv[0]) * ilen); /* .cu:347 */
        py += (uint32_t)float2int_round((dy * inv[1]) * ilen);
        pz += (uint32_t)float2int_round((dz * inv[2]) * ilen);
        if (path) { path[3 * (size_t)it] = px; path[3 * (size_t)it + 1] = py; path[3 * (size_t)it + 2] = pz; }
    }
    ++it;                                                     /* .cu:350 */
    r->remaining = it;
    if (path) {                                               /* .cu:352-358 */
        while (it-- > 0) { path[3 * (size_t)it] = px; path[3 * (size_t)it + 1] = py; path[3 * (size_t)it + 2] = pz; }
    }
    r->pos[0] = px; r->pos[1] = py; r->pos[2] = pz;
    r->dir[0] = dx; r->dir[1] = dy; r->dir[2] = dz;
    *brightness_out = brightness;                             /* .cu:370-373 (0xFFFFFFFF when transl == NULL, .cu:485) */
}

/*
 * Seam-level trace == TraceRaysCu<Diff>::trace_rays_cu<Dir> for dim 3
 * (cuda_volume_raytracer.cu:774-972): fill_struct, per-ray march, read_struct,
 * end_iteration = iterations - remaining, remaining_light = 0xFFFFFFFF
 * (translucency is a DummyArray at every call site, .cu:882-938).
 * Positions are in field coordinates (already shifted by the caller).
 * path: NULL or iterations*3*nrays uint32.  Returns 0, or -1 on bad arguments.
 */
static int trace_impl(int field_type, int dir_type, const void *field, const uint16_t bounds[3],
                      const float invscale[3], size_t nrays,
                      const uint32_t *start_pos, const void *start_dir,
                      uint32_t *end_pos, void *end_dir, uint32_t *end_iter, uint32_t *remaining_light,
                      uint32_t *path, uint32_t iterations, int nthreads,
                      const uint32_t *transl, uint32_t min_brightness)
{
    if (iterations == 0) return -1;
    if (field_type != VRT_F32 && field_type != VRT_I16) return -1;
    if (dir_type != VRT_F32 && dir_type != VRT_I16) return -1;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    (void)nthreads;
#endif
    #pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads)
    for (size_t i = 0; i < nrays; ++i) {
        ray_state r;
        r.pos[0] = start_pos[3 * i]; r.pos[1] = start_pos[3 * i + 1]; r.pos[2] = start_pos[3 * i + 2];
        if (dir_type == VRT_F32) {                            /* .cu:328-331 */
            const float *d = (const float *)start_dir + 3 * i;
            for (int k = 0; k < 3; ++k) r.dir[k] = d[k] * 65536.0f;
        } else {
            const int16_t *d = (const int16_t *)start_dir + 3 * i;
            for (int k = 0; k < 3; ++k) r.dir[k] = (float)d[k] * 256.0f;
        }
        uint32_t bright;
        trace_one(field_type, field, bounds, invscale, &r, iterations,
                  path ? path + (size_t)iterations * 3 * i : NULL, transl, min_brightness, &bright);
        end_pos[3 * i] = r.pos[0]; end_pos[3 * i + 1] = r.pos[1]; end_pos[3 * i + 2] = r.pos[2];
        if (dir_type == VRT_F32) {                            /* .cu:364-367 */
            float *d = (float *)end_dir + 3 * i;
            for (int k = 0; k < 3; ++k) d[k] = r.dir[k] * (1.0f / 65536.0f);
        } else {                                              /* .cu:359-363: roundf, then narrow */
            int16_t *d = (int16_t *)end_dir + 3 * i;
            for (int k = 0; k < 3; ++k) d[k] = (int16_t)float2int_round(r.dir[k] * (1.0f / 256.0f));
        }
        end_iter[i] = iterations - r.remaining;               /* .cu:953-956 */
        remaining_light[i] = bright;                          /* .cu:485 / .cu:370-373 */
    }
    return 0;
}

int vrt_oracle_trace(int field_type, int dir_type, const void *field, const uint16_t bounds[3],
                     const float invscale[3], size_t nrays,
                     const uint32_t *start_pos, const void *start_dir,
                     uint32_t *end_pos, void *end_dir, uint32_t *end_iter, uint32_t *remaining_light,
                     uint32_t *path, uint32_t iterations, int nthreads)
{
    return trace_impl(field_type, dir_type, field, bounds, invscale, nrays, start_pos, start_dir, end_pos, end_dir,
                      end_iter, remaining_light, path, iterations, nthreads, NULL, 0);
}

/* Opt-in absorption mode (PARITY UNPINNED, see trace_one): transl_cropped has one uint32 per field voxel. */
int vrt_oracle_trace_absorb(int field_type, int dir_type, const void *field, const uint16_t bounds[3],
                            const float invscale[3], size_t nrays,
                            const uint32_t *start_pos, const void *start_dir,
                            uint32_t *end_pos, void *end_dir, uint32_t *end_iter, uint32_t *remaining_light,
                            uint32_t *path, uint32_t iterations, int nthreads,
                            const uint32_t *transl_cropped, uint32_t min_brightness)
{
    if (!transl_cropped) return -1;
    return trace_impl(field_type, dir_type, field, bounds, invscale, nrays, start_pos, start_dir, end_pos, end_dir,
                      end_iter, remaining_light, path, iterations, nthreads, transl_cropped, min_brightness);
}

/* ========================================================================== *
 * Scene front end: image_util.cpp:501-643
 * ========================================================================== */

/* divRoundClosest, image_util.h:34-38 */
static inline int32_t div_round_closest_i32(int32_t n, int32_t d)
{
    return ((n < 0) ^ (d < 0)) ? ((n - d / 2) / d) : ((n + d / 2) / d);
}

/* standart_3d_stamp (image_util.cpp:421-425) is the x-derivative stamp with z minor:
 * value(i,j,k) = s(k) * m(i,j), s = {-1,0,1}, m = [[14,47,14],[47,162,47],[14,47,14]].
 * _stamps[2] = st (derivative along z), _stamps[0]/[1] = st with axes swapped.
 * weight = sum|st| = 812; stamp_weight = 812 * 0x100. */
static const int32_t STAMP_M[3][3] = {{14, 47, 14}, {47, 162, 47}, {14, 47, 14}};

/* offsets in stamp (i,j,k) order for derivative axis `ax`, iterated in the
 * reference's order: ascending linear stamp index j, zero entries skipped
 * (image_util.cpp:401-411).  Fills off[18] (input linear offsets) and val[18]. */
static void stamp_terms(int ax, size_t By, size_t Bz, size_t off[18], int32_t val[18])
{
    int n = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) {
                int c[3] = {i, j, k};
                int s = c[ax] - 1;            /* derivative direction */
                int a, b;                     /* the two smoothing coords */
                if (ax == 0) { a = j; b = k; } else if (ax == 1) { a = i; b = k; } else { a = i; b = j; }
                /* swap_dimensions(st, ., ax, 2) maps m's index pair accordingly; m is
                 * symmetric under the swaps used, so m[a][b] is the coefficient */
                int32_t v = s * STAMP_M[a][b];
                if (v != 0) { off[n] = ((size_t)i * By + j) * Bz + k; val[n] = v; ++n; }
            }
}

/*
 * RaytraceScene<float,float,float> ctor, image_util.cpp:576-643 (3-D only):
 *   translucency_cropped = crop by 1 voxel per side           :594
 *   ior_log[i] = (float)(log((double)ior[i]) * 0x420000)                         :611
 *   diff[ax]   = valid 3x3x3 convolution, float accumulation in the reference's
 *                term order, then sum /= (812 * 256.0f)       :239-298, 441-463
 * out bounds = bounds - 2.  diff arrays have (Bx-2)(By-2)(Bz-2) entries.
 * Returns 0; -2 if some ior <= 0 (":605 refraction-index underflow").
 */
int vrt_oracle_scene_build_f32(const size_t bounds[3], const float *ior, const uint32_t *transl,
                               float *dx, float *dy, float *dz, uint32_t *transl_cropped)
{
    const size_t Bx = bounds[0], By = bounds[1], Bz = bounds[2];
    const size_t n = Bx * By * Bz;
    const size_t Ox = Bx - 2, Oy = By - 2, Oz = Bz - 2;
    float *lg = (float *)malloc(n * sizeof(float));
    if (!lg) return -3;
    int bad = 0;
    #pragma omp parallel for
    for (size_t i = 0; i < n; ++i) {
        if (ior[i] <= 0) { bad = 1; lg[i] = 0; continue; }
        /* image_util.cpp:611 "log(_ior[i]) * 0x420000": unqualified ::log is the C double
         * function (checked against the reference build's _ior_log bit-for-bit), the
         * product is double * int -> double, narrowed to float on store */
        lg[i] = (float)(log((double)ior[i]) * (double)0x420000);
    }
    if (bad) { free(lg); return -2; }
    float *out[3] = {dx, dy, dz};
    for (int ax = 0; ax < 3; ++ax) {
        size_t off[18]; int32_t val[18];
        stamp_terms(ax, By, Bz, off, val);
        const float weight = 812.0f * 256.0f;
        #pragma omp parallel for collapse(2)
        for (size_t x = 0; x < Ox; ++x)
            for (size_t y = 0; y < Oy; ++y)
                for (size_t z = 0; z < Oz; ++z) {
                    const float *p = lg + (x * By + y) * Bz + z;
                    float sum = 0;
                    for (int t = 0; t < 18; ++t) sum = fmaf((float)val[t], p[off[t]], sum);
                    out[ax][(x * Oy + y) * Oz + z] = sum / weight;
                }
    }
    #pragma omp parallel for collapse(2)
    for (size_t x = 0; x < Ox; ++x)
        for (size_t y = 0; y < Oy; ++y)
            for (size_t z = 0; z < Oz; ++z)
                transl_cropped[(x * Oy + y) * Oz + z] = transl[((x + 1) * By + (y + 1)) * Bz + (z + 1)];
    free(lg);
    return 0;
}

/*
 * RaytraceScene<ior_t,iorlog_t,diff_t> ctor, image_util.cpp:501-574 (3-D only):
 *   ior_log[i] = round(log(ior/65536.0) * 0x420000) (double), overflow-checked  :531-543
 *   diff[ax]   = int32 convolution, divRoundClosest(sum, 812*256), narrowed to
 *                int16 with overflow check ("differention overflow")           :281-293
 * Returns 0; -2 ior_log overflow; -4 differentiation overflow.
 */
int vrt_oracle_scene_build_i16(const size_t bounds[3], const uint32_t *ior, const uint32_t *transl,
                               int16_t *dx, int16_t *dy, int16_t *dz, uint32_t *transl_cropped)
{
    const size_t Bx = bounds[0], By = bounds[1], Bz = bounds[2];
    const size_t n = Bx * By * Bz;
    const size_t Ox = Bx - 2, Oy = By - 2, Oz = Bz - 2;
    int32_t *lg = (int32_t *)malloc(n * sizeof(int32_t));
    if (!lg) return -3;
    int bad = 0;
    #pragma omp parallel for
    for (size_t i = 0; i < n; ++i) {
        double fior = (double)ior[i] / 65536.0;
        double tmp = log(fior) * (double)0x420000;
        if (tmp > 2147483647.0 || tmp < -2147483648.0) { bad = 1; lg[i] = 0; continue; }
        lg[i] = (int32_t)round(tmp);
    }
    if (bad) { free(lg); return -2; }
    int16_t *out[3] = {dx, dy, dz};
    int ovf = 0;
    for (int ax = 0; ax < 3; ++ax) {
        size_t off[18]; int32_t val[18];
        stamp_terms(ax, By, Bz, off, val);
        const int32_t weight = 812 * 0x100;
        #pragma omp parallel for collapse(2)
        for (size_t x = 0; x < Ox; ++x)
            for (size_t y = 0; y < Oy; ++y)
                for (size_t z = 0; z < Oz; ++z) {
                    const int32_t *p = lg + (x * By + y) * Bz + z;
                    int32_t sum = 0;
                    for (int t = 0; t < 18; ++t) sum = (int32_t)((uint32_t)sum + (uint32_t)(val[t] * p[off[t]]));
                    sum = div_round_closest_i32(sum, weight);
                    int16_t o = (int16_t)sum;
                    if (o != sum) ovf = 1;
                    out[ax][(x * Oy + y) * Oz + z] = o;
                }
    }
    #pragma omp parallel for collapse(2)
    for (size_t x = 0; x < Ox; ++x)
        for (size_t y = 0; y < Oy; ++y)
            for (size_t z = 0; z < Oz; ++z)
                transl_cropped[(x * Oy + y) * Oz + z] = transl[((x + 1) * By + (y + 1)) * Bz + (z + 1)];
    free(lg);
    return ovf ? -4 : 0;
}

/* ========================================================================== *
 * Ray front end: image_util.cpp:645-772
 * ========================================================================== */

/* interpolator<float>, image_util.h:390-431: corners ordered with x major in the
 * index list; lerp d = 0 (x) first over halves, then y, then z; * 2^-48. */
static float interp_ior_f32(const float *ior, const size_t b[3], const uint32_t p[3])
{
    size_t base = ((size_t)(p[0] >> 16) * b[1] + (p[1] >> 16)) * b[2] + (p[2] >> 16);
    float v[8];
    /* _indices after the expansion loop (d = 2,1,0 with offsets 1, Bz, By*Bz):
     * [0, +1, +Bz, +Bz+1, +ByBz, +ByBz+1, +ByBz+Bz, +ByBz+Bz+1] */
    size_t o1 = 1, o2 = b[2], o4 = b[1] * b[2];
    for (int k = 0; k < 8; ++k) v[k] = ior[base + ((k & 1) ? o1 : 0) + ((k & 2) ? o2 : 0) + ((k & 4) ? o4 : 0)];
    int cnt = 8;
    for (int d = 0; d < 3; ++d) {
        size_t multr = p[d] % 0x10000, multl = 0x10000 - multr;
        /* "_image_values[i] * multl + _image_values[i+half] * multr": size_t -> float
         * conversions, contracted as fma(a, l, b*r) like the rest of the object */
        for (int i = 0; i < cnt / 2; ++i) v[i] = fmaf(v[i], (float)multl, v[i + cnt / 2] * (float)multr);
        cnt /= 2;
    }
    return v[0] * (float)(1. / 281474976710656.0);
}

/* interpolator<ior_t>, image_util.h:348-388: uint64 products, divRoundClosest by 0x10000,
 * narrowed back to uint32 at each level. */
static uint32_t interp_ior_u32(const uint32_t *ior, const size_t b[3], const uint32_t p[3])
{
    size_t base = ((size_t)(p[0] >> 16) * b[1] + (p[1] >> 16)) * b[2] + (p[2] >> 16);
    uint32_t v[8];
    size_t o1 = 1, o2 = b[2], o4 = b[1] * b[2];
    for (int k = 0; k < 8; ++k) v[k] = ior[base + ((k & 1) ? o1 : 0) + ((k & 2) ? o2 : 0) + ((k & 4) ? o4 : 0)];
    int cnt = 8;
    for (int d = 0; d < 3; ++d) {
        uint64_t multr = p[d] % 0x10000, multl = 0x10000 - multr;
        for (int i = 0; i < cnt / 2; ++i) {
            uint64_t n = (uint64_t)v[i] * multl + (uint64_t)v[i + cnt / 2] * multr;
            v[i] = (uint32_t)((n + 0x8000) / 0x10000);
        }
        cnt /= 2;
    }
    return v[0];
}

/*
 * RaytraceScene::trace_rays prologue (image_util.cpp:673-722): range check,
 * -0x8000, interpolate ior, scale direction, -0x8000.  In place on pos/dir.
 * Returns 0; 1+ray index of the first out-of-range ray as -(idx+10); -5 on
 * "Normalize length failed" (int16 overflow).
 */
long vrt_oracle_ray_prologue(int ior_type, const size_t bounds[3], const void *ior,
                             size_t nrays, uint32_t *pos, int dir_type, void *dir)
{
    long err = 0;
    for (size_t i = 0; i < nrays; ++i) {
        uint32_t *p = pos + 3 * i;
        for (int d = 0; d < 3; ++d) {
            size_t lhs = p[d], rhs = bounds[d];
            if (lhs < 0x10000 || lhs + 1 >= rhs * 0x10000) { if (!err) err = -(long)(i + 10); }
        }
        if (err) continue;
        for (int d = 0; d < 3; ++d) p[d] -= 0x8000;
        if (ior_type == VRT_F32) {
            float s = interp_ior_f32((const float *)ior, bounds, p);
            if (dir_type == VRT_F32) { float *q = (float *)dir + 3 * i; for (int d = 0; d < 3; ++d) q[d] *= s; }
            else return -6;   /* float scene + int16 dirs is not instantiated (image_util.cpp:940-956) */
        } else {
            uint32_t s = interp_ior_u32((const uint32_t *)ior, bounds, p);
            if (dir_type == VRT_I16) {
                int16_t *q = (int16_t *)dir + 3 * i;
                for (int d = 0; d < 3; ++d) {
                    int64_t n = (int64_t)q[d] * (int64_t)s, dd = 0x10000;
                    int64_t t = ((n < 0) ^ (dd < 0)) ? ((n - dd / 2) / dd) : ((n + dd / 2) / dd);
                    if (t > 32767 || t < -32768) { if (!err) err = -5; }
                    q[d] = (int16_t)t;
                }
            } else return -6;
        }
        for (int d = 0; d < 3; ++d) p[d] -= 0x8000;
    }
    return err;
}

int vrt_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
