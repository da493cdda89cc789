// vrt_engine.hip -- host runtime + C ABI (include/vrt.h) of the MI355X trace_rays engine.
//
// Replaces the reference's engine object / dispatcher
//   TraceRaysCu ctor / trace_rays_cu / dtor   /root/reference/src/cuda_volume_raytracer.cu:644-989
// The reference AoS-packs rays on the host, cuts them into 0x8000-ray chunks and runs one
// blocking launch per chunk (.cu:797-946).  Here the field is interleaved on the device once,
// rays stay in the caller's SoA arrays, and one launch covers all rays of a call.
//
// There is deliberately no CPU fallback in this file: without a HIP device every entry
// point fails with VRT_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>
#include <string>

#include "../../include/vrt.h"
#include "vrt_march.h"
#include "vrt_slab.h"
#include "vrt_sort.h"
#include "vrt_scene.h"

#include <cmath>
#include <thread>
#include <vector>
#include <atomic>

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? VRT_ERR_OUT_OF_MEMORY : VRT_ERR_HIP,           \
                        "%s in %s at line %d (%d)", hipGetErrorString(e_), __FILE__, __LINE__, (int)e_); \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess) { ok = (prev == dev) || hipSetDevice(dev) == hipSuccess; }
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// grow-only device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct vrt_engine {
    int device = 0;
    int field_type = VRT_F32;
    uint32_t B[3] = {0, 0, 0};
    size_t nvox = 0;
    void *field = nullptr;         // interleaved AoS field, resident for the engine's lifetime
    size_t field_bytes = 0;
    hipStream_t stream = nullptr;  // used by the host-pointer entry point
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;   // around the most recent march kernel
    bool ev_valid = false;
    int num_cu = 256;
    // workspaces (grow-only; no allocation in steady state)
    DevBuf sort_keys, sort_keys_alt, sort_vals, sort_vals_alt, sort_tmp, sort_bounds;
    DevBuf transl;                 // translucency per field voxel, kept only for the opt-in absorption mode
    DevBuf group_ctr;                  // group counter of the persistent slab kernel (vrt_options.schedule)
    DevBuf stats;                  // debug counters, allocated only when VRT_DEBUG_STATS=1
    size_t stats_count = 0;
    DevBuf h_in_pos, h_in_dir, h_out_pos, h_out_dir, h_out_iter, h_out_light, h_path;
    vrt_options opt;
};

namespace {

// ---- field build: TraceRaysCu ctor, cuda_volume_raytracer.cu:654-669 ------------------------

__device__ __forceinline__ int32_t w_channel(uint32_t tr)
{
    // (int64)0x7FFFFFFF - (int64)tr always fits int32; C++ '/' truncates like the reference's
    return (int32_t)(0x7FFFFFFFu - tr) / 0x10000;
}

__global__ __launch_bounds__(256) void k_interleave_f32(const float *dx, const float *dy, const float *dz,
                                                        const uint32_t *tr, vrt::vox_f32 *out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        vrt::vox_f32 v;
        v.x = dx[i]; v.y = dy[i]; v.z = dz[i]; v.w = (float)w_channel(tr[i]);
        out[i] = v;
    }
}

__global__ __launch_bounds__(256) void k_interleave_i16(const int16_t *dx, const int16_t *dy, const int16_t *dz,
                                                        const uint32_t *tr, vrt::vox_i16 *out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        vrt::vox_i16 v;
        v.x = dx[i]; v.y = dy[i]; v.z = dz[i]; v.w = (int16_t)w_channel(tr[i]);
        out[i] = v;
    }
}

int launch_interleave(vrt_engine *e, const void *const d_diff[3], const uint32_t *d_tr, size_t off, size_t n,
                      hipStream_t s)
{
    const int grid = (int)std::min<size_t>((n + 255) / 256, (size_t)e->num_cu * 8);
    if (e->field_type == VRT_F32)
        k_interleave_f32<<<grid, 256, 0, s>>>((const float *)d_diff[0], (const float *)d_diff[1],
                                               (const float *)d_diff[2], d_tr, (vrt::vox_f32 *)e->field + off, n);
    else
        k_interleave_i16<<<grid, 256, 0, s>>>((const int16_t *)d_diff[0], (const int16_t *)d_diff[1],
                                               (const int16_t *)d_diff[2], d_tr, (vrt::vox_i16 *)e->field + off, n);
    HIP_TRY(hipGetLastError());
    return VRT_OK;
}

int check_create_args(vrt_handle *out, int dim, const uint16_t *bounds, int field_type, const void *const *diff,
                      const uint32_t *tr, size_t nvox)
{
    if (!out) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_create: out is NULL");
    *out = nullptr;
    if (dim != 3) return fail(VRT_ERR_ILLEGAL_DIMENSION, "Illegal dimension: %d (only 3-D fields are supported)", dim);
    if (!bounds || !diff || !tr) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_create: NULL argument");
    if (field_type != VRT_F32 && field_type != VRT_I16)
        return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_create: unknown field_type %d", field_type);
    for (int d = 0; d < 3; ++d) {
        if (!diff[d]) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_create: diff[%d] is NULL", d);
        if (bounds[d] == 0) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_create: bounds[%d] is zero", d);
    }
    if ((size_t)bounds[0] * bounds[1] * bounds[2] != nvox)
        return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_create: nvox %zu != product of bounds %zu", nvox,
                    (size_t)bounds[0] * bounds[1] * bounds[2]);
    return VRT_OK;
}

int engine_alloc(vrt_engine **pe, const uint16_t *bounds, int field_type, size_t nvox, const vrt_options *opt)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(VRT_ERR_NO_DEVICE, "no HIP device available (this engine has no CPU path)");
    vrt_engine *e = new (std::nothrow) vrt_engine();
    if (!e) return fail(VRT_ERR_OUT_OF_MEMORY, "host allocation failed");
    if (opt) e->opt = *opt; else vrt_options_default(&e->opt);
    int dev = e->opt.device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= ndev) { delete e; return fail(VRT_ERR_INVALID_ARGUMENT, "device %d out of range (%d devices)", dev, ndev); }
    e->device = dev;
    e->field_type = field_type;
    for (int d = 0; d < 3; ++d) e->B[d] = bounds[d];
    e->nvox = nvox;
    e->field_bytes = nvox * (field_type == VRT_F32 ? 16 : 8);
    *pe = e;
    return VRT_OK;
}

int engine_init_device(vrt_engine *e)
{
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, e->device));
    e->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&e->ev_begin));
    HIP_TRY(hipEventCreate(&e->ev_end));
    // +64 bytes of zeroed tail so that a 16-byte pair load at the very last voxel stays inside
    // the allocation (the march never reads it: the loop requires pos>>16 < bounds-1).
    HIP_TRY(hipMalloc(&e->field, e->field_bytes + 64));
    HIP_TRY(hipMemsetAsync((char *)e->field + e->field_bytes, 0, 64, e->stream));
    return VRT_OK;
}

void engine_free(vrt_engine *e)
{
    if (!e) return;
    DeviceGuard g(e->device);
    if (e->stream) { (void)hipStreamSynchronize(e->stream); (void)hipStreamDestroy(e->stream); }
    if (e->field) (void)hipFree(e->field);
    if (e->ev_begin) (void)hipEventDestroy(e->ev_begin);
    if (e->ev_end) (void)hipEventDestroy(e->ev_end);
    DevBuf *bufs[] = {&e->transl, &e->stats, &e->group_ctr, &e->sort_keys, &e->sort_keys_alt, &e->sort_vals, &e->sort_vals_alt, &e->sort_tmp, &e->sort_bounds,
                      &e->h_in_pos, &e->h_in_dir, &e->h_out_pos, &e->h_out_dir, &e->h_out_iter, &e->h_out_light,
                      &e->h_path};
    for (DevBuf *b : bufs) b->release();
    delete e;
}

// ---- launch ---------------------------------------------------------------------------------

template <typename FieldT, typename DirT>
int launch_typed(vrt_engine *e, const vrt::MarchParams &p, int kernel, hipStream_t s)
{
    const bool path = p.path != nullptr;
    if (kernel == VRT_KERNEL_SLAB) {
        return vrt::launch_slab<FieldT, DirT>(p, path, e->num_cu, s) == hipSuccess
                   ? VRT_OK : fail(VRT_ERR_HIP, "slab kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    }
    const unsigned grid = (p.nrays + 255) / 256;
    if (p.transl) {
        if (path) vrt::k_march_direct<FieldT, DirT, true, true><<<grid, 256, 0, s>>>(p);
        else      vrt::k_march_direct<FieldT, DirT, false, true><<<grid, 256, 0, s>>>(p);
    } else {
        if (path) vrt::k_march_direct<FieldT, DirT, true, false><<<grid, 256, 0, s>>>(p);
        else      vrt::k_march_direct<FieldT, DirT, false, false><<<grid, 256, 0, s>>>(p);
    }
    HIP_TRY(hipGetLastError());
    return VRT_OK;
}

// Fast-path window of |dir|^2 for this launch's invscale (vrt_march.h, StepWindow).
void step_window(const float *inv, uint32_t *lo, uint32_t *span)
{
    const uint32_t base_lo = 32u << 23, base_hi = 251u << 23;          // exhaustively verified short-division range
    *lo = base_lo;
    *span = 0;                                                          // empty unless everything below is finite
    const double m = std::max(std::fabs((double)inv[0]), std::max(std::fabs((double)inv[1]), std::fabs((double)inv[2])));
    if (!(m < 1.0e30)) return;                                          // NaN or absurd invscale: always the long way
    const double b = 2.0 * 1107296256.0 * m / 2147483648.0;             // 2C * max|inv| / 2^31
    const double t = b * b * (1.0 + 1.0 / 65536.0);
    uint32_t need = base_lo;
    if (t > 0.0) {
        float tf = (float)t;
        if ((double)tf < t) tf = std::nextafterf(tf, INFINITY);
        if (!(tf < 3.0e38f)) return;
        uint32_t bits;
        memcpy(&bits, &tf, 4);
        need = bits + 1u;                                               // strictly above the bound
    }
    const uint32_t l = std::max(base_lo, need);
    if (l >= base_hi) return;
    *lo = l;
    *span = base_hi - l;
}

int trace_device_impl(vrt_engine *e, int dir_type, size_t nrays, const uint32_t *d_sp, const void *d_sd,
                      uint32_t *d_ep, void *d_ed, uint32_t *d_ei, uint32_t *d_rl, uint32_t *d_path,
                      const float *invscale, uint32_t iterations, hipStream_t s, const vrt_options *opt,
                      uint32_t out_pos_bias = 0, uint32_t min_brightness = 0)
{
    if (nrays == 0) return VRT_OK;
    const vrt_options &o = opt ? *opt : e->opt;
    vrt::MarchParams p;
    p.field = e->field;
    p.Bx = e->B[0]; p.By = e->B[1]; p.Bz = e->B[2];
    p.invx = invscale[0]; p.invy = invscale[1]; p.invz = invscale[2];
    step_window(invscale, &p.win_lo, &p.win_span);
    p.start_pos = d_sp; p.start_dir = d_sd;
    p.end_pos = d_ep; p.end_dir = d_ed; p.end_iter = d_ei; p.remaining_light = d_rl;
    p.path = d_path;
    p.order = nullptr;
    p.wave_order = nullptr;
    p.iterations = iterations;
    p.nrays = (uint32_t)nrays;
    p.out_pos_bias = out_pos_bias;
    p.transl = nullptr;
    p.min_brightness = min_brightness;
    if (o.absorption) {
        if (!e->transl.p)
            return fail(VRT_ERR_INVALID_ARGUMENT, "absorption requested at trace time but the engine was created "
                                                  "without vrt_options.absorption = 1 (translucency not resident)");
        p.transl = (const uint32_t *)e->transl.p;
    }
    p.slab_lds_bytes = 0;
    if (o.slab_lds_bytes > 0) {
        if (o.slab_lds_bytes < 1024 || o.slab_lds_bytes > 40960 || (o.slab_lds_bytes & 15))
            return fail(VRT_ERR_INVALID_ARGUMENT, "slab_lds_bytes must be a multiple of 16 in [1024, 40960]");
        p.slab_lds_bytes = (uint32_t)o.slab_lds_bytes;
    }

    p.stats = nullptr;
    if (const char *dbg = getenv("VRT_DEBUG_STATS"); dbg && dbg[0] == '1') {
        // 64 counters, then (development builds with -DVRT_SLAB_TIMELINE) 4 words per wavefront
        e->stats_count = 64 + 4 * (((size_t)nrays + 63) / 64);
        HIP_TRY(e->stats.reserve(e->stats_count * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(e->stats.p, 0, e->stats_count * sizeof(unsigned long long), s));
        p.stats = (unsigned long long *)e->stats.p;
    }
    int kernel = o.kernel;
    if (kernel == VRT_KERNEL_AUTO) kernel = d_path ? VRT_KERNEL_DIRECT : VRT_KERNEL_SLAB;
    if (kernel != VRT_KERNEL_DIRECT && kernel != VRT_KERNEL_SLAB)
        return fail(VRT_ERR_INVALID_ARGUMENT, "unknown kernel selector %d", kernel);
    if (kernel == VRT_KERNEL_SLAB && d_path) kernel = VRT_KERNEL_DIRECT;   // path tracing is write-bound: direct kernel

    // tile ordering: group rays that enter the field close together into the same wavefront
    const bool sort = o.sort_rays > 0 || (o.sort_rays == 0 && nrays >= 4096);
    if (sort) {
        const uint32_t *order = nullptr, *wave_order = nullptr;
        // longest wavefronts first: always for 3; in auto mode only when the launch has more wavefronts than the
        // GPU holds at once (24 per CU at the default LDS segment) -- otherwise every wavefront starts at t = 0 anyway
        const int wave_lpt = !d_path && (o.sort_rays == 3 || (o.sort_rays == 0 && (nrays + 63) / 64 > (size_t)e->num_cu * 24));
        hipError_t se = vrt::build_tile_order(d_sp, d_sd, dir_type, (uint32_t)nrays, e->B[0], e->B[1], e->B[2], o.sort_rays == 2 ? 1 : 0, e->sort_keys, e->sort_keys_alt,
                                              e->sort_vals, e->sort_vals_alt, e->sort_tmp, e->sort_bounds, s, &order,
                                              wave_lpt, iterations, &wave_order, e->field, e->field_type == VRT_I16, invscale);
        if (se != hipSuccess) return fail(se == hipErrorOutOfMemory ? VRT_ERR_OUT_OF_MEMORY : VRT_ERR_HIP,
                                          "ray tile ordering failed: %s", hipGetErrorString(se));
        p.order = order;
        p.wave_order = wave_order;
    }

    // scheduling of the slab kernel: static (one wavefront per group of 64 rays, hardware dispatch) or
    // persistent (as many wavefronts as the GPU holds, each drawing groups from a counter)
    p.group_counter = nullptr;
    {
        const size_t groups = (nrays + 63) / 64;
        int schedule = o.schedule;
        if (const char *ev = getenv("VRT_SCHEDULE"); ev && schedule == 0) schedule = atoi(ev);
        if (schedule < 0 || schedule > 2) return fail(VRT_ERR_INVALID_ARGUMENT, "unknown schedule %d", schedule);
        // auto: persistent once the launch has more wavefronts than the GPU holds (measured -4 % on config 3)
        if (schedule == 0) schedule = groups > (size_t)e->num_cu * 24 ? 2 : 1;
        if (schedule == 2 && kernel == VRT_KERNEL_SLAB && groups > 1) {
            HIP_TRY(e->group_ctr.reserve(sizeof(uint32_t)));
            HIP_TRY(hipMemsetAsync(e->group_ctr.p, 0, sizeof(uint32_t), s));
            p.group_counter = (uint32_t *)e->group_ctr.p;
        }
    }

    HIP_TRY(hipEventRecord(e->ev_begin, s));
    int rc;
    if (e->field_type == VRT_F32)
        rc = dir_type == VRT_F32 ? launch_typed<float, float>(e, p, kernel, s) : launch_typed<float, int16_t>(e, p, kernel, s);
    else
        rc = dir_type == VRT_F32 ? launch_typed<int16_t, float>(e, p, kernel, s) : launch_typed<int16_t, int16_t>(e, p, kernel, s);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(e->ev_end, s));
    e->ev_valid = true;
    return VRT_OK;
}

int check_trace_args(vrt_handle h, int dir_type, size_t nrays, const void *sp, const void *sd, const void *ep,
                     const void *ed, const void *ei, const void *rl, const float *invscale, uint32_t iterations)
{
    if (!h) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_trace: NULL handle");
    if (dir_type != VRT_F32 && dir_type != VRT_I16)
        return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_trace: unknown dir_type %d", dir_type);
    if (!invscale) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_trace: invscale is NULL");
    if (iterations == 0)
        return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_trace: iterations must be >= 1 (the reference underflows at 0, "
                                              "cuda_volume_raytracer.cu:333)");
    if (nrays > 0xFFFFFFF0u) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_trace: too many rays (%zu)", nrays);
    if (nrays && (!sp || !sd || !ep || !ed || !ei || !rl))
        return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_trace: NULL ray array");
    return VRT_OK;
}

// Host-pointer trace shared by vrt_trace and vrt_scene_trace.  `pre` (optional) rewrites the
// uploaded start arrays on the device before the march (the scene layer's ray prologue).
template <typename Pre>
int trace_host_impl(vrt_engine *e, int dir_type, size_t nrays, const uint32_t *sp, const void *sd, uint32_t *ep,
                    void *ed, uint32_t *ei, uint32_t *rl, uint32_t *path, const float *invscale, uint32_t iterations,
                    const vrt_options *opt, uint32_t out_pos_bias, uint32_t min_brightness, Pre pre)
{
    const vrt_options &o = opt ? *opt : e->opt;
    const size_t dsz = dir_type == VRT_F32 ? 4 : 2;
    hipStream_t s = e->stream;
    // Rays go up in one piece when no path is requested; with path tracing the device path
    // buffer (iterations*12 bytes per ray) bounds the chunk, as the reference's 0x8000-ray
    // chunks bound its own (.cu:798, 840).
    size_t chunk = nrays;
    if (path) {
        const size_t per_ray = (size_t)iterations * 12;
        size_t budget = (size_t)2 << 30;
        if (const char *b = getenv("VRT_PATH_BUDGET_BYTES")) {          // tests shrink it to force several chunks
            const unsigned long long v = strtoull(b, nullptr, 10);
            if (v > 0) budget = (size_t)v;
        }
        chunk = std::max<size_t>(1, std::min(nrays, budget / std::max<size_t>(per_ray, 1)));
    }
    HIP_TRY(e->h_in_pos.reserve(chunk * 12));
    HIP_TRY(e->h_in_dir.reserve(chunk * 3 * dsz));
    HIP_TRY(e->h_out_pos.reserve(chunk * 12));
    HIP_TRY(e->h_out_dir.reserve(chunk * 3 * dsz));
    HIP_TRY(e->h_out_iter.reserve(chunk * 4));
    HIP_TRY(e->h_out_light.reserve(chunk * 4));
    if (path) HIP_TRY(e->h_path.reserve(chunk * (size_t)iterations * 12));

    const size_t nchunks = (nrays + chunk - 1) / chunk;
    bool cap_hit = false;
    for (size_t c = 0, off = 0; off < nrays; ++c, off += chunk) {
        const size_t n = std::min(chunk, nrays - off);
        if (o.loglevel < 0) printf("iteration %zu of %zu\n", c, nchunks);          // .cu:823-826
        HIP_TRY(hipMemcpyAsync(e->h_in_pos.p, sp + 3 * off, n * 12, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(e->h_in_dir.p, (const char *)sd + 3 * off * dsz, n * 3 * dsz, hipMemcpyHostToDevice, s));
        int rc = pre((uint32_t *)e->h_in_pos.p, e->h_in_dir.p, n, off, s);
        if (rc) return rc;
        rc = trace_device_impl(e, dir_type, n, (const uint32_t *)e->h_in_pos.p, e->h_in_dir.p,
                               (uint32_t *)e->h_out_pos.p, e->h_out_dir.p, (uint32_t *)e->h_out_iter.p,
                               (uint32_t *)e->h_out_light.p, path ? (uint32_t *)e->h_path.p : nullptr, invscale,
                               iterations, s, opt, out_pos_bias, min_brightness);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(ep + 3 * off, e->h_out_pos.p, n * 12, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync((char *)ed + 3 * off * dsz, e->h_out_dir.p, n * 3 * dsz, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(ei + off, e->h_out_iter.p, n * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(rl + off, e->h_out_light.p, n * 4, hipMemcpyDeviceToHost, s));
        if (path)
            HIP_TRY(hipMemcpyAsync(path + off * (size_t)iterations * 3, e->h_path.p, n * (size_t)iterations * 12,
                                   hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    for (size_t i = 0; i < nrays && !cap_hit; ++i) cap_hit = (ei[i] == iterations);
    if (cap_hit) printf("Warning, maximum iterations hitted\n");                   // .cu:512-515
    if (o.loglevel < 0) printf("cpu: 0 gpu: %zu\n", nchunks);                      // .cu:948-951
    return VRT_OK;
}

// parallel-for over [0, n) on the host (the log pass of the scene build)
template <typename F>
void host_parallel_for(size_t n, F f)
{
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt == 0 ? 1 : (nt > 64 ? 64 : nt);
    if (n < 65536 || nt == 1) { f(0, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const size_t b = std::min(n, (size_t)t * per), e2 = std::min(n, b + per);
        if (b < e2) th.emplace_back([=] { f(b, e2); });
    }
    for (auto &x : th) x.join();
}

}  // namespace

struct vrt_scene {
    vrt_engine *engine = nullptr;
    int ior_type = VRT_F32;
    uint32_t N[3] = {0, 0, 0};
    void *d_ior = nullptr;          // the ior volume stays resident for the ray prologue's interpolation
    void *d_density = nullptr;      // LUT scenes: the 8-bit volume stays resident instead, with its ior table
    void *d_lut_ior = nullptr;
    void *d_err = nullptr;          // vrt::SceneErr
    DevBuf pro_pos, pro_dir;
};

// =============================================================================================
// C ABI
// =============================================================================================


// Exhaustive self-test of the march step's short division (vrt_march.h: ilen2_short / div_window):
// one block per (sign, biased exponent), threads sweep all 2^23 mantissas.
__global__ __launch_bounds__(256) void k_selftest_ilen2(unsigned long long *out /* [2]: mismatches, admitted */)
{
    const uint32_t expo = blockIdx.x & 255u, sign = blockIdx.x >> 8;
    unsigned long long bad = 0, adm = 0;
    for (uint32_t m = threadIdx.x; m < (1u << 23); m += blockDim.x) {
        const float d = __uint_as_float((sign << 31) | (expo << 23) | m);
        if (!vrt::div_window(d)) continue;
        ++adm;
        const float q = 0x42000000p0f / d;                                   // IEEE divide, as .cu:346
        bad += __float_as_uint(vrt::ilen2_short(d)) != __float_as_uint(q + q);
    }
    for (int o = 32; o > 0; o >>= 1) { bad += __shfl_xor(bad, o); adm += __shfl_xor(adm, o); }
    if ((threadIdx.x & 63u) == 0) { atomicAdd(&out[0], bad); atomicAdd(&out[1], adm); }
}

extern "C" {

void vrt_options_default(vrt_options *opt)
{
    if (!opt) return;
    memset(opt, 0, sizeof *opt);
    opt->loglevel = 0;
    opt->minimum_gpu = 0x80;       // types.h:90
    opt->write_instance = 0;
    opt->max_cpu = 256;
    opt->device = -1;
    opt->kernel = VRT_KERNEL_AUTO;
    opt->sort_rays = 0;
}

int vrt_abi_version(void) { return VRT_ABI_VERSION; }

int vrt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *vrt_last_error(void) { return g_err.c_str(); }

int vrt_create_from_device(vrt_handle *out, int dim, const uint16_t *bounds, int field_type,
                           const void *const *d_diff, const uint32_t *d_tr, size_t nvox, const vrt_options *opt)
{
    int rc = check_create_args(out, dim, bounds, field_type, d_diff, d_tr, nvox);
    if (rc) return rc;
    vrt_engine *e = nullptr;
    rc = engine_alloc(&e, bounds, field_type, nvox, opt);
    if (rc) return rc;
    DeviceGuard g(e->device);
    rc = engine_init_device(e);
    if (!rc && e->opt.absorption) {
        if (e->transl.reserve(nvox * 4) != hipSuccess) rc = fail(VRT_ERR_OUT_OF_MEMORY, "device allocation failed");
        else if (hipMemcpyAsync(e->transl.p, d_tr, nvox * 4, hipMemcpyDeviceToDevice, e->stream) != hipSuccess)
            rc = fail(VRT_ERR_HIP, "translucency copy failed");
    }
    if (!rc) rc = launch_interleave(e, d_diff, d_tr, 0, nvox, e->stream);
    if (!rc && hipStreamSynchronize(e->stream) != hipSuccess) rc = fail(VRT_ERR_HIP, "field interleave failed");
    if (rc) { engine_free(e); return rc; }
    *out = e;
    return VRT_OK;
}

int vrt_create(vrt_handle *out, int dim, const uint16_t *bounds, int field_type, const void *const *diff,
               const uint32_t *tr, size_t nvox, const vrt_options *opt)
{
    int rc = check_create_args(out, dim, bounds, field_type, diff, tr, nvox);
    if (rc) return rc;
    vrt_engine *e = nullptr;
    rc = engine_alloc(&e, bounds, field_type, nvox, opt);
    if (rc) return rc;
    DeviceGuard g(e->device);
    rc = engine_init_device(e);
    if (rc) { engine_free(e); return rc; }

    // Upload + interleave in bounded chunks (the reference interleaves serially on the host and
    // uploads the 4-channel copy, .cu:660-685; here the host arrays are uploaded as they are and
    // interleaved on the device).
    const size_t esz = field_type == VRT_F32 ? 4 : 2;
    const size_t chunk = std::min<size_t>(nvox, (size_t)32 << 20);
    DevBuf tmp[4];
    auto cleanup = [&](int code) { for (DevBuf &b : tmp) b.release(); engine_free(e); return code; };
    for (int k = 0; k < 3; ++k)
        if (tmp[k].reserve(chunk * esz) != hipSuccess) return cleanup(fail(VRT_ERR_OUT_OF_MEMORY, "device allocation failed"));
    if (tmp[3].reserve(chunk * 4) != hipSuccess) return cleanup(fail(VRT_ERR_OUT_OF_MEMORY, "device allocation failed"));
    if (e->opt.absorption && e->transl.reserve(nvox * 4) != hipSuccess)
        return cleanup(fail(VRT_ERR_OUT_OF_MEMORY, "device allocation failed"));
    for (size_t off = 0; off < nvox; off += chunk) {
        const size_t n = std::min(chunk, nvox - off);
        if (e->opt.absorption &&
            hipMemcpyAsync((uint32_t *)e->transl.p + off, tr + off, n * 4, hipMemcpyHostToDevice, e->stream) != hipSuccess)
            return cleanup(fail(VRT_ERR_HIP, "field upload failed"));
        for (int k = 0; k < 3; ++k)
            if (hipMemcpyAsync(tmp[k].p, (const char *)diff[k] + off * esz, n * esz, hipMemcpyHostToDevice, e->stream) != hipSuccess)
                return cleanup(fail(VRT_ERR_HIP, "field upload failed"));
        if (hipMemcpyAsync(tmp[3].p, tr + off, n * 4, hipMemcpyHostToDevice, e->stream) != hipSuccess)
            return cleanup(fail(VRT_ERR_HIP, "field upload failed"));
        const void *d[3] = {tmp[0].p, tmp[1].p, tmp[2].p};
        rc = launch_interleave(e, d, (const uint32_t *)tmp[3].p, off, n, e->stream);
        if (rc) return cleanup(rc);
        if (hipStreamSynchronize(e->stream) != hipSuccess) return cleanup(fail(VRT_ERR_HIP, "field interleave failed"));
    }
    for (DevBuf &b : tmp) b.release();
    *out = e;
    return VRT_OK;
}

int vrt_trace_device(vrt_handle h, int dir_type, size_t nrays, const uint32_t *d_sp, const void *d_sd,
                     uint32_t *d_ep, void *d_ed, uint32_t *d_ei, uint32_t *d_rl, uint32_t *d_path,
                     const float *invscale, uint32_t min_brightness, uint32_t iterations, void *stream,
                     const vrt_options *opt)
{
    int rc = check_trace_args(h, dir_type, nrays, d_sp, d_sd, d_ep, d_ed, d_ei, d_rl, invscale, iterations);
    if (rc) return rc;
    DeviceGuard g(h->device);
    // min_brightness is ignored unless the opt-in absorption mode is on, as in the reference (.cu:785)
    return trace_device_impl(h, dir_type, nrays, d_sp, d_sd, d_ep, d_ed, d_ei, d_rl, d_path, invscale, iterations,
                             (hipStream_t)stream, opt, 0, min_brightness);
}

int vrt_trace(vrt_handle h, int dir_type, size_t nrays, const uint32_t *sp, const void *sd, uint32_t *ep, void *ed,
              uint32_t *ei, uint32_t *rl, uint32_t *path, const float *invscale, uint32_t min_brightness,
              uint32_t iterations, const vrt_options *opt)
{
    int rc = check_trace_args(h, dir_type, nrays, sp, sd, ep, ed, ei, rl, invscale, iterations);
    if (rc) return rc;
    if (nrays == 0) return VRT_OK;
    DeviceGuard g(h->device);
    return trace_host_impl(h, dir_type, nrays, sp, sd, ep, ed, ei, rl, path, invscale, iterations, opt, 0, min_brightness,
                           [](uint32_t *, void *, size_t, size_t, hipStream_t) { return (int)VRT_OK; });
}

// ---- scene layer ---------------------------------------------------------------------------

// Shared by vrt_scene_create (density == nullptr: `ior`/`translucency` are nvox-element arrays) and
// vrt_scene_create_lut (density != nullptr: they are 256-entry tables indexed by the 8-bit volume).
static int scene_create_impl(vrt_scene_handle *out, int dim, const size_t *bounds, int ior_type, const uint8_t *density,
                             const void *ior, const uint32_t *translucency, size_t nvox, const vrt_options *opt)
{
    if (!out) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_scene_create: out is NULL");
    *out = nullptr;
    if (dim == 0) return fail(VRT_ERR_SCENE, "dimension is zero");                                  // image_util.cpp:508-511
    if (dim != 3) return fail(VRT_ERR_ILLEGAL_DIMENSION, "Illegal dimension: %d", dim);             // :559
    if (!bounds || !ior || !translucency) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_scene_create: NULL argument");
    if (ior_type != VRT_F32 && ior_type != VRT_U32)
        return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_scene_create: ior_type must be VRT_F32 or VRT_U32");
    const size_t prod = bounds[0] * bounds[1] * bounds[2];
    if (prod != nvox)                                                                               // :512-515
        return fail(VRT_ERR_SCENE, "imagesizes doesn't match:%zu=%zu=%zu (bounds ior translucency)", prod, nvox, nvox);
    for (int d = 0; d < 3; ++d)
        if (bounds[d] < 3 || bounds[d] > 65535)
            return fail(VRT_ERR_SCENE, "bounds[%d] = %zu: the 3x3x3 stencil needs 3 <= extent <= 65535", d, bounds[d]);

    // ior_log on the host with the libm the reference itself calls (image_util.cpp:531-543, 605-611);
    // for a LUT scene only the 256 table entries need a logarithm.
    const size_t nlog = density ? 256 : nvox;
    std::vector<uint32_t> lg(nlog);                     // float or int32 bit patterns
    std::atomic<int> bad{0};
    if (ior_type == VRT_F32) {
        const float *f = (const float *)ior;
        host_parallel_for(nlog, [&](size_t b, size_t e2) {
            for (size_t i = b; i < e2; ++i) {
                if (f[i] <= 0) { bad.store(1); continue; }
                const float v = (float)(std::log((double)f[i]) * (double)0x420000);
                memcpy(&lg[i], &v, 4);
            }
        });
        if (bad.load()) return fail(VRT_ERR_SCENE, "refraction-index underflow: ior <= 0");
    } else {
        const uint32_t *u = (const uint32_t *)ior;
        host_parallel_for(nlog, [&](size_t b, size_t e2) {
            for (size_t i = b; i < e2; ++i) {
                const double tmp = std::log((double)u[i] / 65536.0) * (double)0x420000;
                if (tmp > 2147483647.0) { bad.store(1); continue; }
                if (tmp < -2147483648.0) { bad.store(2); continue; }
                const int32_t v = (int32_t)std::round(tmp);
                memcpy(&lg[i], &v, 4);
            }
        });
        if (bad.load() == 1) return fail(VRT_ERR_SCENE, "refraction-index overflow");
        if (bad.load() == 2) return fail(VRT_ERR_SCENE, "refraction-index underflow");
    }

    const uint16_t fb[3] = {(uint16_t)(bounds[0] - 2), (uint16_t)(bounds[1] - 2), (uint16_t)(bounds[2] - 2)};
    const size_t fvox = (size_t)fb[0] * fb[1] * fb[2];
    const int field_type = ior_type == VRT_F32 ? VRT_F32 : VRT_I16;
    vrt_engine *e = nullptr;
    int rc = engine_alloc(&e, fb, field_type, fvox, opt);
    if (rc) return rc;
    DeviceGuard g(e->device);
    vrt_scene *sc = new (std::nothrow) vrt_scene();
    if (!sc) { engine_free(e); return fail(VRT_ERR_OUT_OF_MEMORY, "host allocation failed"); }
    sc->engine = e;
    sc->ior_type = ior_type;
    for (int d = 0; d < 3; ++d) sc->N[d] = (uint32_t)bounds[d];
    void *d_lg = nullptr, *d_tr = nullptr;
    auto cleanup = [&](int code) {
        if (d_lg) (void)hipFree(d_lg);
        if (d_tr) (void)hipFree(d_tr);
        vrt_scene_destroy(sc);
        return code;
    };
    rc = engine_init_device(e);
    if (rc) return cleanup(rc);
    hipStream_t s = e->stream;
    const size_t ntab = density ? 256 : nvox;
    if (hipMalloc(&d_lg, ntab * 4) != hipSuccess || hipMalloc(&d_tr, ntab * 4) != hipSuccess ||
        hipMalloc(&sc->d_err, sizeof(vrt::SceneErr)) != hipSuccess)
        return cleanup(fail(VRT_ERR_OUT_OF_MEMORY, "device allocation failed"));
    if (density) {
        if (hipMalloc(&sc->d_density, nvox) != hipSuccess || hipMalloc(&sc->d_lut_ior, 256 * 4) != hipSuccess)
            return cleanup(fail(VRT_ERR_OUT_OF_MEMORY, "device allocation failed"));
        if (hipMemcpyAsync(sc->d_density, density, nvox, hipMemcpyHostToDevice, s) != hipSuccess ||
            hipMemcpyAsync(sc->d_lut_ior, ior, 256 * 4, hipMemcpyHostToDevice, s) != hipSuccess)
            return cleanup(fail(VRT_ERR_HIP, "scene upload failed"));
    } else {
        if (hipMalloc(&sc->d_ior, nvox * 4) != hipSuccess) return cleanup(fail(VRT_ERR_OUT_OF_MEMORY, "device allocation failed"));
        if (hipMemcpyAsync(sc->d_ior, ior, nvox * 4, hipMemcpyHostToDevice, s) != hipSuccess)
            return cleanup(fail(VRT_ERR_HIP, "scene upload failed"));
    }
    if (hipMemcpyAsync(d_lg, lg.data(), ntab * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(d_tr, translucency, ntab * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemsetAsync(sc->d_err, 0, sizeof(vrt::SceneErr), s) != hipSuccess)
        return cleanup(fail(VRT_ERR_HIP, "scene upload failed"));
    uint32_t *d_tc = nullptr;
    if (e->opt.absorption) {
        if (e->transl.reserve(fvox * 4) != hipSuccess) return cleanup(fail(VRT_ERR_OUT_OF_MEMORY, "device allocation failed"));
        d_tc = (uint32_t *)e->transl.p;
    }
    const unsigned grid = (unsigned)((fvox + 255) / 256);
    vrt::SceneErr *derr = (vrt::SceneErr *)sc->d_err;
    const uint8_t *dd = (const uint8_t *)sc->d_density;
    if (ior_type == VRT_F32) {
        if (density)
            vrt::k_scene_build<<<grid, 256, 0, s>>>(vrt::SrcLut<float>{dd, (const float *)d_lg}, vrt::SrcLut<uint32_t>{dd, (const uint32_t *)d_tr},
                                                    sc->N[0], sc->N[1], sc->N[2], (vrt::vox_f32 *)e->field, derr, d_tc);
        else
            vrt::k_scene_build<<<grid, 256, 0, s>>>(vrt::SrcArray<float>{(const float *)d_lg}, vrt::SrcArray<uint32_t>{(const uint32_t *)d_tr},
                                                    sc->N[0], sc->N[1], sc->N[2], (vrt::vox_f32 *)e->field, derr, d_tc);
    } else {
        if (density)
            vrt::k_scene_build<<<grid, 256, 0, s>>>(vrt::SrcLut<int32_t>{dd, (const int32_t *)d_lg}, vrt::SrcLut<uint32_t>{dd, (const uint32_t *)d_tr},
                                                    sc->N[0], sc->N[1], sc->N[2], (vrt::vox_i16 *)e->field, derr, d_tc);
        else
            vrt::k_scene_build<<<grid, 256, 0, s>>>(vrt::SrcArray<int32_t>{(const int32_t *)d_lg}, vrt::SrcArray<uint32_t>{(const uint32_t *)d_tr},
                                                    sc->N[0], sc->N[1], sc->N[2], (vrt::vox_i16 *)e->field, derr, d_tc);
    }
    if (hipGetLastError() != hipSuccess) return cleanup(fail(VRT_ERR_HIP, "scene build launch failed"));
    vrt::SceneErr herr;
    if (hipMemcpyAsync(&herr, sc->d_err, sizeof herr, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
        return cleanup(fail(VRT_ERR_HIP, "scene build failed: %s", hipGetErrorString(hipGetLastError())));
    if (herr.diff_overflow) return cleanup(fail(VRT_ERR_SCENE, "differention overflow"));          // image_util.cpp:291-293
    (void)hipFree(d_lg);
    (void)hipFree(d_tr);
    d_lg = d_tr = nullptr;
    *out = sc;
    return VRT_OK;
}

int vrt_scene_create(vrt_scene_handle *out, int dim, const size_t *bounds, int ior_type, const void *ior,
                     const uint32_t *translucency, size_t nvox, const vrt_options *opt)
{
    return scene_create_impl(out, dim, bounds, ior_type, nullptr, ior, translucency, nvox, opt);
}

int vrt_scene_create_lut(vrt_scene_handle *out, int dim, const size_t *bounds, int ior_type, const uint8_t *density,
                         const void *lut_ior, const uint32_t *lut_translucency, size_t nvox, const vrt_options *opt)
{
    if (!density) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_scene_create_lut: density is NULL");
    return scene_create_impl(out, dim, bounds, ior_type, density, lut_ior, lut_translucency, nvox, opt);
}

int vrt_scene_trace(vrt_scene_handle sc, int dir_type, size_t nrays, const uint32_t *sp, const void *sd, uint32_t *ep,
                    void *ed, uint32_t *ei, uint32_t *rl, uint32_t *path, const float *invscale,
                    uint32_t min_brightness, uint32_t iterations, int /*normalize_length: "|| true", image_util.cpp:668*/,
                    const vrt_options *opt)
{
    if (!sc) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_scene_trace: NULL scene");
    vrt_engine *e = sc->engine;
    int rc = check_trace_args(e, dir_type, nrays, sp, sd, ep, ed, ei, rl, invscale, iterations);
    if (rc) return rc;
    if ((sc->ior_type == VRT_F32) != (dir_type == VRT_F32))
        return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_scene_trace: a %s scene takes %s directions (the reference instantiates "
                    "only <float,float> and <ior_t,dir_t>, image_util.cpp:940-956)",
                    sc->ior_type == VRT_F32 ? "float" : "fixed-point", sc->ior_type == VRT_F32 ? "float" : "int16");
    if (nrays == 0) return VRT_OK;
    DeviceGuard g(e->device);
    const size_t dsz = dir_type == VRT_F32 ? 4 : 2;
    auto pre = [&](uint32_t *d_pos, void *d_dir, size_t n, size_t off, hipStream_t s) -> int {
        HIP_TRY(sc->pro_pos.reserve(n * 12));
        HIP_TRY(sc->pro_dir.reserve(n * 3 * dsz));
        vrt::SceneErr init = {0u, 0xFFFFFFFFu, 0u};
        HIP_TRY(hipMemcpyAsync(sc->d_err, &init, sizeof init, hipMemcpyHostToDevice, s));
        const unsigned grid = (unsigned)((n + 255) / 256);
        vrt::SceneErr *derr = (vrt::SceneErr *)sc->d_err;
        const uint8_t *dd = (const uint8_t *)sc->d_density;
        if (dir_type == VRT_F32) {
            if (dd) vrt::k_ray_prologue<<<grid, 256, 0, s>>>(vrt::SrcLut<float>{dd, (const float *)sc->d_lut_ior}, sc->N[0], sc->N[1], sc->N[2], d_pos,
                                                             (const float *)d_dir, (uint32_t *)sc->pro_pos.p, (float *)sc->pro_dir.p, (uint32_t)n, derr);
            else    vrt::k_ray_prologue<<<grid, 256, 0, s>>>(vrt::SrcArray<float>{(const float *)sc->d_ior}, sc->N[0], sc->N[1], sc->N[2], d_pos,
                                                             (const float *)d_dir, (uint32_t *)sc->pro_pos.p, (float *)sc->pro_dir.p, (uint32_t)n, derr);
        } else {
            if (dd) vrt::k_ray_prologue<<<grid, 256, 0, s>>>(vrt::SrcLut<uint32_t>{dd, (const uint32_t *)sc->d_lut_ior}, sc->N[0], sc->N[1], sc->N[2], d_pos,
                                                             (const int16_t *)d_dir, (uint32_t *)sc->pro_pos.p, (int16_t *)sc->pro_dir.p, (uint32_t)n, derr);
            else    vrt::k_ray_prologue<<<grid, 256, 0, s>>>(vrt::SrcArray<uint32_t>{(const uint32_t *)sc->d_ior}, sc->N[0], sc->N[1], sc->N[2], d_pos,
                                                             (const int16_t *)d_dir, (uint32_t *)sc->pro_pos.p, (int16_t *)sc->pro_dir.p, (uint32_t)n, derr);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(d_pos, sc->pro_pos.p, n * 12, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_dir, sc->pro_dir.p, n * 3 * dsz, hipMemcpyDeviceToDevice, s));
        vrt::SceneErr herr;
        HIP_TRY(hipMemcpyAsync(&herr, sc->d_err, sizeof herr, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (herr.first_bad_ray != 0xFFFFFFFFu) {                                   // image_util.cpp:690-695
            const size_t i = off + herr.first_bad_ray;
            return fail(VRT_ERR_RAY_RANGE, "ray %zu:%g %g %g is not in 0 to %u %u %u", i, sp[3 * i] / 65536.0,
                        sp[3 * i + 1] / 65536.0, sp[3 * i + 2] / 65536.0, sc->N[0], sc->N[1], sc->N[2]);
        }
        if (herr.normalize_overflow)                                               // image_util.cpp:705-708
            return fail(VRT_ERR_SCENE, "Normalize length failed: -32768<=tmp<=32767");
        return VRT_OK;
    };
    return trace_host_impl(e, dir_type, nrays, sp, sd, ep, ed, ei, rl, path, invscale, iterations, opt, 0x10000u, min_brightness, pre);
}

void vrt_scene_destroy(vrt_scene_handle sc)
{
    if (!sc) return;
    if (sc->engine) {
        DeviceGuard g(sc->engine->device);
        if (sc->d_ior) (void)hipFree(sc->d_ior);
        if (sc->d_density) (void)hipFree(sc->d_density);
        if (sc->d_lut_ior) (void)hipFree(sc->d_lut_ior);
        if (sc->d_err) (void)hipFree(sc->d_err);
        sc->pro_pos.release();
        sc->pro_dir.release();
        engine_free(sc->engine);
    }
    delete sc;
}

vrt_handle vrt_scene_engine(vrt_scene_handle sc) { return sc ? sc->engine : nullptr; }

void vrt_destroy(vrt_handle h) { engine_free(h); }

int vrt_engine_device(vrt_handle h) { return h ? h->device : -1; }
size_t vrt_engine_field_bytes(vrt_handle h) { return h ? h->field_bytes : 0; }
const void *vrt_engine_field_device_ptr(vrt_handle h) { return h ? h->field : nullptr; }

int vrt_engine_debug_stats(vrt_handle h, unsigned long long *out, int n)
{
    if (!h || !out || n <= 0) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_engine_debug_stats: bad argument");
    if (!h->stats.p) return fail(VRT_ERR_INVALID_ARGUMENT, "debug stats not enabled (set VRT_DEBUG_STATS=1)");
    DeviceGuard g(h->device);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, h->stats.p, sizeof(unsigned long long) * std::min((size_t)n, h->stats_count), hipMemcpyDeviceToHost));
    return VRT_OK;
}

float vrt_engine_last_march_ms(vrt_handle h)
{
    if (!h || !h->ev_valid) return -1.0f;
    DeviceGuard g(h->device);
    float ms = -1.0f;
    if (hipEventSynchronize(h->ev_end) != hipSuccess) return -1.0f;
    if (hipEventElapsedTime(&ms, h->ev_begin, h->ev_end) != hipSuccess) return -1.0f;
    return ms;
}

int vrt_selftest_ilen2(int device, unsigned long long *mismatches, unsigned long long *admitted)
{
    if (!mismatches || !admitted) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_selftest_ilen2: null output");
    DeviceGuard g(device);
    if (!g.ok) return fail(VRT_ERR_NO_DEVICE, "vrt_selftest_ilen2: cannot select device");
    unsigned long long *d_out = nullptr, h_out[2] = {0, 0};
    HIP_TRY(hipMalloc(&d_out, sizeof(h_out)));
    hipError_t e = hipMemset(d_out, 0, sizeof(h_out));
    if (e == hipSuccess) { k_selftest_ilen2<<<512, 256>>>(d_out); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    HIP_TRY(e);
    *mismatches = h_out[0];
    *admitted = h_out[1];
    return VRT_OK;
}

}  // extern "C"
