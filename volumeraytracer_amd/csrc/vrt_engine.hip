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

#include <dlfcn.h>

#include <cmath>
#include <mutex>
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
    // device memory the L2s do not cache (MTYPE UC): for the few words and parked ray states that wavefronts on different
    // XCDs hand to each other inside one launch
    hipError_t reserve_uncached(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// roctx ranges ("roctx ranges around upload/kernel/gather") so that a rocprofv3 --marker-trace timeline
// of a call reads upload / ray order / march / download.  The marker library is looked up at run time: it is
// present under a profiler and otherwise not needed, and the engine must not depend on it.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        if (const char *off = getenv("VRT_NO_ROCTX"); off && off[0] == '1') return;
        for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            if (void *h = dlopen(name, RTLD_LAZY | RTLD_LOCAL)) {
                push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
                pop = (int (*)())dlsym(h, "roctxRangePop");
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};
const Roctx &roctx() { static Roctx r; return r; }
struct Range {
    bool on;
    explicit Range(const char *name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
    ~Range() { if (on) roctx().pop(); }
};

// Everything one in-flight trace call needs besides the field.  An engine owns kLanes of them: lane 0 serves the
// device-pointer entry points (any caller stream; calls on different streams are ordered by ev_done), all lanes serve
// the host-pointer entry points, whose chunks are pipelined over the lanes' own streams.
struct Workspace {
    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;   // around the most recent march kernel of this lane
    hipEvent_t ev_done = nullptr;                      // after the last launch that used these buffers
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // multi-device device-pointer calls: fan-out from / join into the caller's stream
    bool ev_valid = false, used = false;
    bool last_sliced = false;                          // the lane's most recent march used the time-sliced schedule
    int last_schedule = 0;                             // schedule of the lane's most recent march: 1 static, 2 persistent, 3 time-sliced; 0 = direct kernel / none
    // The time-sliced schedule bounds every wait and raises an error word instead of hanging (vrt_slab.h).  The word lives in
    // pinned host memory the kernel can write (a store happens only on that path: no cost otherwise) and stays raised until
    // somebody asks (vrt_trace_status / the host-pointer calls), so device-pointer callers can find out too.
    uint32_t *pin_sched_err = nullptr;                 // host address
    uint32_t *dev_sched_err = nullptr;                 // the same word as the device sees it
    // Schedule auto-tuning (vrt_options.schedule = 0): whether time slicing pays depends on how unevenly the groups of a
    // launch run, which is not known beforehand (config 5: 14.7 -> 13.7 ms; a 1448 x 1448 lattice on 512^3: 13.8 -> 15.5).
    // The first four launches of a workload shape alternate between the whole-group and the time-sliced schedule, HIP
    // events around each; once they have finished (hipEventQuery, never a wait) the schedule with the smaller minimum of
    // its two trials is kept for that shape (one trial each flipped the choice on a 3 % difference now and then).
    struct Tuner {
        unsigned long long key = 0;
        int state = 0;                                 // 0..3: trial launches (2, 3, 2, 3); 4: waiting for the events; 5: decided
        int choice = 2;
        int drift = 0;                                 // consecutive launches whose duration was > 15 % off the tuned one (state 5)
        hipEvent_t ev[4][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};   // [trial][begin | end]
        float ms[2] = {0.0f, 0.0f};
    } tuner;
    DevBuf sort_keys, sort_keys_alt, sort_vals, sort_vals_alt, sort_tmp, sort_bounds;
    DevBuf group_ctr;              // group counter of the persistent slab kernel (vrt_options.schedule)
    DevBuf park, slice_flags;      // time-sliced schedule: parked ray state (2 KiB per group) and per-group slice flags
    DevBuf clock;                  // (s_memtime, s_memrealtime) at the start and end of one wavefront of the march
    DevBuf stats;                  // debug counters, allocated only when VRT_DEBUG_STATS=1
    size_t stats_count = 0;
    DevBuf in_pos, in_dir, out_pos, out_dir, out_iter, out_light, path;   // device staging of host-pointer calls
    DevBuf pro_pos, pro_dir, err;  // scene layer: prologue outputs of device-pointer calls, error words
    DevBuf err_sticky;             // scene layer, device-pointer calls: error words accumulated over calls until vrt_scene_trace_status reads them
    void *pin_err = nullptr;       // pinned host copy of the error words (scene layer)
    std::vector<DevBuf *> all()
    {
        return {&sort_keys, &sort_keys_alt, &sort_vals, &sort_vals_alt, &sort_tmp, &sort_bounds, &group_ctr, &clock, &stats, &in_pos,
                &in_dir, &out_pos, &out_dir, &out_iter, &out_light, &path, &pro_pos, &pro_dir, &park, &slice_flags, &err, &err_sticky};
    }
};
constexpr int kLanes = 4;

}  // namespace

struct vrt_engine {
    int device = 0;
    int field_type = VRT_F32;
    uint32_t B[3] = {0, 0, 0};
    size_t nvox = 0;
    void *field = nullptr;         // interleaved AoS field, resident for the engine's lifetime
    size_t field_bytes = 0;
    int num_cu = 256;
    float gmax[3] = {INFINITY, INFINITY, INFINITY};   // max |field channel| over the whole field (proven rounds, vrt_slab.h: safe_steps)
    float gnorm = INFINITY;                            // >= max over the field of the gradient vector's length
    uint32_t field_flags = 0;                          // vrt::kFieldPrescaleOk once the field scan has shown it (engine_field_absmax)
    DevBuf transl;                 // translucency per field voxel, kept only for the opt-in absorption mode
    Workspace ws[kLanes];
    std::mutex mu;                 // device-pointer calls: one host thread at a time inside an engine
    // multi-device handles (vrt_options.devices): the field replicated on other devices, as the reference's ctor does
    // (cuda_volume_raytracer.cu:676-686); host-pointer calls cut the rays into one contiguous band per device.
    std::vector<vrt_engine *> replicas;
    bool peer_ok = true;           // replica only: its device can read and write the primary device's memory (device-pointer calls)
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

// ---- per-channel max |value| of the interleaved field (one pass at create; feeds MarchParams::gstep) -------------
// Non-negative floats order like their bit patterns, and a NaN's pattern is above infinity's: an atomic max on the
// patterns of |v| returns the largest magnitude, or a NaN if the field holds one (which then disables proven rounds).
template <typename VoxT>
__global__ __launch_bounds__(256) void k_field_absmax(const VoxT *f, size_t n, uint32_t *out /* [5]: |x|, |y|, |z|, x^2+y^2+z^2, ~(smallest non-zero |x|, |y| or |z|) */)
{
    uint32_t m[5] = {0u, 0u, 0u, 0u, 0u};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const VoxT v = f[i];
        const float a[3] = {(float)v.x, (float)v.y, (float)v.z};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const uint32_t b = __float_as_uint(a[d]) & 0x7FFFFFFFu;
            m[d] = b > m[d] ? b : m[d];
            const uint32_t c = ~(b - 1u);                // complement of (b - 1): a zero wraps to the largest b - 1, i.e. the smallest c
            m[4] = c > m[4] ? c : m[4];
        }
        const uint32_t q = __float_as_uint(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) & 0x7FFFFFFFu;   // squared length of the gradient vector
        m[3] = q > m[3] ? q : m[3];
    }
#pragma unroll
    for (int d = 0; d < 5; ++d) {
        for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)m[d], o); m[d] = t > m[d] ? t : m[d]; }
        if ((threadIdx.x & 63u) == 0) atomicMax(&out[d], m[d]);
    }
}

// the five words of the field scan (k_field_absmax, or fused into the tiled scene build) -> the engine's bounds and flags
void engine_apply_absmax(vrt_engine *e, const uint32_t h[5])
{
    memcpy(e->gmax, h, 3 * sizeof(uint32_t));
    float q;
    memcpy(&q, &h[3], 4);
    e->gnorm = (float)(std::sqrt((double)q) * (1.0 + 1.0 / 1024.0));     // >= the longest gradient vector (float sum of squares: 3 roundings)
    // prescale_ok (vrt_slab.h) for the whole field at once: every non-zero |value| in [2^-55, 2^78), which also rules out
    // NaN and infinity (their patterns are the largest); h[4] = ~(smallest non-zero pattern - 1), 0 for an all-zero field
    const uint32_t hi = std::max(h[0], std::max(h[1], h[2])), lo = h[4] ? ~h[4] + 1u : 0x7F800000u;
    e->field_flags = (e->field_type == VRT_I16 || (hi < ((127u + 78u) << 23) && lo >= ((127u - 55u) << 23))) ? vrt::kFieldPrescaleOk : 0u;
}

int engine_field_absmax(vrt_engine *e, hipStream_t s)
{
    uint32_t *d_out = nullptr, h[5] = {0, 0, 0, 0, 0};
    HIP_TRY(hipMalloc(&d_out, sizeof h));
    hipError_t err = hipMemsetAsync(d_out, 0, sizeof h, s);
    if (err == hipSuccess) {
        const int grid = (int)std::min<size_t>((e->nvox + 255) / 256, (size_t)e->num_cu * 8);
        if (e->field_type == VRT_F32) k_field_absmax<<<grid, 256, 0, s>>>((const vrt::vox_f32 *)e->field, e->nvox, d_out);
        else                          k_field_absmax<<<grid, 256, 0, s>>>((const vrt::vox_i16 *)e->field, e->nvox, d_out);
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipMemcpyAsync(h, d_out, sizeof h, hipMemcpyDeviceToHost, s);
    if (err == hipSuccess) err = hipStreamSynchronize(s);
    (void)hipFree(d_out);
    HIP_TRY(err);
    engine_apply_absmax(e, h);
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
    if (e->opt.devices == -1) e->opt.devices = (int32_t)(ndev >= 32 ? 0xFFFFFFFFu : ((1u << ndev) - 1u));
    // no primary named: the caller's current device (the mask only adds replicas), so that under one process per GPU
    // with every device visible each rank's handle still lives on its own device
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

int workspace_init(Workspace &w)
{
    HIP_TRY(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&w.ev_begin));
    HIP_TRY(hipEventCreate(&w.ev_end));
    HIP_TRY(hipEventCreateWithFlags(&w.ev_done, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&w.ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&w.ev_join, hipEventDisableTiming));
    HIP_TRY(hipHostMalloc((void **)&w.pin_sched_err, sizeof(uint32_t), hipHostMallocMapped));
    *w.pin_sched_err = 0u;
    HIP_TRY(hipHostGetDevicePointer((void **)&w.dev_sched_err, w.pin_sched_err, 0));
    return VRT_OK;
}

void workspace_free(Workspace &w)
{
    if (w.stream) { (void)hipStreamSynchronize(w.stream); (void)hipStreamDestroy(w.stream); }
    for (hipEvent_t ev : {w.ev_begin, w.ev_end, w.ev_done, w.ev_fork, w.ev_join}) if (ev) (void)hipEventDestroy(ev);
    for (auto &pair : w.tuner.ev) for (hipEvent_t ev : pair) if (ev) (void)hipEventDestroy(ev);
    for (DevBuf *b : w.all()) b->release();
    if (w.pin_err) (void)hipHostFree(w.pin_err);
    if (w.pin_sched_err) (void)hipHostFree(w.pin_sched_err);
    w = Workspace();
}

int engine_init_device(vrt_engine *e)
{
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, e->device));
    e->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    for (Workspace &w : e->ws) { const int rc = workspace_init(w); if (rc) return rc; }
    // +64 bytes of zeroed tail so that a 16-byte pair load at the very last voxel stays inside
    // the allocation (the march never reads it: the loop requires pos>>16 < bounds-1).
    HIP_TRY(hipMalloc(&e->field, e->field_bytes + 64));
    HIP_TRY(hipMemsetAsync((char *)e->field + e->field_bytes, 0, 64, e->ws[0].stream));
    return VRT_OK;
}

void engine_free(vrt_engine *e)
{
    if (!e) return;
    for (vrt_engine *r : e->replicas) engine_free(r);
    e->replicas.clear();
    DeviceGuard g(e->device);
    for (Workspace &w : e->ws) workspace_free(w);
    if (e->field) (void)hipFree(e->field);
    e->transl.release();
    delete e;
}

// ---- launch ---------------------------------------------------------------------------------

template <typename FieldT, typename DirT>
int launch_typed(vrt_engine *e, const vrt::MarchParams &p, int kernel, vrt::SlabGeometry geo, hipStream_t s)
{
    const bool path = p.path != nullptr;
    if (kernel == VRT_KERNEL_SLAB) {
        const hipError_t le = vrt::launch_slab<FieldT, DirT>(p, geo, e->num_cu, s);   // (hipGetLastError() clears the error it returns)
        return le == hipSuccess ? VRT_OK : fail(VRT_ERR_HIP, "slab kernel launch failed: %s", hipGetErrorString(le));
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

// Enqueue ray ordering + march for device-resident rays on stream `s`, using workspace `w` of engine `e`.
// The workspace (ray order, group counter, debug counters) is shared by every call on the lane: a call on another
// stream than the previous one first waits for that call's last kernel (ev_done), so two calls in flight on two
// streams of one engine are ordered instead of overwriting each other's permutation.
int trace_device_impl(vrt_engine *e, Workspace &w, int dir_type, size_t nrays, const uint32_t *d_sp, const void *d_sd,
                      uint32_t *d_ep, void *d_ed, uint32_t *d_ei, uint32_t *d_rl, uint32_t *d_path,
                      const float *invscale, uint32_t iterations, hipStream_t s, const vrt_options *opt,
                      uint32_t out_pos_bias = 0, uint32_t min_brightness = 0)
{
    if (nrays == 0) return VRT_OK;
    const vrt_options &o = opt ? *opt : e->opt;
    if (w.used) HIP_TRY(hipStreamWaitEvent(s, w.ev_done, 0));
    vrt::MarchParams p;
    p.field = e->field;
    p.Bx = e->B[0]; p.By = e->B[1]; p.Bz = e->B[2];
    p.invx = invscale[0]; p.invy = invscale[1]; p.invz = invscale[2];
    step_window(invscale, &p.win_lo, &p.win_span);
    p.field_flags = getenv("VRT_DEBUG_BRICK_TESTS") ? 0u : e->field_flags;   // (tests: force the per-brick prescale test)
    p.fast_math = o.fast_math ? 1u : 0u;
    for (int d = 0; d < 3; ++d) {                        // upper bound of |g_d * inv_d| per step, rounded up (safe_steps)
        const double g = (double)e->gmax[d] * std::fabs((double)invscale[d]) * (1.0 + 1.0 / 1048576.0);
        float gf = (float)g;
        if ((double)gf < g) gf = std::nextafterf(gf, INFINITY);
        p.gstep[d] = gf;                                 // NaN / inf stay what they are: no round is proven then
    }
    {   // the direction VECTOR changes by at most max|inv_d| x (longest gradient vector) per step: the interpolated gradient is
        // a convex combination of the corner vectors
        const double im = std::max(std::fabs((double)invscale[0]), std::max(std::fabs((double)invscale[1]), std::fabs((double)invscale[2])));
        const double g = (double)e->gnorm * im * (1.0 + 1.0 / 1048576.0);
        float gf = (float)g;
        if ((double)gf < g) gf = std::nextafterf(gf, INFINITY);
        p.gstep_norm = gf;
    }
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
        w.stats_count = 64 + 4 * (((size_t)nrays + 63) / 64);
        HIP_TRY(w.stats.reserve(w.stats_count * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(w.stats.p, 0, w.stats_count * sizeof(unsigned long long), s));
        p.stats = (unsigned long long *)w.stats.p;
    }
    int kernel = o.kernel;
    if (kernel == VRT_KERNEL_AUTO) kernel = d_path ? VRT_KERNEL_DIRECT : VRT_KERNEL_SLAB;
    if (kernel != VRT_KERNEL_DIRECT && kernel != VRT_KERNEL_SLAB)
        return fail(VRT_ERR_INVALID_ARGUMENT, "unknown kernel selector %d", kernel);
    if (kernel == VRT_KERNEL_SLAB && d_path) kernel = VRT_KERNEL_DIRECT;   // path tracing is write-bound: direct kernel

    // slab kernel geometry: LDS segment per wavefront -> wavefronts resident per CU (vrt_slab.h, slab_geometry)
    const vrt::SlabGeometry geo = vrt::slab_geometry(p.slab_lds_bytes, p.nrays, p.Bx, p.By, p.Bz);
    const size_t resident_waves = (size_t)e->num_cu * 4 * (size_t)geo.occ;

    // tile ordering: group rays that enter the field close together into the same wavefront
    const bool sort = o.sort_rays > 0 || (o.sort_rays == 0 && nrays >= 4096);
    // What the march needs zeroed per call -- the scheduler's counters, its queue ring, the clock words -- is zeroed by the
    // ray order's first kernel when there is one (vrt_sort.h, ZeroList): each was a hipMemsetAsync, i.e. a small launch, of its own.
    const size_t groups_all = (nrays + 63) / 64;
    unsigned long long ring_words = 64;                  // ring of re-queued groups: a power of two >= 2 x groups
    while (ring_words < 2ull * groups_all) ring_words <<= 1;
    bool zeroed = false, zeroed_ring = false;
    vrt::ZeroList zl = {{nullptr, nullptr, nullptr, nullptr}, {0u, 0u, 0u, 0u}};
    if (sort) {
        HIP_TRY(w.group_ctr.reserve(vrt::kCtrWords * sizeof(uint32_t)));
        HIP_TRY(w.clock.reserve(4 * sizeof(unsigned long long)));
        zl.p[0] = (uint32_t *)w.group_ctr.p; zl.words[0] = vrt::kCtrWords;
        zl.p[1] = (uint32_t *)w.clock.p;     zl.words[1] = 8u;
        zeroed = true;
        if (kernel == VRT_KERNEL_SLAB && geo.complete && groups_all > 1 && !p.transl && ring_words <= 0xFFFFFFFFull) {
            HIP_TRY(w.slice_flags.reserve_uncached((size_t)ring_words * sizeof(uint32_t)));
            zl.p[2] = (uint32_t *)w.slice_flags.p; zl.words[2] = (uint32_t)ring_words;
            zeroed_ring = true;
        }
    }
    if (sort) {
        Range range("vrt: ray order");
        const uint32_t *order = nullptr, *wave_order = nullptr;
        // longest wavefronts first: always for 3; in auto mode only when the launch has more wavefronts than the
        // GPU holds at once -- otherwise every wavefront starts at t = 0 anyway
        const int wave_lpt = !d_path && (o.sort_rays == 3 || (o.sort_rays == 0 && (nrays + 63) / 64 > resident_waves));
        hipError_t se = vrt::build_tile_order(d_sp, d_sd, dir_type, (uint32_t)nrays, e->B[0], e->B[1], e->B[2], o.sort_rays == 2 ? 1 : 0, w.sort_keys, w.sort_keys_alt,
                                              w.sort_vals, w.sort_vals_alt, w.sort_tmp, w.sort_bounds, s, &order,
                                              wave_lpt, iterations, &wave_order, e->field, e->field_type == VRT_I16, invscale, &zl);
        if (se != hipSuccess) return fail(se == hipErrorOutOfMemory ? VRT_ERR_OUT_OF_MEMORY : VRT_ERR_HIP,
                                          "ray tile ordering failed: %s", hipGetErrorString(se));
        p.order = order;
        p.wave_order = wave_order;
    }

    // scheduling of the slab kernel: static (one wavefront per group of 64 rays, hardware dispatch) or
    // persistent (as many wavefronts as the GPU holds, each drawing groups from a counter)
    p.group_counter = nullptr;
    int tuner_trial = -1;
    int launched_schedule = 0;
    {
        const size_t groups = (nrays + 63) / 64;
        int schedule = o.schedule;
        if (const char *ev = getenv("VRT_SCHEDULE"); ev && schedule == 0) schedule = atoi(ev);
        if (schedule < 0 || schedule > 3) return fail(VRT_ERR_INVALID_ARGUMENT, "unknown schedule %d", schedule);
        // auto: persistent once the launch has more wavefronts than the GPU holds (measured -4 % on config 3); whether the
        // persistent wavefronts take whole groups (2) or time slices (3) is settled by measurement (Workspace::Tuner)
        int trial = -1;                                  // 0 / 1: this launch is the timed trial of schedule 2 / 3
        if (schedule == 0) {
            schedule = groups > resident_waves ? 2 : 1;
            if (schedule == 2 && kernel == VRT_KERNEL_SLAB && geo.complete && !p.transl) {
                Workspace::Tuner &t = w.tuner;
                unsigned long long key = 1469598103934665603ull;
                auto mix = [&key](unsigned long long v) { key = (key ^ v) * 1099511628211ull; };
                mix(nrays); mix(iterations); mix((unsigned long long)dir_type); mix(geo.seg_bytes); mix((unsigned long long)o.sort_rays);
                mix((unsigned long long)p.fast_math);
                for (int d = 0; d < 3; ++d) { uint32_t b; memcpy(&b, &invscale[d], 4); mix(b); }
                if (t.key != key) { t.key = key; t.state = 0; t.choice = 2; }
                if (t.state < 4) {
                    schedule = (t.state & 1) ? 3 : 2;
                    trial = t.state;
                    for (int k = 0; k < 2; ++k) if (!t.ev[trial][k]) HIP_TRY(hipEventCreate(&t.ev[trial][k]));
                    ++t.state;
                } else if (t.state == 4) {
                    schedule = 2;
                    float ms[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    bool ready = true;
                    for (int k = 0; k < 4 && ready; ++k)
                        ready = hipEventQuery(t.ev[k][1]) == hipSuccess && hipEventElapsedTime(&ms[k], t.ev[k][0], t.ev[k][1]) == hipSuccess;
                    if (ready) {
                        t.ms[0] = std::min(ms[0], ms[2]);
                        t.ms[1] = std::min(ms[1], ms[3]);
                        t.choice = t.ms[1] < 0.99f * t.ms[0] ? 3 : 2;
                        t.state = 5;
                        t.drift = 0;
                        schedule = t.choice;
                        if (o.loglevel < 0) printf("schedule tuned: whole groups %.3f ms, time-sliced %.3f ms -> %d\n", t.ms[0], t.ms[1], t.choice);
                    }
                    (void)hipGetLastError();                 // hipErrorNotReady from the queries is not an error
                } else {
                    schedule = t.choice;
                    // The key cannot see the rays themselves: another ray set of the same size and type would keep a schedule
                    // that was timed on a different work distribution.  The previous launch's duration is free to read once
                    // it has finished; three launches in a row more than 15 % off the tuned time start the trials again.
                    float prev = 0.0f;
                    if (w.ev_valid && hipEventQuery(w.ev_end) == hipSuccess && hipEventElapsedTime(&prev, w.ev_begin, w.ev_end) == hipSuccess) {
                        const float ref = t.ms[t.choice - 2];
                        t.drift = (ref > 0.0f && (prev > 1.15f * ref || prev < 0.85f * ref)) ? t.drift + 1 : 0;
                        if (t.drift >= 3) { t.state = 0; t.drift = 0; }
                    }
                    (void)hipGetLastError();
                }
            }
        }
        p.park = nullptr; p.slice_flags = nullptr; p.slice_steps = 0; p.ticket_limit = 0;
        p.work_sum = nullptr; p.work_unit = (uint32_t)vrt::kCoarse; p.slice_percent = 115;
        p.sched_err = w.dev_sched_err;
        p.spin_limit = 1u << 22;
        if (const char *ev = getenv("VRT_DEBUG_SPIN_LIMIT")) p.spin_limit = (uint32_t)std::max(1L, strtol(ev, nullptr, 10));   // tests: force the give-up branch
        launched_schedule = kernel == VRT_KERNEL_SLAB ? 1 : 0;
        if (schedule >= 2 && kernel == VRT_KERNEL_SLAB && groups > 1 && geo.complete) {
            HIP_TRY(w.group_ctr.reserve(vrt::kCtrWords * sizeof(uint32_t)));
            if (!zeroed) HIP_TRY(hipMemsetAsync(w.group_ctr.p, 0, vrt::kCtrWords * sizeof(uint32_t), s));
            p.group_counter = (uint32_t *)w.group_ctr.p;
            tuner_trial = trial;
            launched_schedule = 2;
            if (schedule == 3 && !p.transl) {
                launched_schedule = 3;
                // time-sliced: parked state and per-group slice flags (vrt_slab.h, SCHED == 2)
                // slice length: derived in the kernel from the pre-march's work estimates when the launch has them (the wavefront
                // launch order was built), else 1024 wave-steps; VRT_SLICE_STEPS / VRT_SLICE_PERCENT override (development)
                uint32_t slice = p.wave_order ? 0u : 1024u;
                if (const char *ev = getenv("VRT_SLICE_STEPS")) slice = (uint32_t)std::max(16L, strtol(ev, nullptr, 10));
                if (const char *ev = getenv("VRT_SLICE_PERCENT")) p.slice_percent = (uint32_t)std::max(10L, strtol(ev, nullptr, 10));
                if (p.wave_order) p.work_sum = &((const vrt::RayBounds *)w.sort_bounds.p)->work_sum;
                // ring of re-queued groups: a power of two >= 2 x groups (at most one word per group is ever occupied)
                const unsigned long long cap = ring_words;
                HIP_TRY(w.park.reserve_uncached(groups * 2048));
                HIP_TRY(w.slice_flags.reserve_uncached((size_t)cap * sizeof(uint32_t)));
                if (!zeroed_ring) HIP_TRY(hipMemsetAsync(w.slice_flags.p, 0, (size_t)cap * sizeof(uint32_t), s));
                p.park = w.park.p;
                p.slice_flags = (uint32_t *)w.slice_flags.p;
                p.slice_steps = slice;
                p.ticket_limit = (uint32_t)cap;
            }
        }
    }

    // one wavefront of the march stamps (s_memtime, s_memrealtime) when it starts and when it ends: the clock the
    // kernel really ran at, for the roofline's peak (vrt_engine_last_march_clock_ghz)
    HIP_TRY(w.clock.reserve(4 * sizeof(unsigned long long)));
    if (!zeroed) HIP_TRY(hipMemsetAsync(w.clock.p, 0, 4 * sizeof(unsigned long long), s));
    p.clock_out = (unsigned long long *)w.clock.p;

    Range range("vrt: march");
    HIP_TRY(hipEventRecord(w.ev_begin, s));
    if (tuner_trial >= 0) HIP_TRY(hipEventRecord(w.tuner.ev[tuner_trial][0], s));
    int rc;
    if (e->field_type == VRT_F32)
        rc = dir_type == VRT_F32 ? launch_typed<float, float>(e, p, kernel, geo, s) : launch_typed<float, int16_t>(e, p, kernel, geo, s);
    else
        rc = dir_type == VRT_F32 ? launch_typed<int16_t, float>(e, p, kernel, geo, s) : launch_typed<int16_t, int16_t>(e, p, kernel, geo, s);
    if (rc) return rc;
    if (tuner_trial >= 0) HIP_TRY(hipEventRecord(w.tuner.ev[tuner_trial][1], s));
    HIP_TRY(hipEventRecord(w.ev_end, s));
    HIP_TRY(hipEventRecord(w.ev_done, s));
    w.ev_valid = true;
    w.used = true;
    w.last_sliced = p.park != nullptr;
    w.last_schedule = launched_schedule;
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

// ---- host-pointer calls ---------------------------------------------------------------------------------------
// The reference copies each 0x8000-ray chunk up, runs it, waits and copies it back, one blocking step after the other
// (cuda_volume_raytracer.cu:843, 905-910).  Here a call is cut into a few large chunks that are pipelined over the
// engine's lanes (each lane: own stream, own staging buffers, own host thread): while lane A marches chunk k, lane B
// uploads chunk k+1 and lane C downloads chunk k-1, so PCIe time hides behind the march.  With several devices
// (vrt_options.devices) the rays are first cut into one contiguous band per device.  Results land directly in the
// caller's arrays at the rays' own indices, so the output is independent of the cut.
struct HostCall {
    int dir_type;
    const uint32_t *sp; const void *sd;
    uint32_t *ep; void *ed; uint32_t *ei; uint32_t *rl; uint32_t *path;
    const float *invscale;
    uint32_t iterations, min_brightness, out_pos_bias;
    const vrt_options *opt;
};

// `pre` (optional) rewrites the uploaded start arrays on the device before the march (the scene layer's ray
// prologue) and `post` inspects what it left behind once the chunk's stream has drained.
struct ChunkHooks {
    virtual int pre(vrt_engine *, Workspace &, uint32_t *, void *, size_t, size_t, hipStream_t) { return VRT_OK; }
    virtual int post(vrt_engine *, Workspace &, size_t, size_t) { return VRT_OK; }
    virtual ~ChunkHooks() {}
};

int trace_host_chunk(vrt_engine *e, Workspace &w, const HostCall &c, size_t off, size_t n, ChunkHooks &hooks)
{
    const size_t dsz = c.dir_type == VRT_F32 ? 4 : 2;
    hipStream_t s = w.stream;
    HIP_TRY(w.in_pos.reserve(n * 12));
    HIP_TRY(w.in_dir.reserve(n * 3 * dsz));
    HIP_TRY(w.out_pos.reserve(n * 12));
    HIP_TRY(w.out_dir.reserve(n * 3 * dsz));
    HIP_TRY(w.out_iter.reserve(n * 4));
    HIP_TRY(w.out_light.reserve(n * 4));
    if (c.path) HIP_TRY(w.path.reserve(n * (size_t)c.iterations * 12));
    {
        Range range("vrt: upload rays");
        HIP_TRY(hipMemcpyAsync(w.in_pos.p, c.sp + 3 * off, n * 12, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(w.in_dir.p, (const char *)c.sd + 3 * off * dsz, n * 3 * dsz, hipMemcpyHostToDevice, s));
    }
    // the lane's error words and prologue buffers are shared with a device-pointer call that may still be in flight on
    // another stream: order behind it before the prologue touches them
    if (w.used) HIP_TRY(hipStreamWaitEvent(s, w.ev_done, 0));
    int rc = hooks.pre(e, w, (uint32_t *)w.in_pos.p, w.in_dir.p, n, off, s);
    if (rc) return rc;
    rc = trace_device_impl(e, w, c.dir_type, n, (const uint32_t *)w.in_pos.p, w.in_dir.p, (uint32_t *)w.out_pos.p,
                           w.out_dir.p, (uint32_t *)w.out_iter.p, (uint32_t *)w.out_light.p,
                           c.path ? (uint32_t *)w.path.p : nullptr, c.invscale, c.iterations, s, c.opt, c.out_pos_bias,
                           c.min_brightness);
    if (rc) return rc;
    // ray errors (scene layer) and scheduler errors are looked at before anything is copied back, so that a failing call
    // leaves the caller's output arrays untouched, as the reference's exceptions do (image_util.cpp:690-708 throw before the march)
    HIP_TRY(hipStreamSynchronize(s));
    if (const uint32_t gave_up = *(volatile uint32_t *)w.pin_sched_err) {
        *w.pin_sched_err = 0u;
        return fail(VRT_ERR_HIP, "time-sliced schedule: a wavefront gave up waiting (%u); results are incomplete", gave_up);
    }
    rc = hooks.post(e, w, n, off);
    if (rc) return rc;
    {
        Range range("vrt: download results");
        HIP_TRY(hipMemcpyAsync(c.ep + 3 * off, w.out_pos.p, n * 12, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync((char *)c.ed + 3 * off * dsz, w.out_dir.p, n * 3 * dsz, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(c.ei + off, w.out_iter.p, n * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(c.rl + off, w.out_light.p, n * 4, hipMemcpyDeviceToHost, s));
        if (c.path)
            HIP_TRY(hipMemcpyAsync(c.path + off * (size_t)c.iterations * 3, w.path.p, n * (size_t)c.iterations * 12,
                                   hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return VRT_OK;
}

// Contiguous cut of `n` rays into `parts` ranges whose sizes are multiples of 64 where possible (whole wavefronts),
// first ranges taking the remainder.  Exported for tests through vrt_debug_band_cut.
void band_cut(size_t n, size_t parts, std::vector<size_t> &begin)
{
    begin.assign(parts + 1, n);
    const size_t groups = (n + 63) / 64;
    size_t g0 = 0;
    for (size_t k = 0; k < parts; ++k) {
        begin[k] = std::min(n, g0 * 64);
        g0 += groups / parts + (k < groups % parts ? 1 : 0);
    }
    begin[parts] = n;
}

// Engines (devices) of a handle that one call of `nrays` rays is spread over.  Calls with no more than Options::_minimum_gpu
// rays stay on one device (the reference keeps them on the CPU, cuda_volume_raytracer.cu:805); device-pointer calls can
// only use replicas whose device reaches the primary's memory (peer access).
std::vector<vrt_engine *> call_engines(vrt_engine *e0, size_t nrays, const vrt_options &o, bool need_peer)
{
    std::vector<vrt_engine *> engs{e0};
    for (vrt_engine *r : e0->replicas) if (!need_peer || r->peer_ok) engs.push_back(r);
    if (nrays < 64 * engs.size() || nrays <= o.minimum_gpu) engs.resize(1);   // too few rays to be worth spreading
    return engs;
}

// Device-pointer call on a handle that spans several devices (the reference spreads every trace_rays_cu call over all
// devices, cuda_volume_raytracer.cu:804-912): one contiguous band of the rays per device, each band ordered and marched by
// its device's engine on that engine's own stream.  The arrays stay where the caller put them, on the primary device: the
// other devices read their band's start rays and write its results straight through peer access (xGMI), at the rays' own
// indices, so there is neither a host round trip nor a merge pass.  The caller's stream forks into the replica streams
// through an event and joins them again: when the call returns, everything is enqueued and `s` completes after all bands.
int trace_device_multi(const std::vector<vrt_engine *> &engs, int dir_type, size_t nrays, const uint32_t *d_sp, const void *d_sd,
                       uint32_t *d_ep, void *d_ed, uint32_t *d_ei, uint32_t *d_rl, uint32_t *d_path, const float *invscale,
                       uint32_t iterations, hipStream_t s, const vrt_options *opt, uint32_t out_pos_bias, uint32_t min_brightness)
{
    vrt_engine *e0 = engs[0];
    if (engs.size() == 1)
        return trace_device_impl(e0, e0->ws[0], dir_type, nrays, d_sp, d_sd, d_ep, d_ed, d_ei, d_rl, d_path, invscale, iterations, s,
                                 opt, out_pos_bias, min_brightness);
    std::vector<size_t> band;
    band_cut(nrays, engs.size(), band);
    const size_t dsz = dir_type == VRT_F32 ? 4 : 2;
    HIP_TRY(hipEventRecord(e0->ws[0].ev_fork, s));
    auto run_band = [&](vrt_engine *e, size_t k, hipStream_t sk) {
        const size_t off = band[k], n = band[k + 1] - band[k];
        if (n == 0) return (int)VRT_OK;
        return trace_device_impl(e, e->ws[0], dir_type, n, d_sp + 3 * off, (const char *)d_sd + 3 * off * dsz, d_ep + 3 * off,
                                 (char *)d_ed + 3 * off * dsz, d_ei + off, d_rl + off,
                                 d_path ? d_path + off * (size_t)iterations * 3 : nullptr, invscale, iterations, sk, opt, out_pos_bias,
                                 min_brightness);
    };
    for (size_t k = 1; k < engs.size(); ++k) {              // the replicas first: their host-side enqueue overlaps nothing otherwise
        vrt_engine *r = engs[k];
        DeviceGuard g(r->device);
        hipStream_t sk = r->ws[0].stream;
        HIP_TRY(hipStreamWaitEvent(sk, e0->ws[0].ev_fork, 0));
        const int rc = run_band(r, k, sk);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(r->ws[0].ev_join, sk));
    }
    const int rc = run_band(e0, 0, s);
    if (rc) return rc;
    for (size_t k = 1; k < engs.size(); ++k) HIP_TRY(hipStreamWaitEvent(s, engs[k]->ws[0].ev_join, 0));
    HIP_TRY(hipEventRecord(e0->ws[0].ev_done, s));           // "the call is complete" for the next call and for vrt_trace_status
    return VRT_OK;
}

// chunks per device for the lane pipeline
size_t host_chunks(const vrt_options &o, size_t nrays, bool path)
{
    if (path) return 1;                                  // path tracing is chunked by the device path budget instead
    int want = o.host_lanes;
    if (const char *ev = getenv("VRT_HOST_LANES"); ev && want == 0) want = atoi(ev);
    // auto = one chunk.  Measured on BASELINE config 3 (tools/host_path.py, 10^6 rays, 24 MB up + 32 MB down from pageable
    // memory): 1 chunk 10.5 ms, 2 chunks 10.7, 4 chunks 13.4, 8 chunks 15.7 against 8.9 ms for the device-resident call --
    // every chunk pays its own ray ordering and its own launch tail, and concurrent marches share one GPU, which costs
    // more than the 1.6 ms of copies a pipeline can hide.  The knob stays for callers whose copies dominate.
    if (want <= 0) want = 1;
    return (size_t)std::min(want, 64);
}

int trace_host_impl(vrt_engine *e0, const HostCall &c, size_t nrays, ChunkHooks &hooks)
{
    const vrt_options &o = c.opt ? *c.opt : e0->opt;
    // one band per device ...
    std::vector<vrt_engine *> engs = call_engines(e0, nrays, o, false);
    std::vector<size_t> band;
    band_cut(nrays, engs.size(), band);

    struct Task { vrt_engine *e; int lane; size_t off, n; };
    std::vector<std::vector<Task>> per_thread;             // one host thread per (device, lane)
    size_t nchunks_total = 0;
    for (size_t d = 0; d < engs.size(); ++d) {
        const size_t b0 = band[d], bn = band[d + 1] - band[d];
        if (bn == 0) continue;
        std::vector<size_t> cut;
        if (c.path) {
            // the device path buffer (iterations*12 bytes per ray) bounds the chunk, as the reference's 0x8000-ray
            // chunks bound its own (.cu:798, 840)
            const size_t per_ray = (size_t)c.iterations * 12;
            size_t budget = (size_t)2 << 30;
            if (const char *b = getenv("VRT_PATH_BUDGET_BYTES")) {          // tests shrink it to force several chunks
                const unsigned long long v = strtoull(b, nullptr, 10);
                if (v > 0) budget = (size_t)v;
            }
            const size_t chunk = std::max<size_t>(1, std::min(bn, budget / std::max<size_t>(per_ray, 1)));
            for (size_t x = 0; x < bn; x += chunk) cut.push_back(x);
            cut.push_back(bn);
        } else {
            band_cut(bn, host_chunks(o, bn, false), cut);
        }
        const size_t nch = cut.size() - 1;
        const int lanes = c.path ? 1 : (int)std::min<size_t>(kLanes, nch);
        const size_t first = per_thread.size();
        per_thread.resize(first + lanes);
        for (size_t k = 0; k < nch; ++k)
            if (cut[k + 1] > cut[k]) {
                per_thread[first + k % lanes].push_back(Task{engs[d], (int)(k % lanes), b0 + cut[k], cut[k + 1] - cut[k]});
                ++nchunks_total;
            }
    }

    std::vector<int> rcs(per_thread.size(), VRT_OK);
    std::vector<std::string> errs(per_thread.size());
    std::atomic<size_t> progress{0};
    auto run = [&](size_t t) {
        for (const Task &k : per_thread[t]) {
            DeviceGuard g(k.e->device);
            if (o.loglevel < 0) printf("iteration %zu of %zu\n", progress.fetch_add(1), nchunks_total);     // .cu:823-826
            const int rc = trace_host_chunk(k.e, k.e->ws[k.lane], c, k.off, k.n, hooks);
            if (rc) { rcs[t] = rc; errs[t] = g_err; return; }
        }
    };
    if (per_thread.size() == 1) run(0);
    else {
        std::vector<std::thread> th;
        for (size_t t = 0; t < per_thread.size(); ++t) th.emplace_back(run, t);
        for (auto &x : th) x.join();
    }
    for (size_t t = 0; t < rcs.size(); ++t)
        if (rcs[t]) { g_err = errs[t]; return rcs[t]; }                 // the worker's thread-local message, on this thread
    bool cap_hit = false;
    for (size_t i = 0; i < nrays && !cap_hit; ++i) cap_hit = (c.ei[i] == c.iterations);
    if (cap_hit) printf("Warning, maximum iterations hitted\n");                   // .cu:512-515
    if (o.loglevel < 0) printf("cpu: 0 gpu: %zu\n", nchunks_total);               // .cu:948-951
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

struct SceneSrc {                   // what the ray prologue interpolates the ior from, resident on one device
    void *d_ior = nullptr;          // the ior volume
    void *d_density = nullptr;      // LUT scenes: the 8-bit volume instead, with its ior table
    void *d_lut_ior = nullptr;
};
struct vrt_scene {
    vrt_engine *engine = nullptr;
    int ior_type = VRT_F32;
    uint32_t N[3] = {0, 0, 0};
    std::vector<SceneSrc> src;      // [0] on the primary device, [k] on the device of engine->replicas[k - 1]
    bool device_call_pending = false;   // a vrt_scene_trace_device call whose error words have not been looked at yet
    const SceneSrc &source_of(const vrt_engine *e) const
    {
        for (size_t k = 0; k < engine->replicas.size(); ++k) if (engine->replicas[k] == e && k + 1 < src.size()) return src[k + 1];
        return src[0];
    }
};

// =============================================================================================
// C ABI
// =============================================================================================


// Exhaustive self-test of the march step's short division (vrt_march.h: ilen_short / div_window):
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
        bad += __float_as_uint(vrt::ilen_short(d)) != __float_as_uint(q);
    }
    for (int o = 32; o > 0; o >>= 1) { bad += __shfl_xor(bad, o); adm += __shfl_xor(adm, o); }
    if ((threadIdx.x & 63u) == 0) { atomicAdd(&out[0], bad); atomicAdd(&out[1], adm); }
}

// Exhaustive self-test of the march step's position update (vrt_march.h: add_round): every float s with |s| < 2^31 (both
// signs, biased exponents 0..157, all mantissas) against std::round + cvttss2si as the reference computes the step (.cu:347).
__global__ __launch_bounds__(256) void k_selftest_round(unsigned long long *out /* [2]: mismatches, checked */)
{
    const uint32_t expo = blockIdx.x & 255u, sign = blockIdx.x >> 8;
    unsigned long long bad = 0, n = 0;
    const uint32_t pos0 = 0x12345678u + blockIdx.x * 0x9E3779B9u;
    for (uint32_t m = threadIdx.x; m < (1u << 23); m += blockDim.x) {
        const float v = __uint_as_float((sign << 31) | (expo << 23) | m);
        if (expo <= 157u) {
            ++n;
            bad += vrt::add_round(pos0, v) != pos0 + (uint32_t)vrt::round_to_i32(v);
        }
    }
    for (int o = 32; o > 0; o >>= 1) { bad += __shfl_xor(bad, o); n += __shfl_xor(n, o); }
    if ((threadIdx.x & 63u) == 0) { atomicAdd(&out[0], bad); atomicAdd(&out[1], n); }
}

// Exhaustive self-test of the staging loop's division by multiplication (vrt_slab.h: stage_magic / div22 / nmad24): every
// divisor 2 <= d <= 1024 (one per block) with every dividend n < 1024 against the integer quotient and remainder.
__global__ __launch_bounds__(256) void k_selftest_stage_division(unsigned long long *out /* [2]: mismatches, checked */)
{
    const uint32_t d = blockIdx.x + 2u;
    const uint32_t m = vrt::stage_magic(d);
    unsigned long long bad = 0, n_checked = 0;
    for (uint32_t n = threadIdx.x; n < 1024u; n += blockDim.x) {
        const uint32_t q = vrt::div22(n, m), r = vrt::nmad24(q, d, n);
        bad += q != n / d;
        bad += r != n % d;
        ++n_checked;
    }
    for (int o = 32; o > 0; o >>= 1) { bad += __shfl_xor(bad, o); n_checked += __shfl_xor(n_checked, o); }
    if ((threadIdx.x & 63u) == 0) { atomicAdd(&out[0], bad); atomicAdd(&out[1], n_checked); }
}

// Devices a new handle spans: vrt_options.devices is a bit mask of HIP ordinals (0 = just the primary, -1 = every visible
// device); the primary (vrt_options.device, or the caller's current device) comes first.  VRT_DEBUG_REPLICAS=n (tests on a
// one-GPU box) turns a handle that would have a single device into n engines on that device, so that the band cut, the
// fork / join of the streams and the merge run without peers.
std::vector<int> handle_devices(const vrt_options &o, int primary, int ndev)
{
    std::vector<int> devs{primary};
    if (o.devices != 0)
        for (int d = 0; d < ndev && d < 32; ++d)
            if (d != primary && ((uint32_t)o.devices >> d & 1u)) devs.push_back(d);
    if (const char *ev = getenv("VRT_DEBUG_REPLICAS"); ev && atoi(ev) > 1 && devs.size() == 1)
        devs.assign((size_t)std::min(atoi(ev), 16), primary);
    return devs;
}

// Replicate a finished engine's field (and translucency) onto the other devices of the handle, as the reference's
// constructor does with cudaMemcpy from the host copy (cuda_volume_raytracer.cu:676-686) -- here device to device, and to
// all peers at once: every copy is issued on its destination's stream before any of them is waited for, so the 7 xGMI links
// of the primary carry the 7 copies side by side instead of one after the other.
int engine_replicate(vrt_engine *e)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 1;
    const std::vector<int> devs = handle_devices(e->opt, e->device, ndev);
    const uint16_t b16[3] = {(uint16_t)e->B[0], (uint16_t)e->B[1], (uint16_t)e->B[2]};
    for (size_t k = 1; k < devs.size(); ++k) {
        vrt_engine *r = nullptr;
        vrt_options ro = e->opt;
        ro.device = devs[k];
        ro.devices = 0;
        int rc = engine_alloc(&r, b16, e->field_type, e->nvox, &ro);
        if (rc) return rc;
        e->replicas.push_back(r);                       // owned by e from here on (engine_free walks the list)
        DeviceGuard g(r->device);
        rc = engine_init_device(r);
        if (rc) return rc;
        memcpy(r->gmax, e->gmax, sizeof r->gmax);
        r->gnorm = e->gnorm;
        r->field_flags = e->field_flags;
        hipStream_t s = r->ws[0].stream;
        if (r->device != e->device) {
            // both directions: the replica reads the primary's memory (field copy, the caller's ray arrays of device-pointer
            // calls) and the primary's streams wait on the replica's events
            int can = 0;
            r->peer_ok = hipDeviceCanAccessPeer(&can, r->device, e->device) == hipSuccess && can != 0;
            if (r->peer_ok) {
                const hipError_t pe = hipDeviceEnablePeerAccess(e->device, 0);
                r->peer_ok = pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled;
            }
            (void)hipGetLastError();                     // "already enabled" is fine
            HIP_TRY(hipMemcpyPeerAsync(r->field, r->device, e->field, e->device, e->field_bytes, s));
        } else {
            HIP_TRY(hipMemcpyAsync(r->field, e->field, e->field_bytes, hipMemcpyDeviceToDevice, s));
        }
        if (e->transl.p) {
            HIP_TRY(r->transl.reserve(e->nvox * 4));
            if (r->device != e->device) HIP_TRY(hipMemcpyPeerAsync(r->transl.p, r->device, e->transl.p, e->device, e->nvox * 4, s));
            else HIP_TRY(hipMemcpyAsync(r->transl.p, e->transl.p, e->nvox * 4, hipMemcpyDeviceToDevice, s));
        }
    }
    for (vrt_engine *r : e->replicas) {                  // all copies are in flight: now wait for them
        DeviceGuard g(r->device);
        HIP_TRY(hipStreamSynchronize(r->ws[0].stream));
    }
    return VRT_OK;
}

// copy `bytes` of device memory of engine `from` into a fresh allocation on engine `to`'s device (scene sources)
int replicate_buffer(vrt_engine *from, vrt_engine *to, const void *src, size_t bytes, void **out)
{
    *out = nullptr;
    if (!src) return VRT_OK;
    DeviceGuard g(to->device);
    HIP_TRY(hipMalloc(out, bytes));
    if (to->device != from->device) HIP_TRY(hipMemcpyPeerAsync(*out, to->device, src, from->device, bytes, to->ws[0].stream));
    else HIP_TRY(hipMemcpyAsync(*out, src, bytes, hipMemcpyDeviceToDevice, to->ws[0].stream));
    return VRT_OK;
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
    opt->devices = 0;
    opt->host_lanes = 0;
    opt->fast_math = 0;
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
        else if (hipMemcpyAsync(e->transl.p, d_tr, nvox * 4, hipMemcpyDeviceToDevice, e->ws[0].stream) != hipSuccess)
            rc = fail(VRT_ERR_HIP, "translucency copy failed");
    }
    if (!rc) rc = launch_interleave(e, d_diff, d_tr, 0, nvox, e->ws[0].stream);
    if (!rc && hipStreamSynchronize(e->ws[0].stream) != hipSuccess) rc = fail(VRT_ERR_HIP, "field interleave failed");
    if (!rc) rc = engine_field_absmax(e, e->ws[0].stream);
    if (!rc) rc = engine_replicate(e);
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
            hipMemcpyAsync((uint32_t *)e->transl.p + off, tr + off, n * 4, hipMemcpyHostToDevice, e->ws[0].stream) != hipSuccess)
            return cleanup(fail(VRT_ERR_HIP, "field upload failed"));
        for (int k = 0; k < 3; ++k)
            if (hipMemcpyAsync(tmp[k].p, (const char *)diff[k] + off * esz, n * esz, hipMemcpyHostToDevice, e->ws[0].stream) != hipSuccess)
                return cleanup(fail(VRT_ERR_HIP, "field upload failed"));
        if (hipMemcpyAsync(tmp[3].p, tr + off, n * 4, hipMemcpyHostToDevice, e->ws[0].stream) != hipSuccess)
            return cleanup(fail(VRT_ERR_HIP, "field upload failed"));
        const void *d[3] = {tmp[0].p, tmp[1].p, tmp[2].p};
        rc = launch_interleave(e, d, (const uint32_t *)tmp[3].p, off, n, e->ws[0].stream);
        if (rc) return cleanup(rc);
        if (hipStreamSynchronize(e->ws[0].stream) != hipSuccess) return cleanup(fail(VRT_ERR_HIP, "field interleave failed"));
    }
    for (DevBuf &b : tmp) b.release();
    rc = engine_field_absmax(e, e->ws[0].stream);
    if (!rc) rc = engine_replicate(e);
    if (rc) { engine_free(e); return rc; }
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
    std::lock_guard<std::mutex> lock(h->mu);
    // min_brightness is ignored unless the opt-in absorption mode is on, as in the reference (.cu:785)
    return trace_device_multi(call_engines(h, nrays, opt ? *opt : h->opt, true), dir_type, nrays, d_sp, d_sd, d_ep, d_ed, d_ei, d_rl,
                              d_path, invscale, iterations, (hipStream_t)stream, opt, 0, min_brightness);
}

int vrt_trace(vrt_handle h, int dir_type, size_t nrays, const uint32_t *sp, const void *sd, uint32_t *ep, void *ed,
              uint32_t *ei, uint32_t *rl, uint32_t *path, const float *invscale, uint32_t min_brightness,
              uint32_t iterations, const vrt_options *opt)
{
    int rc = check_trace_args(h, dir_type, nrays, sp, sd, ep, ed, ei, rl, invscale, iterations);
    if (rc) return rc;
    if (nrays == 0) return VRT_OK;
    DeviceGuard g(h->device);
    std::lock_guard<std::mutex> lock(h->mu);
    const HostCall c{dir_type, sp, sd, ep, ed, ei, rl, path, invscale, iterations, min_brightness, 0u, opt};
    ChunkHooks none;
    return trace_host_impl(h, c, nrays, none);
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
    sc->src.resize(1);
    sc->ior_type = ior_type;
    for (int d = 0; d < 3; ++d) sc->N[d] = (uint32_t)bounds[d];
    void *d_lg = nullptr, *d_tr = nullptr, *d_err = nullptr;
    auto cleanup = [&](int code) {
        if (d_lg) (void)hipFree(d_lg);
        if (d_tr) (void)hipFree(d_tr);
        if (d_err) (void)hipFree(d_err);
        vrt_scene_destroy(sc);
        return code;
    };
    rc = engine_init_device(e);
    if (rc) return cleanup(rc);
    hipStream_t s = e->ws[0].stream;
    const size_t ntab = density ? 256 : nvox;
    if (hipMalloc(&d_lg, ntab * 4) != hipSuccess || hipMalloc(&d_tr, ntab * 4) != hipSuccess ||
        hipMalloc(&d_err, sizeof(vrt::SceneErr)) != hipSuccess)
        return cleanup(fail(VRT_ERR_OUT_OF_MEMORY, "device allocation failed"));
    if (density) {
        if (hipMalloc(&sc->src[0].d_density, nvox) != hipSuccess || hipMalloc(&sc->src[0].d_lut_ior, 256 * 4) != hipSuccess)
            return cleanup(fail(VRT_ERR_OUT_OF_MEMORY, "device allocation failed"));
        if (hipMemcpyAsync(sc->src[0].d_density, density, nvox, hipMemcpyHostToDevice, s) != hipSuccess ||
            hipMemcpyAsync(sc->src[0].d_lut_ior, ior, 256 * 4, hipMemcpyHostToDevice, s) != hipSuccess)
            return cleanup(fail(VRT_ERR_HIP, "scene upload failed"));
    } else {
        if (hipMalloc(&sc->src[0].d_ior, nvox * 4) != hipSuccess) return cleanup(fail(VRT_ERR_OUT_OF_MEMORY, "device allocation failed"));
        if (hipMemcpyAsync(sc->src[0].d_ior, ior, nvox * 4, hipMemcpyHostToDevice, s) != hipSuccess)
            return cleanup(fail(VRT_ERR_HIP, "scene upload failed"));
    }
    if (hipMemcpyAsync(d_lg, lg.data(), ntab * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(d_tr, translucency, ntab * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemsetAsync(d_err, 0, sizeof(vrt::SceneErr), s) != hipSuccess)
        return cleanup(fail(VRT_ERR_HIP, "scene upload failed"));
    uint32_t *d_tc = nullptr;
    if (e->opt.absorption) {
        if (e->transl.reserve(fvox * 4) != hipSuccess) return cleanup(fail(VRT_ERR_OUT_OF_MEMORY, "device allocation failed"));
        d_tc = (uint32_t *)e->transl.p;
    }
    vrt::SceneErr *derr = (vrt::SceneErr *)d_err;
    const uint8_t *dd = (const uint8_t *)sc->src[0].d_density;
    // LDS-tiled build with the field scan fused in (vrt_scene.h); VRT_SCENE_BUILD=pervoxel selects the round-2 kernel
    // (one thread per voxel, 27 global loads) for A/B runs and for the test that holds the two to the same bits
    bool tiled = true;
    if (const char *ev = getenv("VRT_SCENE_BUILD")) tiled = strcmp(ev, "pervoxel") != 0;
    // runs aligned to four voxels of the field's linear index (k_scene_build_tiled): possible when the shift of a row is the same in every plane
    uint32_t zshift = 0;
    {
        const unsigned long long oy = sc->N[1] - 2, oz = sc->N[2] - 2;
        if ((oy * oz) % 4ull == 0 && oz % 4ull != 0 && !(getenv("VRT_SCENE_ZSHIFT") && atoi(getenv("VRT_SCENE_ZSHIFT")) == 0)) zshift = vrt::kZS;
    }
    const dim3 tgrid((sc->N[2] - 2 + zshift + vrt::kTZ - 1) / vrt::kTZ, (sc->N[1] - 2 + vrt::kTY - 1) / vrt::kTY, (sc->N[0] - 2 + vrt::kXS - 1) / vrt::kXS);
    const size_t nwg = (size_t)tgrid.x * tgrid.y * tgrid.z;
    const unsigned grid = (unsigned)((fvox + 255) / 256);
    void *d_part = nullptr;
    if (tiled && (tgrid.y > 65535u || tgrid.z > 65535u)) tiled = false;
    if (tiled && hipMalloc(&d_part, (nwg + 1) * 5 * sizeof(uint32_t)) != hipSuccess) return cleanup(fail(VRT_ERR_OUT_OF_MEMORY, "device allocation failed"));
    uint32_t *part = (uint32_t *)d_part;
    Range range("vrt: scene build");
#define VRT_BUILD(LOGSRC, TRSRC, VOX)                                                                                        \
    do {                                                                                                                     \
        if (tiled) vrt::k_scene_build_tiled<<<tgrid, vrt::kTY * vrt::kTZ, 0, s>>>(LOGSRC, TRSRC, sc->N[0], sc->N[1], sc->N[2], (VOX *)e->field, derr, d_tc, part, zshift); \
        else       vrt::k_scene_build<<<grid, 256, 0, s>>>(LOGSRC, TRSRC, sc->N[0], sc->N[1], sc->N[2], (VOX *)e->field, derr, d_tc);          \
    } while (0)
    if (ior_type == VRT_F32) {
        if (density) VRT_BUILD((vrt::SrcLut<float>{dd, (const float *)d_lg}), (vrt::SrcLut<uint32_t>{dd, (const uint32_t *)d_tr}), vrt::vox_f32);
        else         VRT_BUILD((vrt::SrcArray<float>{(const float *)d_lg}), (vrt::SrcArray<uint32_t>{(const uint32_t *)d_tr}), vrt::vox_f32);
    } else {
        if (density) VRT_BUILD((vrt::SrcLut<int32_t>{dd, (const int32_t *)d_lg}), (vrt::SrcLut<uint32_t>{dd, (const uint32_t *)d_tr}), vrt::vox_i16);
        else         VRT_BUILD((vrt::SrcArray<int32_t>{(const int32_t *)d_lg}), (vrt::SrcArray<uint32_t>{(const uint32_t *)d_tr}), vrt::vox_i16);
    }
#undef VRT_BUILD
    if (const hipError_t le = hipGetLastError(); le != hipSuccess) {
        if (d_part) (void)hipFree(d_part);
        return cleanup(fail(VRT_ERR_HIP, "scene build launch failed: %s", hipGetErrorString(le)));
    }
    uint32_t habs[5] = {0, 0, 0, 0, 0};
    hipError_t be = hipSuccess;
    if (tiled) {
        be = hipMemsetAsync(part + nwg * 5, 0, 5 * sizeof(uint32_t), s);
        const unsigned rgrid = (unsigned)std::min<size_t>((nwg + 255) / 256, 128);
        if (be == hipSuccess) { vrt::k_absmax_reduce<<<rgrid, 256, 0, s>>>(part, nwg, part + nwg * 5); be = hipGetLastError(); }
        if (be == hipSuccess) be = hipMemcpyAsync(habs, part + nwg * 5, sizeof habs, hipMemcpyDeviceToHost, s);
    }
    vrt::SceneErr herr;
    if (be == hipSuccess) be = hipMemcpyAsync(&herr, d_err, sizeof herr, hipMemcpyDeviceToHost, s);
    if (be == hipSuccess) be = hipStreamSynchronize(s);
    if (d_part) (void)hipFree(d_part);
    if (be != hipSuccess) return cleanup(fail(VRT_ERR_HIP, "scene build failed: %s", hipGetErrorString(be)));
    if (herr.diff_overflow) return cleanup(fail(VRT_ERR_SCENE, "differention overflow"));          // image_util.cpp:291-293
    if (tiled) engine_apply_absmax(e, habs);
    else rc = engine_field_absmax(e, s);
    if (rc) return cleanup(rc);
    (void)hipFree(d_lg);
    (void)hipFree(d_tr);
    (void)hipFree(d_err);
    d_lg = d_tr = d_err = nullptr;
    // a scene over several devices (vrt_options.devices): the field and what the ray prologue reads, on every one of them
    // (RaytraceScene constructs its TraceRaysCu over all CUDA devices, image_util.cpp:561 -> cuda_volume_raytracer.cu:676-686)
    rc = engine_replicate(e);
    if (rc) return cleanup(rc);
    sc->src.resize(1 + e->replicas.size());
    for (size_t k = 0; k < e->replicas.size() && !rc; ++k) {
        vrt_engine *r = e->replicas[k];
        SceneSrc &dst = sc->src[k + 1];
        rc = replicate_buffer(e, r, sc->src[0].d_ior, nvox * 4, &dst.d_ior);
        if (!rc) rc = replicate_buffer(e, r, sc->src[0].d_density, nvox, &dst.d_density);
        if (!rc) rc = replicate_buffer(e, r, sc->src[0].d_lut_ior, 256 * 4, &dst.d_lut_ior);
    }
    for (vrt_engine *r : e->replicas) {
        DeviceGuard gr(r->device);
        if (hipStreamSynchronize(r->ws[0].stream) != hipSuccess && !rc) rc = fail(VRT_ERR_HIP, "scene replication failed");
    }
    if (rc) return cleanup(rc);
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

}  // extern "C"

namespace {

// The scene layer's ray front end (RaytraceScene::trace_rays, image_util.cpp:673-722) as chunk hooks: the prologue
// kernel rewrites the start arrays in place on the chunk's stream, the march follows without any host round trip, and
// the error words (range check, normalisation overflow) come back with the results and are looked at afterwards.
// `sticky`: the error words are the lane's accumulating set (device-pointer calls: kept over calls until
// vrt_scene_trace_status reads them) instead of the per-call set that travels back through pinned memory (host-pointer calls).
int scene_prologue(vrt_scene *sc, const SceneSrc &src, Workspace &w, int dir_type, const uint32_t *d_pos_in, const void *d_dir_in,
                   uint32_t *d_pos_out, void *d_dir_out, size_t n, hipStream_t s, bool sticky)
{
    vrt::SceneErr *derr = nullptr;
    if (sticky) {
        if (!w.err_sticky.p) {
            HIP_TRY(w.err_sticky.reserve(sizeof(vrt::SceneErr)));
            HIP_TRY(hipMemsetAsync(w.err_sticky.p, 0, sizeof(vrt::SceneErr), s));
            HIP_TRY(hipMemsetAsync((char *)w.err_sticky.p + offsetof(vrt::SceneErr, first_bad_ray), 0xFF, 4, s));
        }
        derr = (vrt::SceneErr *)w.err_sticky.p;
    } else {
        HIP_TRY(w.err.reserve(sizeof(vrt::SceneErr)));
        if (!w.pin_err) HIP_TRY(hipHostMalloc(&w.pin_err, sizeof(vrt::SceneErr), hipHostMallocDefault));
        HIP_TRY(hipMemsetAsync(w.err.p, 0, sizeof(vrt::SceneErr), s));
        HIP_TRY(hipMemsetAsync((char *)w.err.p + offsetof(vrt::SceneErr, first_bad_ray), 0xFF, 4, s));
        derr = (vrt::SceneErr *)w.err.p;
    }
    const unsigned grid = (unsigned)((n + 255) / 256);
    const uint8_t *dd = (const uint8_t *)src.d_density;
    Range range("vrt: ray prologue");
    if (dir_type == VRT_F32) {
        if (dd) vrt::k_ray_prologue<<<grid, 256, 0, s>>>(vrt::SrcLut<float>{dd, (const float *)src.d_lut_ior}, sc->N[0], sc->N[1], sc->N[2], d_pos_in,
                                                         (const float *)d_dir_in, d_pos_out, (float *)d_dir_out, (uint32_t)n, derr);
        else    vrt::k_ray_prologue<<<grid, 256, 0, s>>>(vrt::SrcArray<float>{(const float *)src.d_ior}, sc->N[0], sc->N[1], sc->N[2], d_pos_in,
                                                         (const float *)d_dir_in, d_pos_out, (float *)d_dir_out, (uint32_t)n, derr);
    } else {
        if (dd) vrt::k_ray_prologue<<<grid, 256, 0, s>>>(vrt::SrcLut<uint32_t>{dd, (const uint32_t *)src.d_lut_ior}, sc->N[0], sc->N[1], sc->N[2], d_pos_in,
                                                         (const int16_t *)d_dir_in, d_pos_out, (int16_t *)d_dir_out, (uint32_t)n, derr);
        else    vrt::k_ray_prologue<<<grid, 256, 0, s>>>(vrt::SrcArray<uint32_t>{(const uint32_t *)src.d_ior}, sc->N[0], sc->N[1], sc->N[2], d_pos_in,
                                                         (const int16_t *)d_dir_in, d_pos_out, (int16_t *)d_dir_out, (uint32_t)n, derr);
    }
    HIP_TRY(hipGetLastError());
    if (!sticky) HIP_TRY(hipMemcpyAsync(w.pin_err, w.err.p, sizeof(vrt::SceneErr), hipMemcpyDeviceToHost, s));   // pinned: truly asynchronous
    return VRT_OK;
}

struct SceneHooks : ChunkHooks {
    vrt_scene *sc;
    int dir_type;
    std::atomic<unsigned long long> first_bad{~0ull};
    std::atomic<int> normalize_overflow{0};
    SceneHooks(vrt_scene *s, int dt) : sc(s), dir_type(dt) {}
    int pre(vrt_engine *e, Workspace &w, uint32_t *d_pos, void *d_dir, size_t n, size_t, hipStream_t s) override
    {
        return scene_prologue(sc, sc->source_of(e), w, dir_type, d_pos, d_dir, d_pos, d_dir, n, s, false);
    }
    int post(vrt_engine *, Workspace &w, size_t, size_t off) override
    {
        const vrt::SceneErr *h = (const vrt::SceneErr *)w.pin_err;
        if (h->first_bad_ray != 0xFFFFFFFFu) {
            unsigned long long i = off + h->first_bad_ray, cur = first_bad.load();
            while (i < cur && !first_bad.compare_exchange_weak(cur, i)) {}
            return VRT_ERR_RAY_RANGE;
        }
        if (h->normalize_overflow) { normalize_overflow.store(1); return VRT_ERR_SCENE; }
        return VRT_OK;
    }
};

int scene_check_trace_args(vrt_scene *sc, int dir_type)
{
    if ((sc->ior_type == VRT_F32) != (dir_type == VRT_F32))
        return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_scene_trace: a %s scene takes %s directions (the reference instantiates "
                    "only <float,float> and <ior_t,dir_t>, image_util.cpp:940-956)",
                    sc->ior_type == VRT_F32 ? "float" : "fixed-point", sc->ior_type == VRT_F32 ? "float" : "int16");
    return VRT_OK;
}

}  // namespace

extern "C" {

int vrt_scene_trace(vrt_scene_handle sc, int dir_type, size_t nrays, const uint32_t *sp, const void *sd, uint32_t *ep,
                    void *ed, uint32_t *ei, uint32_t *rl, uint32_t *path, const float *invscale,
                    uint32_t min_brightness, uint32_t iterations, int /*normalize_length: "|| true", image_util.cpp:668*/,
                    const vrt_options *opt)
{
    if (!sc) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_scene_trace: NULL scene");
    vrt_engine *e = sc->engine;
    int rc = check_trace_args(e, dir_type, nrays, sp, sd, ep, ed, ei, rl, invscale, iterations);
    if (rc) return rc;
    rc = scene_check_trace_args(sc, dir_type);
    if (rc) return rc;
    if (nrays == 0) return VRT_OK;
    {   // the range check first and on the host, as the reference does it (image_util.cpp:673-697 throws before anything is
        // traced, so the outputs stay untouched); the device prologue repeats it for the device-pointer entry point
        std::atomic<unsigned long long> first_bad{~0ull};
        const unsigned long long lim[3] = {(unsigned long long)sc->N[0] * 0x10000ull, (unsigned long long)sc->N[1] * 0x10000ull,
                                           (unsigned long long)sc->N[2] * 0x10000ull};
        host_parallel_for(nrays, [&](size_t b, size_t e2) {
            for (size_t i = b; i < e2; ++i) {
                bool bad = false;
                for (int d = 0; d < 3; ++d) bad |= sp[3 * i + d] < 0x10000u || (unsigned long long)sp[3 * i + d] + 1ull >= lim[d];
                if (bad) {
                    unsigned long long cur = first_bad.load();
                    while (i < cur && !first_bad.compare_exchange_weak(cur, i)) {}
                    return;                                     // this slice's first bad ray is its lowest
                }
            }
        });
        if (first_bad.load() != ~0ull) {
            const size_t i = (size_t)first_bad.load();
            return fail(VRT_ERR_RAY_RANGE, "ray %zu:%g %g %g is not in 0 to %u %u %u", i, sp[3 * i] / 65536.0,
                        sp[3 * i + 1] / 65536.0, sp[3 * i + 2] / 65536.0, sc->N[0], sc->N[1], sc->N[2]);
        }
    }
    DeviceGuard g(e->device);
    std::lock_guard<std::mutex> lock(e->mu);
    const HostCall c{dir_type, sp, sd, ep, ed, ei, rl, path, invscale, iterations, min_brightness, 0x10000u, opt};
    SceneHooks hooks(sc, dir_type);
    rc = trace_host_impl(e, c, nrays, hooks);
    if (hooks.first_bad.load() != ~0ull) {                                          // image_util.cpp:690-695
        const size_t i = (size_t)hooks.first_bad.load();
        return fail(VRT_ERR_RAY_RANGE, "ray %zu:%g %g %g is not in 0 to %u %u %u", i, sp[3 * i] / 65536.0,
                    sp[3 * i + 1] / 65536.0, sp[3 * i + 2] / 65536.0, sc->N[0], sc->N[1], sc->N[2]);
    }
    if (hooks.normalize_overflow.load())                                            // image_util.cpp:705-708
        return fail(VRT_ERR_SCENE, "Normalize length failed: -32768<=tmp<=32767");
    return rc;
}

int vrt_scene_trace_device(vrt_scene_handle sc, int dir_type, size_t nrays, const uint32_t *d_sp, const void *d_sd,
                           uint32_t *d_ep, void *d_ed, uint32_t *d_ei, uint32_t *d_rl, uint32_t *d_path,
                           const float *invscale, uint32_t min_brightness, uint32_t iterations, void *stream,
                           const vrt_options *opt)
{
    if (!sc) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_scene_trace_device: NULL scene");
    vrt_engine *e = sc->engine;
    int rc = check_trace_args(e, dir_type, nrays, d_sp, d_sd, d_ep, d_ed, d_ei, d_rl, invscale, iterations);
    if (rc) return rc;
    rc = scene_check_trace_args(sc, dir_type);
    if (rc) return rc;
    if (nrays == 0) return VRT_OK;
    DeviceGuard g(e->device);
    std::lock_guard<std::mutex> lock(e->mu);
    Workspace &w = e->ws[0];
    hipStream_t s = (hipStream_t)stream;
    const size_t dsz = dir_type == VRT_F32 ? 4 : 2;
    if (w.used) HIP_TRY(hipStreamWaitEvent(s, w.ev_done, 0));      // the prologue buffers below belong to the lane too
    HIP_TRY(w.pro_pos.reserve(nrays * 12));
    HIP_TRY(w.pro_dir.reserve(nrays * 3 * dsz));
    // the prologue is O(rays) and runs on the primary device for all of them; the bands of a multi-device scene then read
    // its output through peer access like any other device-resident start arrays
    rc = scene_prologue(sc, sc->src[0], w, dir_type, d_sp, d_sd, (uint32_t *)w.pro_pos.p, w.pro_dir.p, nrays, s, true);
    if (rc) return rc;
    sc->device_call_pending = true;
    return trace_device_multi(call_engines(e, nrays, opt ? *opt : e->opt, true), dir_type, nrays, (const uint32_t *)w.pro_pos.p,
                              w.pro_dir.p, d_ep, d_ed, d_ei, d_rl, d_path, invscale, iterations, s, opt, 0x10000u, min_brightness);
}

int vrt_scene_trace_status(vrt_scene_handle sc)
{
    if (!sc) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_scene_trace_status: NULL scene");
    vrt_engine *e = sc->engine;
    DeviceGuard g(e->device);
    vrt::SceneErr h{0u, 0xFFFFFFFFu, 0u};
    {
        std::lock_guard<std::mutex> lock(e->mu);
        if (!sc->device_call_pending) return VRT_OK;
        Workspace &w = e->ws[0];
        HIP_TRY(hipEventSynchronize(w.ev_done));
        sc->device_call_pending = false;
        if (w.err_sticky.p) {
            // the accumulated error words of every device-pointer call since the last status call (lowest bad ray index of
            // any of them), then cleared
            HIP_TRY(hipMemcpy(&h, w.err_sticky.p, sizeof h, hipMemcpyDeviceToHost));
            const vrt::SceneErr clear{0u, 0xFFFFFFFFu, 0u};
            HIP_TRY(hipMemcpy(w.err_sticky.p, &clear, sizeof clear, hipMemcpyHostToDevice));
        }
    }
    if (h.first_bad_ray != 0xFFFFFFFFu)
        return fail(VRT_ERR_RAY_RANGE, "ray %u is not in 0 to %u %u %u", h.first_bad_ray, sc->N[0], sc->N[1], sc->N[2]);
    if (h.normalize_overflow) return fail(VRT_ERR_SCENE, "Normalize length failed: -32768<=tmp<=32767");
    return vrt_trace_status(e);                                     // scheduler errors of the march itself
}

void vrt_scene_destroy(vrt_scene_handle sc)
{
    if (!sc) return;
    if (sc->engine) {
        for (size_t k = 0; k < sc->src.size(); ++k) {
            const vrt_engine *owner = k == 0 ? sc->engine : (k - 1 < sc->engine->replicas.size() ? sc->engine->replicas[k - 1] : sc->engine);
            DeviceGuard g(owner->device);
            for (void *q : {sc->src[k].d_ior, sc->src[k].d_density, sc->src[k].d_lut_ior}) if (q) (void)hipFree(q);
        }
        engine_free(sc->engine);
    }
    delete sc;
}

vrt_handle vrt_scene_engine(vrt_scene_handle sc) { return sc ? sc->engine : nullptr; }

void vrt_destroy(vrt_handle h) { engine_free(h); }

int vrt_engine_device(vrt_handle h) { return h ? h->device : -1; }
size_t vrt_engine_field_bytes(vrt_handle h) { return h ? h->field_bytes : 0; }
const void *vrt_engine_field_device_ptr(vrt_handle h) { return h ? h->field : nullptr; }

int vrt_engine_replica_count(vrt_handle h) { return h ? 1 + (int)h->replicas.size() : 0; }

int vrt_engine_last_schedule(vrt_handle h) { return h ? h->ws[0].last_schedule : 0; }

int vrt_engine_download_field(vrt_handle h, void *dst, size_t bytes)
{
    if (!h || !dst) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_engine_download_field: NULL argument");
    if (bytes > h->field_bytes) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_engine_download_field: %zu bytes asked, the field has %zu", bytes, h->field_bytes);
    DeviceGuard g(h->device);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(dst, h->field, bytes, hipMemcpyDeviceToHost));
    return VRT_OK;
}

int vrt_engine_field_scan(vrt_handle h, float *bounds4, uint32_t *flags)
{
    if (!h || !bounds4 || !flags) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_engine_field_scan: NULL argument");
    for (int d = 0; d < 3; ++d) bounds4[d] = h->gmax[d];
    bounds4[3] = h->gnorm;
    *flags = h->field_flags;
    return VRT_OK;
}

int vrt_trace_status(vrt_handle h)
{
    if (!h) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_trace_status: NULL handle");
    std::lock_guard<std::mutex> lock(h->mu);
    std::vector<vrt_engine *> engs{h};
    for (vrt_engine *r : h->replicas) engs.push_back(r);
    uint32_t gave_up = 0;
    for (vrt_engine *e : engs) {
        DeviceGuard g(e->device);
        for (Workspace &w : e->ws) {
            if (!w.used) continue;
            HIP_TRY(hipEventSynchronize(w.ev_done));
            gave_up += *(volatile uint32_t *)w.pin_sched_err;
            *w.pin_sched_err = 0u;
        }
    }
    if (gave_up) return fail(VRT_ERR_HIP, "time-sliced schedule: a wavefront gave up waiting (%u); results are incomplete", gave_up);
    return VRT_OK;
}

int vrt_engine_debug_stats(vrt_handle h, unsigned long long *out, int n)
{
    if (!h || !out || n <= 0) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_engine_debug_stats: bad argument");
    Workspace &w = h->ws[0];
    if (!w.stats.p) return fail(VRT_ERR_INVALID_ARGUMENT, "debug stats not enabled (set VRT_DEBUG_STATS=1)");
    DeviceGuard g(h->device);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, w.stats.p, sizeof(unsigned long long) * std::min((size_t)n, w.stats_count), hipMemcpyDeviceToHost));
    if (n >= 64 && w.group_ctr.p) {                        // scheduler words of the last launch: tickets, groups finished, error
        uint32_t c[vrt::kCtrWords] = {0};
        HIP_TRY(hipMemcpy(c, w.group_ctr.p, std::min<size_t>(sizeof c, w.group_ctr.cap), hipMemcpyDeviceToHost));
        for (int k = 0; k < 4; ++k) out[60 + k] = c[k * vrt::kCtrStride];
    }
    return VRT_OK;
}

float vrt_engine_last_march_ms(vrt_handle h)
{
    if (!h || !h->ws[0].ev_valid) return -1.0f;
    DeviceGuard g(h->device);
    float ms = -1.0f;
    if (hipEventSynchronize(h->ws[0].ev_end) != hipSuccess) return -1.0f;
    if (hipEventElapsedTime(&ms, h->ws[0].ev_begin, h->ws[0].ev_end) != hipSuccess) return -1.0f;
    return ms;
}

float vrt_engine_last_march_clock_ghz(vrt_handle h)
{
    if (!h || !h->ws[0].ev_valid || !h->ws[0].clock.p) return -1.0f;
    DeviceGuard g(h->device);
    unsigned long long c[4] = {0, 0, 0, 0};
    if (hipEventSynchronize(h->ws[0].ev_end) != hipSuccess) return -1.0f;
    if (hipMemcpy(c, h->ws[0].clock.p, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) return -1.0f;
    if (c[3] <= c[1] || c[2] <= c[0]) return -1.0f;                   // (s_memrealtime counts at 100 MHz)
    return (float)((double)(c[2] - c[0]) / (double)(c[3] - c[1]) * 0.1);
}

int vrt_debug_band_cut(size_t n, size_t parts, size_t *begin /* parts + 1 */)
{
    if (!begin || parts == 0) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_debug_band_cut: bad argument");
    std::vector<size_t> b;
    band_cut(n, parts, b);
    for (size_t k = 0; k <= parts; ++k) begin[k] = b[k];
    return VRT_OK;
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

int vrt_selftest_round(int device, unsigned long long *mismatches, unsigned long long *checked)
{
    if (!mismatches || !checked) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_selftest_round: null output");
    DeviceGuard g(device);
    if (!g.ok) return fail(VRT_ERR_NO_DEVICE, "vrt_selftest_round: cannot select device");
    unsigned long long *d_out = nullptr, h_out[2] = {0, 0};
    HIP_TRY(hipMalloc(&d_out, sizeof(h_out)));
    hipError_t e = hipMemset(d_out, 0, sizeof(h_out));
    if (e == hipSuccess) { k_selftest_round<<<512, 256>>>(d_out); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    HIP_TRY(e);
    *mismatches = h_out[0];
    *checked = h_out[1];
    return VRT_OK;
}

int vrt_selftest_stage_division(int device, unsigned long long *mismatches, unsigned long long *checked)
{
    if (!mismatches || !checked) return fail(VRT_ERR_INVALID_ARGUMENT, "vrt_selftest_stage_division: null output");
    DeviceGuard g(device);
    if (!g.ok) return fail(VRT_ERR_NO_DEVICE, "vrt_selftest_stage_division: cannot select device");
    unsigned long long *d_out = nullptr, h_out[2] = {0, 0};
    HIP_TRY(hipMalloc(&d_out, sizeof(h_out)));
    hipError_t e = hipMemset(d_out, 0, sizeof(h_out));
    if (e == hipSuccess) { k_selftest_stage_division<<<1023, 256>>>(d_out); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    HIP_TRY(e);
    *mismatches = h_out[0];
    *checked = h_out[1];
    return VRT_OK;
}

}  // extern "C"
