/*
 * vrt.h -- C ABI of the MI355X-native trace_rays engine (libvrt_hip.so).
 *
 * This is the drop-in boundary for the reference's ray-march seam.  The reference has
 * no C ABI: its seam is the C++ class TraceRaysCu<DiffType>
 * (/root/reference/src/cuda_volume_raytracer.h:61-115), whose object file exports
 * exactly six symbols to the scene layer (image_util.cpp:561, 630, 752, 777).  Each
 * entry point below names the reference interface it replaces.  INTEGRATION.md shows
 * the C++ shim that re-creates those six symbols on top of this header, plus the
 * pybind11 / JNI / ctypes bindings.
 *
 * Conventions: plain pointers and sizes only; all buffers caller-owned; functions
 * return VRT_OK (0) or a negative vrt_status and never throw; the message of the last
 * failure on the calling thread is available from vrt_last_error().  There is no CPU
 * fallback: without a usable gfx950 device vrt_create fails with VRT_ERR_NO_DEVICE.
 *
 * Data conventions (identical to the reference seam):
 *   positions   uint32 16.16 fixed point, 3 per ray (x,y,z), in FIELD coordinates (the
 *               caller has already applied the -0x10000 shift, image_util.cpp:690-709)
 *   directions  VRT_F32: float, unit 1.0;  VRT_I16: int16 8.8 fixed, unit 0x100 (types.h:41-49)
 *   field       dim+1 = 4 channels per voxel {dx,dy,dz,w}, index ((x*By)+y)*Bz+z
 *               (cuda_volume_raytracer.cu:111-113, 456-466)
 */
#ifndef VRT_H
#define VRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VRT_ABI_VERSION 3

/* element types: DiffType / DirType of the reference's template instantiations
 * (cuda_volume_raytracer.cu:992-1049) */
#define VRT_F32 0
#define VRT_I16 1

typedef enum vrt_status {
    VRT_OK = 0,
    VRT_ERR_INVALID_ARGUMENT = -1,
    VRT_ERR_ILLEGAL_DIMENSION = -2, /* "Illegal dimension", cuda_volume_raytracer.cu:770 */
    VRT_ERR_HIP = -3,               /* HIP runtime failure; text in vrt_last_error() (cf. HANDLE_ERROR, .cu:54-63) */
    VRT_ERR_OUT_OF_MEMORY = -4,
    VRT_ERR_NO_DEVICE = -5,
    VRT_ERR_SCENE = -6,             /* scene front end rejected the input (image_util.cpp:508-516, 537-543, 605) */
    VRT_ERR_RAY_RANGE = -7          /* "ray i: ... is not in 0 to ...", image_util.cpp:690-695 */
} vrt_status;

/* struct Options, types.h:83-91.  Zero-initialise, then call vrt_options_default(). */
typedef struct vrt_options {
    int32_t  loglevel;       /* _loglevel: <0 prints progress to stdout (cuda_volume_raytracer.cu:823, 948) */
    uint64_t minimum_gpu;    /* _minimum_gpu (0x80): the reference keeps calls with no more rays than this on the CPU
                                (cuda_volume_raytracer.cu:805); this engine has no CPU path and always runs on the GPU, but a
                                multi-device handle keeps such a call on its primary device instead of cutting it into bands */
    int32_t  write_instance; /* _write_instance: kept for ABI parity (debug record/replay is out of scope) */
    uint64_t max_cpu;        /* _max_cpu (256): kept for ABI parity; unused (no CPU path) */
    int32_t  device;         /* extension: HIP device ordinal, -1 = current device */
    int32_t  kernel;         /* extension: VRT_KERNEL_* selector, 0 = auto */
    int32_t  sort_rays;      /* extension: 0 = auto (3 for >= 4096 rays), 1 = tile ordering, 2 = tile ordering with per-ray length buckets,
                                3 = tile ordering + wavefronts launched longest-first, -1 = keep caller order */
    int32_t  slab_lds_bytes; /* extension: LDS bytes per wavefront for the slab kernel, 0 = default (10240, or 6144 for ray sets
                                denser than 8 rays per voxel column); rounded up to one of 6144 8192 10240 12288 16384 */
    int32_t  absorption;     /* extension, opt-in: 1 = run the absorption branch the reference compiles out
                              * (cuda_volume_raytracer.cu:337-341): per step brightness -= min(brightness,
                              * 0xFFFFFFFF - translucency[voxel]); stop when brightness < min_brightness;
                              * remaining_light returns the brightness.  Must be set at vrt_create (keeps the
                              * translucency resident) AND at vrt_trace.  PARITY UNPINNED: no reference build
                              * can execute this branch; the checker is oracle/vrt_oracle.c's restatement. */
    int32_t  schedule;       /* extension: slab-kernel scheduling. 0 = auto, 1 = static (one wavefront per group of 64
                                rays, hardware dispatch order), 2 = persistent wavefronts drawing groups from a counter,
                                3 = persistent and time-sliced (a group marches one slice -- 1.15 x the mean group length the
                                pre-march estimated, or 1024 wave-steps without an estimate -- parks its ray state and is resumed
                                by whichever wavefront draws the item from the FIFO; the long groups' remainders end the launch).
                                In auto mode the engine times its first four launches of a workload shape alternately with 2 and
                                3 and keeps the faster (vrt_engine_last_schedule tells which a launch used); a launch that runs
                                more than 15 % off the tuned time three times in a row starts the trials again. */
    int32_t  devices;        /* extension, read at vrt_create / vrt_scene_create: bit mask of HIP device ordinals the handle spans
                              * (bit d = device d), 0 = only the primary (default), -1 = every visible device.  The primary is
                              * `device`, or the caller's current device when device = -1; the mask adds replicas.  The field
                              * (and a scene's ior / density volume) is replicated on each of them, all copies in flight at once,
                              * as the reference's constructor does (cuda_volume_raytracer.cu:676-686), and EVERY trace call cuts
                              * the rays into one contiguous band per device, the way the reference spreads its chunks over all
                              * devices (.cu:804-912): host-pointer calls upload / march / download each band on its device;
                              * device-pointer calls leave the arrays on the primary device and the other devices read and write
                              * their bands through peer access (xGMI), forked from and joined into the caller's stream by events.
                              * Memory: one field copy per device of the mask -- under one process per GPU with all devices
                              * visible, -1 makes every rank replicate everywhere; give such ranks devices = 0. */
    int32_t  host_lanes;     /* extension: chunks a host-pointer call is cut into per device (pipelined over up to 4 streams so
                              * that uploads and downloads overlap the march), 0 = auto = 1 (measured best: every chunk pays
                              * its own ray ordering and launch tail) */
    int32_t  fast_math;      /* extension, opt-in, read at trace time: 1 = TOLERANCE MODE.  The default (0) reproduces the reference's
                              * CPU build bit for bit; with 1 the default slab kernel takes the step length from the hardware
                              * reciprocal without refinement and rounds the position update half-up in one conversion (8 of the
                              * step's 79 vector instructions; march -4 % on BASELINE config 3).  Results are then no longer the
                              * reference's bits: end_iteration equal on > 99.9 % of rays and +-1 otherwise, positions and directions
                              * within 1e-5 relative on smooth fields (tests/test_gpu_fastmath.py; config 3: 2e-6 at most) -- but a
                              * ray that grazes a strongly refracting structure amplifies any rounding difference (config 5: 99.9 %
                              * of the rays within 1e-5, the worst 6e-5 / 2e-4), as the reference's own CUDA and CPU builds differ from
                              * each other.  Other kernels (direct, 6 KiB, absorption) ignore it. */
} vrt_options;

#define VRT_KERNEL_AUTO   0
#define VRT_KERNEL_DIRECT 1   /* one lane per ray, trilinear corners gathered straight from L1/L2 */
#define VRT_KERNEL_SLAB   2   /* wave-cooperative LDS slab staging */

typedef struct vrt_engine *vrt_handle;

void vrt_options_default(vrt_options *opt);

/* Number of usable HIP devices; replaces init()'s cudaGetDeviceCount (cuda_volume_raytracer.cu:82-101). */
int vrt_device_count(void);

/* ABI version of the loaded library (== VRT_ABI_VERSION). */
int vrt_abi_version(void);

/*
 * Replaces TraceRaysCu<DiffType>::TraceRaysCu(bounds, diff, translucency_cropped)
 * (cuda_volume_raytracer.cu:644-720): builds w = (0x7FFFFFFF - translucency)/0x10000,
 * interleaves {diff[0..dim), w} per voxel and keeps the result resident in HBM.
 *   dim            must be 3 (2 -> VRT_ERR_ILLEGAL_DIMENSION: the reference's 2-D
 *                  interpolation is defective, .cu:207-208, and no config uses it)
 *   bounds[dim]    field extents (= volume extents - 2)
 *   field_type     VRT_F32 (diff is float*) or VRT_I16 (diff is int16_t*)
 *   diff[dim]      HOST pointers, nvox elements each
 *   transl_cropped HOST pointer, nvox uint32
 */
int vrt_create(vrt_handle *out, int dim, const uint16_t *bounds, int field_type,
               const void *const *diff, const uint32_t *transl_cropped, size_t nvox,
               const vrt_options *opt);

/*
 * Same engine from DEVICE-resident inputs (no reference counterpart: the reference
 * re-uploads from host vectors, .cu:684-685).  diff[i] / transl_cropped are device
 * pointers on the engine's device; the interleave runs on the GPU.
 */
int vrt_create_from_device(vrt_handle *out, int dim, const uint16_t *bounds, int field_type,
                           const void *const *d_diff, const uint32_t *d_transl_cropped, size_t nvox,
                           const vrt_options *opt);

/*
 * Replaces TraceRaysCu<DiffType>::trace_rays_cu<DirType>(...) (cuda_volume_raytracer.cu:722-972).
 * All pointers are HOST pointers; the call returns when the outputs are complete.
 *   dir_type        VRT_F32 / VRT_I16 (type of start_dir and end_dir)
 *   start_pos       3*nrays uint32        start_dir  3*nrays
 *   end_pos         3*nrays uint32        end_dir    3*nrays
 *   end_iter        nrays  (= iterations - remaining, .cu:953-956)
 *   remaining_light nrays  (always 0xFFFFFFFF: translucency is a DummyArray at every
 *                   reference call site, .cu:882-938, 485)
 *   path            NULL, or iterations*3*nrays uint32 laid out as the reference's
 *                   path vector (.cu:791-794, 333-358)
 *   invscale        3 floats
 *   min_brightness  ignored, as in the reference (.cu:785), unless the opt-in absorption mode is on
 *                   (vrt_options.absorption), where it is the brightness cut-off of .cu:337-341
 *   iterations      per-ray step cap, must be >= 1
 * A handle created over several devices (vrt_options.devices) marches one contiguous band of the rays on each.
 * One call at a time per handle: concurrent calls from several host threads are serialised.
 */
int vrt_trace(vrt_handle h, int dir_type, size_t nrays,
              const uint32_t *start_pos, const void *start_dir,
              uint32_t *end_pos, void *end_dir, uint32_t *end_iter, uint32_t *remaining_light,
              uint32_t *path, const float *invscale, uint32_t min_brightness,
              uint32_t iterations, const vrt_options *opt);

/*
 * Device-pointer form of vrt_trace: every array argument is a DEVICE pointer on the
 * engine's (primary) device (invscale stays a host pointer).  Work is enqueued on `stream`
 * (a hipStream_t, NULL = the null stream) and the call returns without synchronising.
 * The ray-order and scheduling workspaces belong to the handle, not to the stream: calls issued on
 * different streams of one handle are ordered one after the other (each waits, on the device, for the
 * previous call's last kernel), so they are safe but do not overlap; use one handle per stream to overlap.
 */
int vrt_trace_device(vrt_handle h, int dir_type, size_t nrays,
                     const uint32_t *d_start_pos, const void *d_start_dir,
                     uint32_t *d_end_pos, void *d_end_dir, uint32_t *d_end_iter, uint32_t *d_remaining_light,
                     uint32_t *d_path, const float *invscale, uint32_t min_brightness,
                     uint32_t iterations, void *stream, const vrt_options *opt);

/*
 * Completion status of the device-pointer calls issued on a handle so far (all of its devices): waits for the most recent
 * one and returns VRT_OK, or VRT_ERR_HIP if the time-sliced schedule had to give up a wait since the last status call
 * ("a wavefront gave up waiting": some rays were left unfinished -- the waits are bounded so that a launch can never hang,
 * and a bound that is hit must not pass silently).  The condition is sticky until it has been reported once.  The
 * host-pointer calls (vrt_trace, vrt_scene_trace) check it themselves and leave the output arrays untouched on failure.
 * The reference reports device failures by throwing from trace_rays_cu (HANDLE_ERROR, cuda_volume_raytracer.cu:54-63);
 * an asynchronous call cannot, hence this function.
 */
int vrt_trace_status(vrt_handle h);

/* Replaces TraceRaysCu<DiffType>::~TraceRaysCu() (cuda_volume_raytracer.cu:974-989). NULL is a no-op. */
void vrt_destroy(vrt_handle h);

/* Introspection (extension). */
int    vrt_engine_device(vrt_handle h);             /* device ordinal the field lives on */
size_t vrt_engine_field_bytes(vrt_handle h);        /* resident interleaved field size */
const void *vrt_engine_field_device_ptr(vrt_handle h);
/* Duration in milliseconds of the most recent march kernel launched through this engine,
 * from HIP events recorded on the launch stream around the kernel (waits for it to finish).
 * Negative if nothing has been launched.  Used by bench.py for the roofline figure. */
float vrt_engine_last_march_ms(vrt_handle h);
/* Shader clock, in GHz, that the most recent march kernel really ran at: one of its wavefronts reads s_memtime and
 * s_memrealtime when it starts and when it ends.  Negative if unavailable.  The roofline's VALU-issue peak is
 * SIMDs x this clock / 2 (bench.py). */
float vrt_engine_last_march_clock_ghz(vrt_handle h);
/* Number of devices (engines) behind a handle: 1 + replicas made for vrt_options.devices. */
int vrt_engine_replica_count(vrt_handle h);
/* Schedule the most recent march kernel of the handle's primary engine was launched with: 1 static, 2 persistent wavefronts
 * drawing whole groups, 3 persistent and time-sliced (see vrt_options.schedule); 0 = the direct kernel, or nothing launched
 * yet.  bench.py uses it to pick the counter profile of the kernel that actually ran. */
int vrt_engine_last_schedule(vrt_handle h);
/* Copies the first `bytes` bytes of the resident interleaved field ({dx,dy,dz,w} per voxel, index ((x*By)+y)*Bz+z: the
 * array the reference's constructor builds on the host, cuda_volume_raytracer.cu:654-669) to host memory.  Tests compare
 * the scene build's output with the reference's through it. */
int vrt_engine_download_field(vrt_handle h, void *host_dst, size_t bytes);
/* What the engine learnt from its one scan of the field: bounds4 = max |dx|, max |dy|, max |dz| and an upper bound of the
 * longest gradient vector (they bound the change of a direction per step for the proven rounds); *flags bit 0 = every
 * value is zero or within [2^-55, 2^78), i.e. bricks may be stored prescaled without a per-brick test. */
int vrt_engine_field_scan(vrt_handle h, float *bounds4, uint32_t *flags);
/* The contiguous cut of n rays into `parts` bands that multi-device handles and the chunk pipeline use
 * (begin[0..parts], whole groups of 64 rays where possible).  Exposed for tests. */
int vrt_debug_band_cut(size_t n, size_t parts, size_t *begin);
/* Development aid: with VRT_DEBUG_STATS=1 in the environment the slab kernel counts planning
 * rounds / staged voxels / LDS steps; this copies up to n counters of the last launch (64 counters, then,
 * in -DVRT_SLAB_TIMELINE development builds, a begin/end clock pair per wavefront). */
int vrt_engine_debug_stats(vrt_handle h, unsigned long long *out, int n);
/* Exhaustive device self-test of the march step's short replacement for the reference's IEEE divide
 * 0x42000000p0f / dot (cuda_volume_raytracer.cu:346): for every float bit pattern the short path admits,
 * compares it with the IEEE quotient.  *mismatches must come back 0; *admitted = number of patterns
 * taking the short path (219 exponents x 2^23 mantissas). */
int vrt_selftest_ilen2(int device, unsigned long long *mismatches, unsigned long long *admitted);
/* Exhaustive device self-test of the position update pos += roundf(s) (cuda_volume_raytracer.cu:347, std::round on the
 * step) as the march kernels compute it inside the step window (v_cvt_rpi_i32_f32 on |s|, sign folded into the add):
 * every float with |s| < 2^31 against std::round + the x86 conversion.  *mismatches must come back 0;
 * *checked = 2 x 158 x 2^23 patterns. */
int vrt_selftest_round(int device, unsigned long long *mismatches, unsigned long long *checked);
/* Exhaustive device self-test of the integer division by multiplication with which the slab kernel splits a brick's voxel
 * number into coordinates while staging: every divisor 2..1024 with every dividend below 1024, quotient and remainder.
 * *mismatches must come back 0; *checked = 1023 x 1024. */
int vrt_selftest_stage_division(int device, unsigned long long *mismatches, unsigned long long *checked);

/* ---------------------------------------------------------------------------------------------
 * Scene layer: the reference's RaytraceScene<Ior,IorLog,Diff>
 * (image_util.h:132-195), i.e. what python_binding.cpp:38 and java_binding.cpp:179-287 call.
 * Instantiations as in the reference (image_util.cpp:940-956):
 *   ior_type VRT_F32: float ior, fp32 field, float directions only
 *   ior_type VRT_U32: ior_t 16.16 fixed, int16 field, int16 (8.8) directions only
 * --------------------------------------------------------------------------------------------- */
#define VRT_U32 2

typedef struct vrt_scene *vrt_scene_handle;

/*
 * Replaces RaytraceScene::RaytraceScene(bound_vec, ior, translucency, opt) (image_util.cpp:501-643):
 * size checks, translucency crop, ior_log = log(ior)*0x420000 (host libm, as the reference),
 * 3x3x3 derivative stencil + interleave on the device, engine creation.
 *   bounds[dim]  volume extents (the field is bounds-2); dim must be 3
 *   ior, translucency  HOST pointers, nvox elements
 * Errors: VRT_ERR_SCENE with the reference's message ("imagesizes doesn't match...",
 * "refraction-index underflow...", "differention overflow...").
 */
int vrt_scene_create(vrt_scene_handle *out, int dim, const size_t *bounds, int ior_type, const void *ior,
                     const uint32_t *translucency, size_t nvox, const vrt_options *opt);

/*
 * BASELINE config 5 (SURVEY §0 reading A, row f-3): an 8-bit density volume with a 256-entry transfer
 * function d -> (ior, translucency).  The reference has no 8-bit path; this call is defined as
 * vrt_scene_create on the expanded arrays ior[i] = lut_ior[density[i]], translucency[i] =
 * lut_translucency[density[i]] (so the reference on the expanded arrays is the oracle), but only the
 * 1-byte volume is uploaded and kept: the logarithm is taken of the 256 table entries and the stencil
 * and the ray front end read the volume through the table.
 *   density          HOST pointer, nvox bytes
 *   lut_ior          256 entries, float (VRT_F32) or ior_t 16.16 (VRT_U32)
 *   lut_translucency 256 entries
 */
int vrt_scene_create_lut(vrt_scene_handle *out, int dim, const size_t *bounds, int ior_type, const uint8_t *density,
                         const void *lut_ior, const uint32_t *lut_translucency, size_t nvox, const vrt_options *opt);

/*
 * Replaces RaytraceScene::trace_rays<DirType>(...) (image_util.cpp:645-772): range check
 * (VRT_ERR_RAY_RANGE, "ray i:... is not in 0 to ..."), -0x8000 shift, ior interpolation, direction
 * normalisation (always on, as the reference's "normalize_length || true"), second shift, the march,
 * and the +0x10000 epilogue on end_pos and path.  Positions are in VOLUME coordinates.
 * HOST pointers; dir_type must match the scene's family.
 */
int vrt_scene_trace(vrt_scene_handle s, int dir_type, size_t nrays, const uint32_t *start_pos, const void *start_dir,
                    uint32_t *end_pos, void *end_dir, uint32_t *end_iter, uint32_t *remaining_light,
                    uint32_t *path, const float *invscale, uint32_t min_brightness, uint32_t iterations,
                    int normalize_length, const vrt_options *opt);

/*
 * Device-pointer form of vrt_scene_trace ("lets Python pass device buffers"): all ray arrays are DEVICE
 * pointers on the scene's device, the ray front end, the march and the +0x10000 epilogue are enqueued on `stream`
 * and the call returns without synchronising.  The start arrays are not modified.  Range / normalisation errors of
 * the rays cannot be known when the call returns: vrt_scene_trace_status() waits for the call and reports them.
 */
int vrt_scene_trace_device(vrt_scene_handle s, int dir_type, size_t nrays, const uint32_t *d_start_pos,
                           const void *d_start_dir, uint32_t *d_end_pos, void *d_end_dir, uint32_t *d_end_iter,
                           uint32_t *d_remaining_light, uint32_t *d_path, const float *invscale,
                           uint32_t min_brightness, uint32_t iterations, void *stream, const vrt_options *opt);
/* Waits for the most recent vrt_scene_trace_device call of the scene and returns VRT_OK, VRT_ERR_RAY_RANGE
 * ("ray i is not in 0 to ...", image_util.cpp:690-695), VRT_ERR_SCENE ("Normalize length failed", :705-708) or what
 * vrt_trace_status reports for the march.  Ray errors accumulate over the device-pointer calls since the previous status
 * call (the lowest failing ray index of any of them is reported) and are cleared by it. */
int vrt_scene_trace_status(vrt_scene_handle s);

/* Replaces RaytraceScene::~RaytraceScene (image_util.cpp:774-778). */
void vrt_scene_destroy(vrt_scene_handle s);

/* The seam-level engine inside a scene (owned by the scene). */
vrt_handle vrt_scene_engine(vrt_scene_handle s);

/* Message of the last failure on this thread ("" if none).  Valid until the next failing call. */
const char *vrt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* VRT_H */
