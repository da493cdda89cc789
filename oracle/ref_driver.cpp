/*
 * ref_driver.cpp -- C-ABI handle on the REAL reference CPU build (test infrastructure).
 *
 * This file is ours; it only #includes the reference's headers where they lie
 * (-I/root/reference/src) and is linked with objects compiled from the reference's
 * own sources by oracle/Makefile into oracle/_ref/libvrt_ref.so (git-ignored).
 * No reference source is copied into this repository.
 *
 * Exports two levels:
 *   ref_seam_*   : TraceRaysCu<Diff>(bounds, diff, transl_cropped).trace_rays_cu<Dir>(...)
 *                  (cuda_volume_raytracer.h:61-115) -- the drop-in seam.
 *   ref_scene_*  : RaytraceScene<Ior,IorLog,Diff>(...).trace_rays<Dir>(...)
 *                  (image_util.h:132-195) -- what python_binding.cpp:38 / java_binding.cpp:127 call.
 *   ref_build_*  : scene build only, returning the derivative arrays (private
 *                  members reached with the usual test-only access hack).
 */
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <stdexcept>
#include <memory>
#include <sstream>
#include <iostream>
#include <fstream>
#include <numeric>
#include <algorithm>
#include <array>
#include <map>
#include <set>
#include <functional>
#include <iterator>
#include <cmath>
#include <limits>
#include <cinttypes>
#include <iomanip>
#include <chrono>
#include <thread>
#include <mutex>
#include <type_traits>
#include <experimental/filesystem>

/* test-only access to RaytraceScene's private derivative arrays; every std header the
 * reference headers pull in is included above so the macro touches reference code only */
#define private public
#include "image_util.h"
#undef private
#include "cuda_volume_raytracer.h"
#include "types.h"

static thread_local std::string g_err;

template <typename Diff, typename Dir>
static int seam_trace(const size_t *bounds, const Diff *dx, const Diff *dy, const Diff *dz,
                      const uint32_t *transl_cropped, size_t nvox, size_t nrays,
                      const uint32_t *start_pos, const Dir *start_dir,
                      uint32_t *end_pos, Dir *end_dir, uint32_t *end_iter, uint32_t *remaining,
                      uint32_t *path, const float *invscale, uint32_t iterations, int max_cpu)
{
    try {
        std::vector<size_t> b(bounds, bounds + 3);
        std::vector<std::vector<Diff>> diff(3);
        diff[0].assign(dx, dx + nvox); diff[1].assign(dy, dy + nvox); diff[2].assign(dz, dz + nvox);
        std::vector<uint32_t> tr(transl_cropped, transl_cropped + nvox);
        TraceRaysCu<Diff> eng(b, diff, tr);
        std::vector<pos_t> sp(start_pos, start_pos + 3 * nrays), ep(3 * nrays), pth;
        std::vector<Dir> sd(start_dir, start_dir + 3 * nrays), ed(3 * nrays);
        std::vector<uint32_t> ei(nrays), rl(nrays);
        std::vector<float> inv(invscale, invscale + 3);
        Options opt;
        opt._loglevel = 0;
        if (max_cpu > 0) opt._max_cpu = max_cpu;
        eng.template trace_rays_cu<Dir>(sp, sd, ep, ed, ei, rl, pth, inv, 0, iterations, path != nullptr, opt);
        std::memcpy(end_pos, ep.data(), ep.size() * 4);
        std::memcpy(end_dir, ed.data(), ed.size() * sizeof(Dir));
        std::memcpy(end_iter, ei.data(), ei.size() * 4);
        std::memcpy(remaining, rl.data(), rl.size() * 4);
        if (path) std::memcpy(path, pth.data(), pth.size() * 4);
        return 0;
    } catch (std::exception const &e) { g_err = e.what(); return -1; }
}

template <typename Ior, typename IorLog, typename Diff, typename Dir>
static int scene_trace(const size_t *bounds, const Ior *ior, const uint32_t *transl, size_t nvox, size_t nrays,
                       const uint32_t *start_pos, const Dir *start_dir,
                       uint32_t *end_pos, Dir *end_dir, uint32_t *end_iter, uint32_t *remaining,
                       uint32_t *path, const float *invscale, uint32_t iterations, int max_cpu)
{
    try {
        std::vector<size_t> b(bounds, bounds + 3);
        std::vector<Ior> iv(ior, ior + nvox);
        std::vector<uint32_t> tr(transl, transl + nvox);
        Options opt;
        opt._loglevel = 0;
        if (max_cpu > 0) opt._max_cpu = max_cpu;
        RaytraceScene<Ior, IorLog, Diff> scene(b, iv, tr, opt);
        std::vector<pos_t> sp(start_pos, start_pos + 3 * nrays), ep, pth;
        std::vector<Dir> sd(start_dir, start_dir + 3 * nrays), ed;
        std::vector<uint32_t> ei, rl;
        std::vector<float> inv(invscale, invscale + 3);
        RayTraceRayInstanceRef<Dir> ref(sp, sd, inv, 0, iterations, path != nullptr, true);
        scene.template trace_rays<Dir>(ref, ep, ed, ei, rl, pth, opt);
        std::memcpy(end_pos, ep.data(), ep.size() * 4);
        std::memcpy(end_dir, ed.data(), ed.size() * sizeof(Dir));
        std::memcpy(end_iter, ei.data(), ei.size() * 4);
        std::memcpy(remaining, rl.data(), rl.size() * 4);
        if (path) std::memcpy(path, pth.data(), pth.size() * 4);
        return 0;
    } catch (std::exception const &e) { g_err = e.what(); return -1; }
}

template <typename Ior, typename IorLog, typename Diff>
static int scene_build(const size_t *bounds, const Ior *ior, const uint32_t *transl, size_t nvox,
                       Diff *dx, Diff *dy, Diff *dz, uint32_t *transl_cropped, IorLog *ior_log)
{
    try {
        std::vector<size_t> b(bounds, bounds + 3);
        std::vector<Ior> iv(ior, ior + nvox);
        std::vector<uint32_t> tr(transl, transl + nvox);
        Options opt;
        RaytraceScene<Ior, IorLog, Diff> scene(b, iv, tr, opt);
        size_t n = scene._diff[0].size();
        std::memcpy(dx, scene._diff[0].data(), n * sizeof(Diff));
        std::memcpy(dy, scene._diff[1].data(), n * sizeof(Diff));
        std::memcpy(dz, scene._diff[2].data(), n * sizeof(Diff));
        std::memcpy(transl_cropped, scene._translucency_cropped.data(), n * 4);
        if (ior_log) std::memcpy(ior_log, scene._ior_log.data(), nvox * sizeof(IorLog));
        return 0;
    } catch (std::exception const &e) { g_err = e.what(); return -1; }
}

extern "C" {

const char *ref_last_error(void) { return g_err.c_str(); }

#define SEAM(NAME, DIFF, DIR)                                                                              \
    int NAME(const size_t *bounds, const DIFF *dx, const DIFF *dy, const DIFF *dz, const uint32_t *tc,      \
             size_t nvox, size_t nrays, const uint32_t *sp, const DIR *sd, uint32_t *ep, DIR *ed,           \
             uint32_t *ei, uint32_t *rl, uint32_t *path, const float *inv, uint32_t iters, int max_cpu)     \
    { return seam_trace<DIFF, DIR>(bounds, dx, dy, dz, tc, nvox, nrays, sp, sd, ep, ed, ei, rl, path, inv, iters, max_cpu); }

SEAM(ref_seam_f32_f32, float, float)
SEAM(ref_seam_f32_i16, float, int16_t)
SEAM(ref_seam_i16_f32, int16_t, float)
SEAM(ref_seam_i16_i16, int16_t, int16_t)

int ref_scene_f32(const size_t *bounds, const float *ior, const uint32_t *tr, size_t nvox, size_t nrays,
                  const uint32_t *sp, const float *sd, uint32_t *ep, float *ed, uint32_t *ei, uint32_t *rl,
                  uint32_t *path, const float *inv, uint32_t iters, int max_cpu)
{ return scene_trace<float, float, float, float>(bounds, ior, tr, nvox, nrays, sp, sd, ep, ed, ei, rl, path, inv, iters, max_cpu); }

int ref_scene_i16(const size_t *bounds, const uint32_t *ior, const uint32_t *tr, size_t nvox, size_t nrays,
                  const uint32_t *sp, const int16_t *sd, uint32_t *ep, int16_t *ed, uint32_t *ei, uint32_t *rl,
                  uint32_t *path, const float *inv, uint32_t iters, int max_cpu)
{ return scene_trace<ior_t, iorlog_t, diff_t, dir_t>(bounds, ior, tr, nvox, nrays, sp, sd, ep, ed, ei, rl, path, inv, iters, max_cpu); }

int ref_build_f32(const size_t *bounds, const float *ior, const uint32_t *tr, size_t nvox,
                  float *dx, float *dy, float *dz, uint32_t *tc, float *ior_log)
{ return scene_build<float, float, float>(bounds, ior, tr, nvox, dx, dy, dz, tc, ior_log); }

int ref_build_i16(const size_t *bounds, const uint32_t *ior, const uint32_t *tr, size_t nvox,
                  int16_t *dx, int16_t *dy, int16_t *dz, uint32_t *tc, int32_t *ior_log)
{ return scene_build<ior_t, iorlog_t, diff_t>(bounds, ior, tr, nvox, dx, dy, dz, tc, ior_log); }

/* write debug_scene_instance / debug_ray_instance with the reference's own serializer
 * (image_util.cpp:75-92, image_util.h:236-246) -- golden files for the wire-format tests */
int ref_write_instances_i16(const char *scene_path, const char *ray_path, const size_t *bounds, const uint32_t *ior,
                            const uint32_t *tr, size_t nvox, size_t nrays, const uint32_t *sp, const int16_t *sd,
                            const float *inv, uint32_t min_brightness, uint32_t iters, int trace_path, int normalize)
{
    try {
        RayTraceSceneInstance<ior_t> sc;
        sc._bound_vec.assign(bounds, bounds + 3);
        sc._ior.assign(ior, ior + nvox);
        sc._translucency.assign(tr, tr + nvox);
        std::ofstream f1(scene_path, std::ios::binary);
        SERIALIZE::write_value(f1, sc);
        f1.close();
        std::vector<pos_t> vsp(sp, sp + 3 * nrays);
        std::vector<dir_t> vsd(sd, sd + 3 * nrays);
        std::vector<float> vinv(inv, inv + 3);
        RayTraceRayInstanceRef<dir_t> ray(vsp, vsd, vinv, min_brightness, iters, trace_path != 0, normalize != 0);
        std::ofstream f2(ray_path, std::ios::binary);
        SERIALIZE::write_value(f2, ray);
        f2.close();
        return 0;
    } catch (std::exception const &e) { g_err = e.what(); return -1; }
}

}  // extern "C"
