// reference_seam_shim.cpp -- what a maintainer of the reference adds to run trace_rays on MI355X.
//
// The reference's scene layer (image_util.o) imports exactly six symbols from its hot-path object
// (cuda_volume_raytracer.o): TraceRaysCu<short|float>::{ctor, dtor, trace_rays_cu<short|float>}
// (cuda_volume_raytracer.h:61-115; call sites image_util.cpp:561, 630, 752, 777).  This translation
// unit defines those symbols on top of the C ABI in include/vrt.h, so that replacing
// cuda_volume_raytracer.o by this object + libvrt_hip.so in the reference's link line
// (Makefile:75-105) moves the march to the GPU with no other change: python_binding.cpp,
// java_binding.cpp and image_util.cpp are compiled from the reference tree untouched.
//
// Compiled against the reference's own header where it lies (-I/root/reference/src); nothing from
// the reference is copied here.  The engine handle is kept in the class's spare `void *_tex` member
// (cuda_volume_raytracer.h:70), which the CPU build never touches.
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <vector>

#include "cuda_volume_raytracer.h"   // the reference's class declaration
#include "vrt.h"

namespace {
template <typename T> struct elem;
template <> struct elem<float> { static constexpr int value = VRT_F32; };
template <> struct elem<short> { static constexpr int value = VRT_I16; };
void check(int rc) { if (rc != VRT_OK) throw std::runtime_error(vrt_last_error()); }
}  // namespace

template <typename DiffType>
TraceRaysCu<DiffType>::TraceRaysCu(std::vector<size_t> const &bounds, std::vector<std::vector<DiffType> const *> const &diff,
                                   std::vector<translucency_t> const &translucency_cropped)
    : _translucency_cropped(translucency_cropped), _cudaTexture(nullptr), _tex(nullptr)
{
    for (size_t b : bounds) _output_sizes.push_back(static_cast<uint16_t>(b));
    std::vector<const void *> d;
    for (auto const *v : diff) d.push_back(v->data());
    // the reference replicates the field on every CUDA device (cuda_volume_raytracer.cu:676-686) and trace_rays_cu spreads
    // its chunks over all of them (:804-912): the handle spans every visible device and vrt_trace cuts the rays into
    // one band per device (calls with no more than Options::_minimum_gpu rays stay on one, as :805 keeps them on the CPU)
    // The primary engine lives on the caller's current device; the mask adds replicas.  A host application that runs one
    // process per GPU with every device visible would replicate everywhere from every process: VRT_DEVICES (a bit mask,
    // 0 = the current device only) narrows it without touching the reference's code.
    vrt_options o;
    vrt_options_default(&o);
    o.devices = -1;
    if (const char *m = std::getenv("VRT_DEVICES")) o.devices = static_cast<int32_t>(std::strtol(m, nullptr, 0));
    vrt_handle h = nullptr;
    check(vrt_create(&h, static_cast<int>(bounds.size()), _output_sizes.data(), elem<DiffType>::value, d.data(),
                     translucency_cropped.data(), translucency_cropped.size(), &o));
    _tex = h;
}

template <typename DiffType>
TraceRaysCu<DiffType>::TraceRaysCu(std::vector<size_t> const &bounds, std::vector<std::vector<DiffType>> const &diff,
                                   std::vector<translucency_t> const &translucency_cropped)
    : TraceRaysCu(bounds, [&] { std::vector<std::vector<DiffType> const *> r; for (auto const &v : diff) r.push_back(&v); return r; }(),
                  translucency_cropped) {}

template <typename DiffType>
template <typename DirType>
void TraceRaysCu<DiffType>::trace_rays_cu(std::vector<pos_t> const &start_position, std::vector<DirType> const &start_direction,
                                          std::vector<pos_t> &end_position, std::vector<DirType> &end_direction,
                                          std::vector<uint32_t> &end_iterations, std::vector<brightness_t> &remaining_light,
                                          std::vector<pos_t> &path, std::vector<float> const &scale_vec,
                                          brightness_t minimum_brightness, uint32_t iterations, bool trace_paths,
                                          Options const &opt)
{
    const size_t dim = _output_sizes.size();
    if (dim != 3) throw std::runtime_error("Illegal dimension");
    const size_t num_rays = start_position.size() / dim;
    if (trace_paths) path.resize(static_cast<size_t>(iterations) * dim * num_rays);
    vrt_options o;
    vrt_options_default(&o);
    o.loglevel = opt._loglevel;
    o.minimum_gpu = opt._minimum_gpu;
    o.write_instance = opt._write_instance;
    o.max_cpu = opt._max_cpu;
    check(vrt_trace(static_cast<vrt_handle>(_tex), elem<DirType>::value, num_rays, start_position.data(),
                    start_direction.data(), end_position.data(), end_direction.data(), end_iterations.data(),
                    remaining_light.data(), trace_paths ? path.data() : nullptr, scale_vec.data(), minimum_brightness,
                    iterations, &o));
}

template <typename DiffType>
TraceRaysCu<DiffType>::~TraceRaysCu() { vrt_destroy(static_cast<vrt_handle>(_tex)); }

// the same explicit instantiations as cuda_volume_raytracer.cu:992-1049
template class TraceRaysCu<diff_t>;
template class TraceRaysCu<float>;
#define VRT_INST(DIFF, DIR)                                                                                          \
    template void TraceRaysCu<DIFF>::trace_rays_cu<DIR>(std::vector<pos_t> const &, std::vector<DIR> const &,         \
        std::vector<pos_t> &, std::vector<DIR> &, std::vector<uint32_t> &, std::vector<brightness_t> &,               \
        std::vector<pos_t> &, std::vector<float> const &, brightness_t, uint32_t, bool, Options const &);
VRT_INST(diff_t, dir_t) VRT_INST(float, dir_t) VRT_INST(diff_t, float) VRT_INST(float, float)
