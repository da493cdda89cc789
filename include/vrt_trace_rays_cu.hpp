// vrt_trace_rays_cu.hpp -- C++ host-side mirror of the reference seam over the C ABI (vrt.h).
//
// Same names, argument meaning and error behaviour as the reference's engine class
//   template <typename DiffType> class TraceRaysCu      /root/reference/src/cuda_volume_raytracer.h:61-115
// so that code (and tests) written against the reference read the same here:
//   - ctor(bounds, diff, translucency_cropped)           .cu:644-720
//   - trace_rays_cu<DirType>(start_position, start_direction, end_position, end_direction,
//       end_iterations, remaining_light, path, scale_vec, minimum_brightness, iterations,
//       trace_paths, opt)                                .cu:722-972
//   - failures surface as std::runtime_error             .cu:54-63, 770
// Header-only; link against libvrt_hip.so.  All compute happens on the MI355X behind the C ABI.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <cstdlib>
#include <vector>

#include "vrt.h"

namespace vrt {

// struct Options, /root/reference/src/types.h:83-91
struct Options {
    int _loglevel = 0;
    size_t _minimum_gpu = 0x80;
    bool _write_instance = false;
    size_t _max_cpu = 256;
    Options() = default;
    Options(size_t loglevel_, size_t minimum_gpu_, bool write_instance_)
        : _loglevel((int)loglevel_), _minimum_gpu(minimum_gpu_), _write_instance(write_instance_) {}
    vrt_options c() const
    {
        vrt_options o;
        vrt_options_default(&o);
        o.loglevel = _loglevel;
        o.minimum_gpu = _minimum_gpu;
        o.write_instance = _write_instance ? 1 : 0;
        o.max_cpu = _max_cpu;
        return o;
    }
};

template <typename T> struct elem_type;
template <> struct elem_type<float>   { static constexpr int value = VRT_F32; };
template <> struct elem_type<int16_t> { static constexpr int value = VRT_I16; };

inline void check(int rc)
{
    if (rc != VRT_OK) throw std::runtime_error(vrt_last_error());
}

template <typename DiffType>
class TraceRaysCu {
    vrt_handle _h = nullptr;

public:
    std::vector<uint16_t> _output_sizes;   // public in the reference too (cuda_volume_raytracer.h:75)

    TraceRaysCu(std::vector<size_t> const &output_sizes_, std::vector<std::vector<DiffType> const *> const &diff_,
                std::vector<uint32_t> const &translucency_cropped_)
    {
        for (size_t b : output_sizes_) _output_sizes.push_back((uint16_t)b);      // make_struct<uint16_t>, .cu:813
        std::vector<const void *> d;
        for (auto const *v : diff_) d.push_back(v->data());
        // every visible device, as the reference's constructor replicates onto every CUDA device (.cu:676-686);
        // trace_rays_cu then marches one band of the rays per device (.cu:804-912)
        vrt_options o;
        vrt_options_default(&o);
        o.devices = -1;
        if (const char *m = std::getenv("VRT_DEVICES")) o.devices = (int32_t)std::strtol(m, nullptr, 0);   // one process per GPU: VRT_DEVICES=0
        check(vrt_create(&_h, (int)output_sizes_.size(), _output_sizes.data(), elem_type<DiffType>::value, d.data(),
                         translucency_cropped_.data(), translucency_cropped_.size(), &o));
    }

    TraceRaysCu(std::vector<size_t> const &output_sizes_, std::vector<std::vector<DiffType>> const &diff_,
                std::vector<uint32_t> const &translucency_cropped_)
        : TraceRaysCu(output_sizes_, refs(diff_), translucency_cropped_) {}

    TraceRaysCu(TraceRaysCu const &) = delete;
    TraceRaysCu &operator=(TraceRaysCu const &) = delete;

    template <typename DirType>
    void trace_rays_cu(std::vector<uint32_t> const &start_position, std::vector<DirType> const &start_direction,
                       std::vector<uint32_t> &end_position, std::vector<DirType> &end_direction,
                       std::vector<uint32_t> &end_iterations, std::vector<uint32_t> &remaining_light,
                       std::vector<uint32_t> &path, std::vector<float> const &scale_vec, uint32_t minimum_brightness,
                       uint32_t iterations, bool trace_paths, Options const &opt)
    {
        const size_t dim = _output_sizes.size();
        if (dim != 3) throw std::runtime_error("Illegal dimension");                 // .cu:770
        const size_t num_rays = start_position.size() / dim;
        if (trace_paths) path.resize((size_t)iterations * dim * num_rays);           // .cu:791-794
        // outputs are pre-sized by the caller in the reference (image_util.cpp:738-741); be lenient
        if (end_position.size() < num_rays * dim) end_position.resize(num_rays * dim);
        if (end_direction.size() < num_rays * dim) end_direction.resize(num_rays * dim);
        if (end_iterations.size() < num_rays) end_iterations.resize(num_rays);
        if (remaining_light.size() < num_rays) remaining_light.resize(num_rays);
        vrt_options o = opt.c();
        check(vrt_trace(_h, elem_type<DirType>::value, num_rays, start_position.data(), start_direction.data(),
                        end_position.data(), end_direction.data(), end_iterations.data(), remaining_light.data(),
                        trace_paths ? path.data() : nullptr, scale_vec.data(), minimum_brightness, iterations, &o));
    }

    ~TraceRaysCu() { vrt_destroy(_h); }

private:
    static std::vector<std::vector<DiffType> const *> refs(std::vector<std::vector<DiffType>> const &data)
    {
        std::vector<std::vector<DiffType> const *> r;
        for (auto const &d : data) r.push_back(&d);
        return r;
    }
};

}  // namespace vrt
