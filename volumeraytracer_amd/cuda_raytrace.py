"""Drop-in for the reference's pybind11 module `cuda_raytrace`
(/root/reference/src/python_binding.cpp:10-52), running on the MI355X engine.

    from volumeraytracer_amd import cuda_raytrace
    end_position, end_direction, end_iteration, remaining_light = cuda_raytrace.cuda_raytrace(
        bound_vec, ior, translucency, start_position, start_direction, scale,
        minimum_brightness, iterations, trace_paths)

Same nine positional arguments, same 4-tuple of lists, same slot order (slot 2 receives
end_iteration and slot 3 remaining_light, python_binding.cpp:38-45; the path is computed when
trace_paths is set and discarded, as there), same instantiation (ior_t 16.16 fixed-point ior,
dir_t 8.8 directions, python_binding.cpp:38).  The reference also writes a junk debug file
`debug_raytrace_instance` into the cwd on every call (python_binding.cpp:21-34); that side effect
is not reproduced.
"""
import numpy as np

from . import capi

__doc__ += "\nA function which casts rays with cuda"   # m.doc(), python_binding.cpp:50


def cuda_raytrace(bound_vec, ior, transculency, start_position, start_direction, scale, minimum_brightness,
                  iterations, trace_paths):
    """A function which casts rays with cuda"""
    ior = np.asarray(ior, dtype=np.uint32)
    tr = np.asarray(transculency, dtype=np.uint32)
    sp = np.asarray(start_position, dtype=np.uint32)
    sd = np.asarray(start_direction, dtype=np.int16)
    with capi.Scene([int(b) for b in bound_vec], ior, tr) as scene:
        out = scene.trace_rays(sp, sd, np.asarray(scale, dtype=np.float32), int(iterations), bool(trace_paths),
                               int(minimum_brightness), True)
    return (out["end_position"].tolist(), out["end_direction"].tolist(), out["end_iteration"].tolist(),
            out["remaining_light"].tolist())
