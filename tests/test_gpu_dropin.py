"""GPU (-m gpu): the reference's OWN pybind11 module (python_binding.cpp compiled unmodified in the
build container, integration/Makefile) running on the MI355X engine through the seam shim, checked
against the golden vectors the reference's CPU build produced.  Skipped if the prebuilt module did
not travel or this host cannot load it.  Reads nothing from /root/reference at run time."""
import glob
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _load_module():
    from volumeraytracer_amd import capi
    capi.load()     # torch's HIP runtime first, then ours (same order as everywhere else)
    hits = glob.glob(os.path.join(ROOT, "integration", "_build", "cuda_raytrace*.so"))
    if not hits:
        pytest.skip("integration/_build/cuda_raytrace*.so not present (built only where /root/reference exists)")
    spec = importlib.util.spec_from_file_location("cuda_raytrace", hits[0])
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
    except ImportError as e:
        pytest.skip("cannot load the prebuilt module here: %s" % e)
    return mod


def test_reference_python_binding_on_gpu(golden_dir, tmp_path, monkeypatch):
    mod = _load_module()
    monkeypatch.chdir(tmp_path)          # python_binding.cpp:21 drops a debug file into the cwd
    g = np.load(os.path.join(golden_dir, "L4_scene_i16.npz"))
    # py_trace_rays(bound_vec, ior, translucency, start_position, start_direction, scale,
    #               minimum_brightness, iterations, trace_paths)            python_binding.cpp:10-19
    res = mod.cuda_raytrace(g["bounds"].tolist(), g["ior"].tolist(), g["translucency"].tolist(),
                            g["start_position"].tolist(), g["start_direction"].tolist(),
                            g["invscale"].tolist(), 0, int(g["iterations"]), False)
    end_position, end_direction, slot2, slot3 = res
    assert np.array_equal(np.array(end_position, np.uint32), g["end_position"])
    assert np.array_equal(np.array(end_direction, np.int16), g["end_direction"])
    # tuple slot 2 receives end_iteration, slot 3 remaining_light (python_binding.cpp:38-45)
    assert np.array_equal(np.array(slot2, np.uint32), g["end_iteration"])
    assert np.array_equal(np.array(slot3, np.uint32), g["remaining_light"])
