"""CPU, build container only: the drop-in link.  The reference's scene layer and its unmodified
python_binding.cpp are compiled from /root/reference/src and linked against our seam shim
(integration/reference_seam_shim.cpp) + libvrt_hip.so instead of cuda_volume_raytracer.o."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference tree not present")


def test_reference_links_against_engine():
    subprocess.run(["make", "-C", os.path.join(ROOT, "integration")], check=True, stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, "integration", "_build", "libraytracer_vrt.so")
    syms = subprocess.run(["nm", "-C", "--defined-only", so], check=True, capture_output=True, text=True).stdout
    # the six symbols image_util.o imports, now defined by the shim
    for needle in ("TraceRaysCu<short>::TraceRaysCu(", "TraceRaysCu<short>::~TraceRaysCu()",
                   "void TraceRaysCu<short>::trace_rays_cu<short>(", "TraceRaysCu<float>::TraceRaysCu(",
                   "TraceRaysCu<float>::~TraceRaysCu()", "void TraceRaysCu<float>::trace_rays_cu<float>("):
        assert needle in syms, needle
    undefined = subprocess.run(["nm", "-D", "--undefined-only", so], check=True, capture_output=True, text=True).stdout
    for needle in ("vrt_create", "vrt_trace", "vrt_destroy"):
        assert needle in undefined
    # nothing of the reference's CPU march is linked in
    assert "trace_rays_cpu" not in syms
