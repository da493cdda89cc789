"""volumeraytracer_amd -- MI355X-native trace_rays engine (host-side mirror of the reference seam).

Compute happens only in the HIP library behind include/vrt.h (volumeraytracer_amd.capi);
there is no CPU fallback.  See DESIGN.md / INTEGRATION.md.
"""
from . import capi, workloads  # noqa: F401
from .capi import Engine, Options, Scene, VrtError, default_options, device_count  # noqa: F401

__all__ = ["capi", "workloads", "Engine", "Scene", "Options", "VrtError", "default_options", "device_count"]
