"""ctypes binding of the C ABI in include/vrt.h (libvrt_hip.so).

This is the *only* compute path of the package: if the HIP library is missing or no
gfx950 device is usable, calls raise -- there is no CPU fallback (and nothing here
imports oracle/).
"""
import ctypes as C
import os

import numpy as np

from . import _build

VRT_F32, VRT_I16, VRT_U32 = 0, 1, 2
KERNEL_AUTO, KERNEL_DIRECT, KERNEL_SLAB = 0, 1, 2

# every symbol include/vrt.h declares (checked by tests/test_abi.py against the header text)
ABI_SYMBOLS = (
    "vrt_options_default", "vrt_device_count", "vrt_abi_version", "vrt_create", "vrt_create_from_device",
    "vrt_trace", "vrt_trace_device", "vrt_destroy", "vrt_engine_device", "vrt_engine_field_bytes",
    "vrt_engine_field_device_ptr", "vrt_engine_last_march_ms", "vrt_engine_debug_stats", "vrt_selftest_ilen2", "vrt_selftest_round", "vrt_selftest_stage_division",
    "vrt_last_error", "vrt_scene_create", "vrt_scene_create_lut", "vrt_scene_trace", "vrt_scene_destroy", "vrt_scene_engine",
    "vrt_engine_last_march_clock_ghz", "vrt_engine_replica_count", "vrt_debug_band_cut", "vrt_scene_trace_device",
    "vrt_scene_trace_status", "vrt_trace_status", "vrt_engine_last_schedule",
    "vrt_engine_download_field", "vrt_engine_field_scan",
)


class VrtError(RuntimeError):
    """A C-ABI call returned a negative vrt_status (message from vrt_last_error())."""

    def __init__(self, code, msg):
        super().__init__("%s (vrt_status %d)" % (msg, code))
        self.code = code


class Options(C.Structure):
    """struct vrt_options == the reference's struct Options (types.h:83-91) + extensions."""
    _fields_ = [("loglevel", C.c_int32), ("minimum_gpu", C.c_uint64), ("write_instance", C.c_int32),
                ("max_cpu", C.c_uint64), ("device", C.c_int32), ("kernel", C.c_int32),
                ("sort_rays", C.c_int32), ("slab_lds_bytes", C.c_int32), ("absorption", C.c_int32), ("schedule", C.c_int32), ("devices", C.c_int32), ("host_lanes", C.c_int32), ("fast_math", C.c_int32)]


_lib = None


def lib_path():
    # VRT_LIB selects another build of the same library (A/B runs of kernel variants in one snapshot)
    return os.environ.get("VRT_LIB") or _build.LIB


def load():
    """Load libvrt_hip.so (never builds implicitly on import of the package; raises if absent)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64
    # (same SONAME as /opt/rocm's).  If our library pulled in the system copy first, torch would
    # later fail with "No HIP GPUs are available".  torch is only plumbing here (device memory,
    # streams, torch.distributed), but it has to come first.
    if os.environ.get("VRT_NO_TORCH_PRELOAD", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(path):
        raise ImportError("libvrt_hip.so not built: run `python -m volumeraytracer_amd._build` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(path)
    vp, u32p, sz = C.c_void_p, C.c_void_p, C.c_size_t
    L.vrt_options_default.argtypes = [C.POINTER(Options)]
    L.vrt_options_default.restype = None
    L.vrt_device_count.restype = C.c_int
    L.vrt_abi_version.restype = C.c_int
    L.vrt_last_error.restype = C.c_char_p
    L.vrt_create.argtypes = [C.POINTER(vp), C.c_int, vp, C.c_int, C.POINTER(vp), u32p, sz, C.POINTER(Options)]
    L.vrt_create_from_device.argtypes = L.vrt_create.argtypes
    L.vrt_trace.argtypes = [vp, C.c_int, sz, u32p, vp, u32p, vp, u32p, u32p, u32p, vp, C.c_uint32, C.c_uint32,
                            C.POINTER(Options)]
    L.vrt_trace_device.argtypes = [vp, C.c_int, sz, u32p, vp, u32p, vp, u32p, u32p, u32p, vp, C.c_uint32,
                                   C.c_uint32, vp, C.POINTER(Options)]
    L.vrt_destroy.argtypes = [vp]
    L.vrt_destroy.restype = None
    L.vrt_engine_device.argtypes = [vp]
    L.vrt_engine_field_bytes.argtypes = [vp]
    L.vrt_engine_field_bytes.restype = sz
    L.vrt_engine_field_device_ptr.argtypes = [vp]
    L.vrt_engine_field_device_ptr.restype = vp
    L.vrt_engine_debug_stats.argtypes = [vp, C.POINTER(C.c_ulonglong), C.c_int]
    if hasattr(L, "vrt_selftest_ilen2"):            # absent only in older A/B builds selected through VRT_LIB
        L.vrt_selftest_ilen2.argtypes = [C.c_int, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    L.vrt_engine_last_march_ms.argtypes = [vp]
    L.vrt_engine_last_march_ms.restype = C.c_float
    if hasattr(L, "vrt_engine_last_march_clock_ghz"):
        L.vrt_engine_last_march_clock_ghz.argtypes = [vp]
        L.vrt_engine_last_march_clock_ghz.restype = C.c_float
        L.vrt_engine_replica_count.argtypes = [vp]
        L.vrt_debug_band_cut.argtypes = [sz, sz, C.POINTER(C.c_size_t)]
        L.vrt_scene_trace_device.argtypes = [vp, C.c_int, sz, u32p, vp, u32p, vp, u32p, u32p, u32p, vp, C.c_uint32,
                                             C.c_uint32, vp, C.POINTER(Options)]
        L.vrt_scene_trace_status.argtypes = [vp]
    if hasattr(L, "vrt_trace_status"):               # ABI 3
        L.vrt_trace_status.argtypes = [vp]
        L.vrt_engine_last_schedule.argtypes = [vp]
        L.vrt_engine_download_field.argtypes = [vp, vp, sz]
        L.vrt_engine_field_scan.argtypes = [vp, vp, vp]
    if hasattr(L, "vrt_scene_create"):
        L.vrt_scene_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_size_t), C.c_int, vp, u32p, sz,
                                       C.POINTER(Options)]
        L.vrt_scene_create_lut.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_size_t), C.c_int, vp, vp, u32p, sz,
                                           C.POINTER(Options)]
        L.vrt_scene_trace.argtypes = [vp, C.c_int, sz, u32p, vp, u32p, vp, u32p, u32p, u32p, vp, C.c_uint32,
                                      C.c_uint32, C.c_int, C.POINTER(Options)]
        L.vrt_scene_destroy.argtypes = [vp]
        L.vrt_scene_destroy.restype = None
        L.vrt_scene_engine.argtypes = [vp]
        L.vrt_scene_engine.restype = vp
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise VrtError(rc, load().vrt_last_error().decode(errors="replace"))


def default_options(**kw):
    o = Options()
    load().vrt_options_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def device_count():
    return int(load().vrt_device_count())


def band_cut(n, parts):
    """The contiguous cut of n rays into `parts` bands used by multi-device handles and the chunk pipeline."""
    b = (C.c_size_t * (int(parts) + 1))()
    check(load().vrt_debug_band_cut(int(n), int(parts), b))
    return [int(x) for x in b]


def selftest_ilen2(device=0):
    """Exhaustive device check of the short division in the march step -> (mismatches, admitted patterns)."""
    bad, adm = C.c_ulonglong(0), C.c_ulonglong(0)
    check(load().vrt_selftest_ilen2(int(device), C.byref(bad), C.byref(adm)))
    return int(bad.value), int(adm.value)


def selftest_round(device=0):
    """Exhaustive device check of the proven loop's pos += roundf(s) -> (mismatches, checked patterns)."""
    bad, n = C.c_ulonglong(0), C.c_ulonglong(0)
    L = load()
    L.vrt_selftest_round.argtypes = [C.c_int, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    check(L.vrt_selftest_round(int(device), C.byref(bad), C.byref(n)))
    return int(bad.value), int(n.value)


def selftest_stage_division(device=0):
    """Exhaustive device check of the staging loop's division by multiplication -> (mismatches, checked pairs)."""
    bad, n = C.c_ulonglong(0), C.c_ulonglong(0)
    L = load()
    L.vrt_selftest_stage_division.argtypes = [C.c_int, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    check(L.vrt_selftest_stage_division(int(device), C.byref(bad), C.byref(n)))
    return int(bad.value), int(n.value)


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _type_of(dt):
    dt = np.dtype(dt)
    if dt == np.float32:
        return VRT_F32
    if dt == np.int16:
        return VRT_I16
    raise TypeError("expected float32 or int16, got %s" % dt)


class Engine:
    """Owning handle on a vrt_engine (a field resident in HBM on one device)."""

    def __init__(self, handle, field_type, bounds, owned=True):
        self._h = handle
        self.field_type = field_type
        self.bounds = [int(b) for b in bounds]
        self._owned = owned     # False: a view on the engine inside a Scene (the scene frees it)

    @classmethod
    def from_host(cls, bounds, diff, transl_cropped, options=None):
        """vrt_create: diff = 3 host arrays (float32 or int16), transl_cropped uint32."""
        L = load()
        ft = _type_of(diff[0].dtype)
        d = [np.ascontiguousarray(x, dtype=diff[0].dtype).reshape(-1) for x in diff]
        tr = np.ascontiguousarray(transl_cropped, dtype=np.uint32).reshape(-1)
        b = np.ascontiguousarray(bounds, dtype=np.uint16)
        if len(b) != len(d):
            raise VrtError(-2, "Illegal dimension")
        ptrs = (C.c_void_p * len(d))(*[x.ctypes.data for x in d])
        h = C.c_void_p()
        check(L.vrt_create(C.byref(h), len(b), _np_ptr(b), ft, ptrs, _np_ptr(tr), d[0].size,
                           C.byref(options) if options is not None else None))
        return cls(h, ft, b)

    @classmethod
    def from_device(cls, bounds, diff_ptrs, transl_ptr, field_type, nvox, options=None):
        """vrt_create_from_device: raw device pointers (ints)."""
        L = load()
        b = np.ascontiguousarray(bounds, dtype=np.uint16)
        ptrs = (C.c_void_p * len(diff_ptrs))(*[int(p) for p in diff_ptrs])
        h = C.c_void_p()
        check(L.vrt_create_from_device(C.byref(h), len(b), _np_ptr(b), field_type, ptrs, C.c_void_p(int(transl_ptr)),
                                       int(nvox), C.byref(options) if options is not None else None))
        return cls(h, field_type, b)

    @property
    def handle(self):
        return self._h

    @property
    def device(self):
        return int(load().vrt_engine_device(self._h))

    @property
    def field_bytes(self):
        return int(load().vrt_engine_field_bytes(self._h))

    def debug_raw(self, n):
        """First n raw 64-bit debug words of the last launch (needs VRT_DEBUG_STATS=1): 64 counters, then the
        per-wavefront begin/end clocks of a -DVRT_SLAB_TIMELINE development build."""
        buf = (C.c_ulonglong * int(n))()
        check(load().vrt_engine_debug_stats(self._h, buf, int(n)))
        return np.frombuffer(buf, dtype=np.uint64).copy()

    def debug_stats(self):
        """Slab-kernel debug counters of the last launch (needs VRT_DEBUG_STATS=1)."""
        buf = (C.c_ulonglong * 18)()
        check(load().vrt_engine_debug_stats(self._h, buf, 18))
        names = ("rounds", "planned_K_sum", "lds_wave_steps", "direct_rounds", "direct_wave_steps", "vox_staged",
                 "early_exit", "proven_rounds", "fb_ex_sum", "fb_ey_sum", "fb_ez_sum", "fb_live_lanes_sum", "fb_steps_done_sum", "proven_steps", "fb_fit2_K8", "fb_fit4_K8",
                 "fb_fit2_K16", "fb_fit4_K16")
        return dict(zip(names, [int(x) for x in buf]))

    def last_march_ms(self):
        """HIP-event duration of the most recent march kernel (waits for it)."""
        return float(load().vrt_engine_last_march_ms(self._h))

    def last_march_clock_ghz(self):
        """Shader clock the most recent march kernel ran at (s_memtime / s_memrealtime of one of its wavefronts)."""
        return float(load().vrt_engine_last_march_clock_ghz(self._h))

    def download_field(self):
        """The resident interleaved field as a host array [nvox, 4] (float32 or int16)."""
        dt = np.float32 if self.field_type == VRT_F32 else np.int16
        out = np.empty((self.field_bytes // (4 * np.dtype(dt).itemsize), 4), dt)
        check(load().vrt_engine_download_field(self._h, _np_ptr(out), out.nbytes))
        return out

    def field_scan(self):
        """(max|dx|, max|dy|, max|dz|, bound of the longest gradient vector), flags -- from the engine's field scan."""
        b = np.zeros(4, np.float32)
        f = np.zeros(1, np.uint32)
        check(load().vrt_engine_field_scan(self._h, _np_ptr(b), _np_ptr(f)))
        return b, int(f[0])

    def last_schedule(self):
        """Schedule of the most recent march launch: 1 static, 2 persistent, 3 time-sliced, 0 direct kernel / none."""
        return int(load().vrt_engine_last_schedule(self._h))

    def status(self):
        """vrt_trace_status: wait for the device-pointer calls issued so far and raise if the time-sliced schedule gave
        up a wait (results incomplete)."""
        check(load().vrt_trace_status(self._h))

    @property
    def replica_count(self):
        """Devices (engines) behind the handle: 1 + replicas made for options.devices."""
        return int(load().vrt_engine_replica_count(self._h))

    def trace(self, start_pos, start_dir, invscale, iterations, trace_paths=False, min_brightness=0, options=None):
        """vrt_trace on host numpy arrays; returns dict of numpy outputs."""
        L = load()
        sd = np.ascontiguousarray(start_dir).reshape(-1)
        dt = _type_of(sd.dtype)
        sp = np.ascontiguousarray(start_pos, dtype=np.uint32).reshape(-1)
        if sp.size != sd.size or sp.size % 3:
            raise VrtError(-1, "raycounts doesn't match, dimension is: 3 raysizes are start_position: %d "
                               "start_direction: %d" % (sp.size, sd.size))
        n = sp.size // 3
        inv = np.ascontiguousarray(invscale, dtype=np.float32)
        ep = np.empty(3 * n, np.uint32)
        ed = np.empty(3 * n, sd.dtype)
        ei = np.empty(n, np.uint32)
        rl = np.empty(n, np.uint32)
        path = np.empty(int(iterations) * 3 * n, np.uint32) if trace_paths else None
        check(L.vrt_trace(self._h, dt, n, _np_ptr(sp), _np_ptr(sd), _np_ptr(ep), _np_ptr(ed), _np_ptr(ei),
                          _np_ptr(rl), _np_ptr(path) if path is not None else None, _np_ptr(inv),
                          int(min_brightness), int(iterations), C.byref(options) if options is not None else None))
        return dict(end_position=ep, end_direction=ed, end_iteration=ei, remaining_light=rl, path=path)

    def trace_device(self, dir_type, nrays, d_start_pos, d_start_dir, d_end_pos, d_end_dir, d_end_iter,
                     d_remaining_light, invscale, iterations, d_path=0, stream=0, min_brightness=0, options=None, check_status=False):
        """vrt_trace_device on raw device pointers (ints); asynchronous on `stream` unless check_status is set, which
        waits for the call and raises on a scheduler failure (vrt_trace_status)."""
        L = load()
        inv = np.ascontiguousarray(invscale, dtype=np.float32)
        check(L.vrt_trace_device(self._h, dir_type, int(nrays), C.c_void_p(int(d_start_pos)),
                                 C.c_void_p(int(d_start_dir)), C.c_void_p(int(d_end_pos)), C.c_void_p(int(d_end_dir)),
                                 C.c_void_p(int(d_end_iter)), C.c_void_p(int(d_remaining_light)),
                                 C.c_void_p(int(d_path)) if d_path else None, _np_ptr(inv), int(min_brightness),
                                 int(iterations), C.c_void_p(int(stream)) if stream else None,
                                 C.byref(options) if options is not None else None))
        if check_status:
            self.status()

    def close(self):
        if self._h is not None and _lib is not None and self._owned:
            _lib.vrt_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Scene:
    """Owning handle on a vrt_scene == the reference's RaytraceScene<Ior,IorLog,Diff>
    (image_util.h:132-195): ior volume -> derivative field on the GPU; trace_rays with the ray
    front end (range check, ior normalisation, coordinate shifts) on the GPU."""

    def __init__(self, bounds, ior, translucency, options=None):
        L = load()
        ior = np.ascontiguousarray(ior).reshape(-1)
        if ior.dtype == np.float32:
            self.ior_type = VRT_F32
        else:
            ior = np.ascontiguousarray(ior, dtype=np.uint32)
            self.ior_type = VRT_U32
        tr = np.ascontiguousarray(translucency, dtype=np.uint32).reshape(-1)
        if tr.size != ior.size:
            raise VrtError(-6, "imagesizes doesn't match:%d=%d=%d (bounds ior translucency)"
                           % (int(np.prod([int(b) for b in bounds])), ior.size, tr.size))
        b = (C.c_size_t * len(bounds))(*[int(x) for x in bounds])
        h = C.c_void_p()
        check(L.vrt_scene_create(C.byref(h), len(bounds), b, self.ior_type, _np_ptr(ior), _np_ptr(tr), ior.size,
                                 C.byref(options) if options is not None else None))
        self._h = h
        self.bounds = [int(x) for x in bounds]

    @classmethod
    def from_lut(cls, bounds, density, lut_ior, lut_translucency, options=None):
        """vrt_scene_create_lut: uint8 density volume + 256-entry (ior, translucency) transfer function."""
        L = load()
        self = cls.__new__(cls)
        dens = np.ascontiguousarray(density, dtype=np.uint8).reshape(-1)
        lut_ior = np.ascontiguousarray(lut_ior).reshape(-1)
        if lut_ior.dtype == np.float32:
            self.ior_type = VRT_F32
        else:
            lut_ior = np.ascontiguousarray(lut_ior, dtype=np.uint32)
            self.ior_type = VRT_U32
        lut_tr = np.ascontiguousarray(lut_translucency, dtype=np.uint32).reshape(-1)
        if lut_ior.size != 256 or lut_tr.size != 256:
            raise VrtError(-1, "transfer function tables need 256 entries")
        b = (C.c_size_t * len(bounds))(*[int(x) for x in bounds])
        h = C.c_void_p()
        check(L.vrt_scene_create_lut(C.byref(h), len(bounds), b, self.ior_type, _np_ptr(dens), _np_ptr(lut_ior),
                                     _np_ptr(lut_tr), dens.size, C.byref(options) if options is not None else None))
        self._h = h
        self.bounds = [int(x) for x in bounds]
        return self

    def engine(self):
        """Non-owning Engine view on the seam-level engine inside the scene (valid while the scene lives)."""
        h = C.c_void_p(load().vrt_scene_engine(self._h))
        return Engine(h, VRT_F32 if self.ior_type == VRT_F32 else VRT_I16, [b - 2 for b in self.bounds], owned=False)

    def trace_rays(self, start_position, start_direction, invscale, iterations, trace_paths=False,
                   minimum_brightness=0, normalize_length=True, options=None):
        L = load()
        sd = np.ascontiguousarray(start_direction).reshape(-1)
        dt = _type_of(sd.dtype)
        sp = np.ascontiguousarray(start_position, dtype=np.uint32).reshape(-1)
        if sp.size != sd.size or sp.size % 3:
            raise VrtError(-1, "raycounts doesn't match, dimension is: 3 raysizes are start_position: %d "
                               "start_direction: %d" % (sp.size, sd.size))      # image_util.cpp:664-667
        n = sp.size // 3
        inv = np.ascontiguousarray(invscale, dtype=np.float32)
        ep = np.empty(3 * n, np.uint32)
        ed = np.empty(3 * n, sd.dtype)
        ei = np.empty(n, np.uint32)
        rl = np.empty(n, np.uint32)
        path = np.empty(int(iterations) * 3 * n, np.uint32) if trace_paths else None
        check(L.vrt_scene_trace(self._h, dt, n, _np_ptr(sp), _np_ptr(sd), _np_ptr(ep), _np_ptr(ed), _np_ptr(ei),
                                _np_ptr(rl), _np_ptr(path) if path is not None else None, _np_ptr(inv),
                                int(minimum_brightness), int(iterations), 1 if normalize_length else 0,
                                C.byref(options) if options is not None else None))
        return dict(end_position=ep, end_direction=ed, end_iteration=ei, remaining_light=rl, path=path)

    def trace_rays_device(self, dir_type, nrays, d_start_pos, d_start_dir, d_end_pos, d_end_dir, d_end_iter,
                          d_remaining_light, invscale, iterations, d_path=0, stream=0, minimum_brightness=0, options=None):
        """vrt_scene_trace_device on raw device pointers (ints); asynchronous on `stream`.  Ray errors surface in
        status()."""
        L = load()
        inv = np.ascontiguousarray(invscale, dtype=np.float32)
        check(L.vrt_scene_trace_device(self._h, dir_type, int(nrays), C.c_void_p(int(d_start_pos)),
                                       C.c_void_p(int(d_start_dir)), C.c_void_p(int(d_end_pos)), C.c_void_p(int(d_end_dir)),
                                       C.c_void_p(int(d_end_iter)), C.c_void_p(int(d_remaining_light)),
                                       C.c_void_p(int(d_path)) if d_path else None, _np_ptr(inv), int(minimum_brightness),
                                       int(iterations), C.c_void_p(int(stream)) if stream else None,
                                       C.byref(options) if options is not None else None))

    def status(self):
        """Wait for the last trace_rays_device call and raise its ray error, if any."""
        check(load().vrt_scene_trace_status(self._h))

    def close(self):
        if self._h is not None and _lib is not None:
            _lib.vrt_scene_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
