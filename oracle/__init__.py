"""CPU checkers for the trace_rays hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package, and only as the checker / reported baseline -- never as the thing measured
or shipped.  The product (volumeraytracer_amd) does not import it.

  oracle.port  -- ctypes handle on oracle/libvrt_oracle.so, our C restatement
                  (vrt_oracle.c; cites reference file:line per function).
  oracle.ref   -- ctypes handle on oracle/_ref/libvrt_ref.so, the REAL reference CPU
                  path compiled from /root/reference/src by oracle/Makefile.  Exists only
                  where that build was possible; `oracle.ref.available()` says so.
"""
import contextlib
import ctypes
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
VRT_F32, VRT_I16 = 0, 1

_c = ctypes
_u32p = _c.POINTER(_c.c_uint32)


def _ptr(a):
    return a.ctypes.data_as(_c.c_void_p) if a is not None else None


def _as(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def build(target="oracle"):
    """Run oracle/Makefile for `target` ("oracle" or "ref")."""
    subprocess.run(["make", "-C", HERE, target], check=True, stdout=subprocess.DEVNULL)


@contextlib.contextmanager
def _quiet_stdout():
    """The reference prints progress/warnings on fd 1 (cuda_volume_raytracer.cu:512-515, 91);
    keep test output readable."""
    sys.stdout.flush()
    saved = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    try:
        os.dup2(devnull, 1)
        yield
    finally:
        os.dup2(saved, 1)
        os.close(saved)
        os.close(devnull)


def _dtype_of(t):
    return np.float32 if t == VRT_F32 else np.int16


class _Port:
    """Our C restatement (vrt_oracle.c)."""

    def __init__(self):
        self._lib = None

    def lib(self):
        if self._lib is None:
            so = os.path.join(HERE, "libvrt_oracle.so")
            try:
                if not os.path.exists(so):
                    build("oracle")
                self._lib = _c.CDLL(so)
            except OSError:
                build("oracle")
                self._lib = _c.CDLL(so)
            self._lib.vrt_oracle_ray_prologue.restype = _c.c_long
        return self._lib

    def num_threads(self):
        return int(self.lib().vrt_oracle_num_threads())

    def interleave(self, dx, dy, dz, transl_cropped):
        """TraceRaysCu ctor's field build: returns the AoS {dx,dy,dz,w} array [nvox,4]."""
        dt = dx.dtype
        assert dt in (np.float32, np.int16)
        n = dx.size
        out = np.empty((n, 4), dt)
        fn = self.lib().vrt_oracle_interleave_f32 if dt == np.float32 else self.lib().vrt_oracle_interleave_i16
        fn(_ptr(_as(dx, dt)), _ptr(_as(dy, dt)), _ptr(_as(dz, dt)), _ptr(_as(transl_cropped, np.uint32)),
           _c.c_size_t(n), _ptr(out))
        return out

    def trace(self, field, bounds, start_pos, start_dir, invscale, iterations, trace_paths=False, nthreads=0,
              absorb_translucency=None, min_brightness=0):
        """Seam-level trace on an interleaved field.  Returns dict of outputs.
        absorb_translucency (uint32 per field voxel) switches on the opt-in absorption mode
        (the reference's dead branch cuda_volume_raytracer.cu:337-341; parity unpinned)."""
        field = np.ascontiguousarray(field)
        ft = VRT_F32 if field.dtype == np.float32 else VRT_I16
        start_dir = np.ascontiguousarray(start_dir)
        dtp = VRT_F32 if start_dir.dtype == np.float32 else VRT_I16
        sp = _as(start_pos, np.uint32).reshape(-1)
        n = sp.size // 3
        b = _as(bounds, np.uint16)
        inv = _as(invscale, np.float32)
        ep = np.empty(3 * n, np.uint32)
        ed = np.empty(3 * n, _dtype_of(dtp))
        ei = np.empty(n, np.uint32)
        rl = np.empty(n, np.uint32)
        path = np.empty(int(iterations) * 3 * n, np.uint32) if trace_paths else None
        if absorb_translucency is None:
            rc = self.lib().vrt_oracle_trace(ft, dtp, _ptr(field), _ptr(b), _ptr(inv), _c.c_size_t(n), _ptr(sp),
                                             _ptr(start_dir.reshape(-1)), _ptr(ep), _ptr(ed), _ptr(ei), _ptr(rl),
                                             _ptr(path), _c.c_uint32(int(iterations)), int(nthreads))
        else:
            tr = _as(absorb_translucency, np.uint32).reshape(-1)
            rc = self.lib().vrt_oracle_trace_absorb(ft, dtp, _ptr(field), _ptr(b), _ptr(inv), _c.c_size_t(n), _ptr(sp),
                                                    _ptr(start_dir.reshape(-1)), _ptr(ep), _ptr(ed), _ptr(ei), _ptr(rl),
                                                    _ptr(path), _c.c_uint32(int(iterations)), int(nthreads), _ptr(tr),
                                                    _c.c_uint32(int(min_brightness)))
        if rc != 0:
            raise RuntimeError("vrt_oracle_trace failed: %d" % rc)
        return dict(end_position=ep, end_direction=ed, end_iteration=ei, remaining_light=rl, path=path)

    def scene_build(self, bounds, ior, translucency):
        """RaytraceScene ctor (3-D): returns (dx, dy, dz, transl_cropped, diff_bounds)."""
        b = np.asarray(bounds, dtype=np.uint64)
        ob = b - 2
        n = int(np.prod(ob))
        ior = np.ascontiguousarray(ior)
        tr = _as(translucency, np.uint32)
        tc = np.empty(n, np.uint32)
        if ior.dtype == np.float32:
            dt, fn = np.float32, self.lib().vrt_oracle_scene_build_f32
        else:
            ior = _as(ior, np.uint32)
            dt, fn = np.int16, self.lib().vrt_oracle_scene_build_i16
        d = [np.empty(n, dt) for _ in range(3)]
        bb = (_c.c_size_t * 3)(*[int(x) for x in b])
        rc = fn(bb, _ptr(ior), _ptr(tr), _ptr(d[0]), _ptr(d[1]), _ptr(d[2]), _ptr(tc))
        if rc != 0:
            raise RuntimeError("scene build failed: %d" % rc)
        return d[0], d[1], d[2], tc, [int(x) for x in ob]

    def ray_prologue(self, bounds, ior, start_pos, start_dir):
        """RaytraceScene::trace_rays prologue; returns (pos, dir) copies, shifted and ior-scaled."""
        ior = np.ascontiguousarray(ior)
        it = VRT_F32 if ior.dtype == np.float32 else VRT_I16
        if it == VRT_I16:
            ior = _as(ior, np.uint32)
        pos = np.array(start_pos, dtype=np.uint32).reshape(-1).copy()
        d = np.array(start_dir).reshape(-1).copy()
        dtp = VRT_F32 if d.dtype == np.float32 else VRT_I16
        bb = (_c.c_size_t * 3)(*[int(x) for x in bounds])
        rc = self.lib().vrt_oracle_ray_prologue(it, bb, _ptr(ior), _c.c_size_t(pos.size // 3), _ptr(pos), dtp, _ptr(d))
        if rc != 0:
            raise RuntimeError("ray prologue failed: %d" % rc)
        return pos, d

    def scene_trace(self, bounds, ior, translucency, start_pos, start_dir, invscale, iterations,
                    trace_paths=False, nthreads=0):
        """Full L4 path == RaytraceScene(...).trace_rays(...) (image_util.cpp:501-772)."""
        dx, dy, dz, tc, ob = self.scene_build(bounds, ior, translucency)
        field = self.interleave(dx, dy, dz, tc)
        pos, d = self.ray_prologue(bounds, ior, start_pos, start_dir)
        out = self.trace(field, ob, pos, d, invscale, iterations, trace_paths, nthreads)
        out["end_position"] = out["end_position"] + np.uint32(0x10000)       # image_util.cpp:771
        if out["path"] is not None:
            out["path"] = out["path"] + np.uint32(0x10000)                    # image_util.cpp:770
        return out


class _Ref:
    """The real reference CPU path (oracle/_ref/libvrt_ref.so)."""

    def __init__(self):
        self._lib = None

    def path(self):
        return os.path.join(HERE, "_ref", "libvrt_ref.so")

    def available(self):
        if self._lib is not None:
            return True
        try:
            self._lib = _c.CDLL(self.path())
            self._lib.ref_last_error.restype = _c.c_char_p
            return True
        except OSError:
            return False

    def lib(self):
        if not self.available():
            raise RuntimeError("reference build not present (run `make -C oracle ref` where /root/reference exists)")
        return self._lib

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("reference raised: " + self.lib().ref_last_error().decode(errors="replace"))

    def seam_trace(self, bounds, dx, dy, dz, transl_cropped, start_pos, start_dir, invscale, iterations,
                   trace_paths=False, max_cpu=0):
        dt = dx.dtype
        start_dir = np.ascontiguousarray(start_dir).reshape(-1)
        name = "ref_seam_%s_%s" % ("f32" if dt == np.float32 else "i16",
                                   "f32" if start_dir.dtype == np.float32 else "i16")
        sp = _as(start_pos, np.uint32).reshape(-1)
        n = sp.size // 3
        ep = np.empty(3 * n, np.uint32)
        ed = np.empty(3 * n, start_dir.dtype)
        ei = np.empty(n, np.uint32)
        rl = np.empty(n, np.uint32)
        path = np.empty(int(iterations) * 3 * n, np.uint32) if trace_paths else None
        bb = (_c.c_size_t * 3)(*[int(x) for x in bounds])
        inv = _as(invscale, np.float32)
        with _quiet_stdout():
            rc = getattr(self.lib(), name)(bb, _ptr(_as(dx, dt)), _ptr(_as(dy, dt)), _ptr(_as(dz, dt)),
                                           _ptr(_as(transl_cropped, np.uint32)), _c.c_size_t(dx.size), _c.c_size_t(n),
                                           _ptr(sp), _ptr(start_dir), _ptr(ep), _ptr(ed), _ptr(ei), _ptr(rl), _ptr(path),
                                           _ptr(inv), _c.c_uint32(int(iterations)), int(max_cpu))
        self._check(rc)
        return dict(end_position=ep, end_direction=ed, end_iteration=ei, remaining_light=rl, path=path)

    def scene_trace(self, bounds, ior, translucency, start_pos, start_dir, invscale, iterations,
                    trace_paths=False, max_cpu=0):
        ior = np.ascontiguousarray(ior)
        start_dir = np.ascontiguousarray(start_dir).reshape(-1)
        if ior.dtype == np.float32:
            fn = self.lib().ref_scene_f32
            assert start_dir.dtype == np.float32
        else:
            ior = _as(ior, np.uint32)
            fn = self.lib().ref_scene_i16
            assert start_dir.dtype == np.int16
        sp = _as(start_pos, np.uint32).reshape(-1)
        n = sp.size // 3
        ep = np.empty(3 * n, np.uint32)
        ed = np.empty(3 * n, start_dir.dtype)
        ei = np.empty(n, np.uint32)
        rl = np.empty(n, np.uint32)
        path = np.empty(int(iterations) * 3 * n, np.uint32) if trace_paths else None
        bb = (_c.c_size_t * 3)(*[int(x) for x in bounds])
        inv = _as(invscale, np.float32)
        with _quiet_stdout():
            rc = fn(bb, _ptr(ior), _ptr(_as(translucency, np.uint32)), _c.c_size_t(ior.size), _c.c_size_t(n),
                    _ptr(sp), _ptr(start_dir), _ptr(ep), _ptr(ed), _ptr(ei), _ptr(rl), _ptr(path), _ptr(inv),
                    _c.c_uint32(int(iterations)), int(max_cpu))
        self._check(rc)
        return dict(end_position=ep, end_direction=ed, end_iteration=ei, remaining_light=rl, path=path)

    def scene_build(self, bounds, ior, translucency):
        b = np.asarray(bounds, dtype=np.uint64)
        ob = b - 2
        n = int(np.prod(ob))
        ior = np.ascontiguousarray(ior)
        tc = np.empty(n, np.uint32)
        if ior.dtype == np.float32:
            dt, fn, lg = np.float32, self.lib().ref_build_f32, np.empty(ior.size, np.float32)
        else:
            ior = _as(ior, np.uint32)
            dt, fn, lg = np.int16, self.lib().ref_build_i16, np.empty(ior.size, np.int32)
        d = [np.empty(n, dt) for _ in range(3)]
        bb = (_c.c_size_t * 3)(*[int(x) for x in b])
        with _quiet_stdout():
            rc = fn(bb, _ptr(ior), _ptr(_as(translucency, np.uint32)), _c.c_size_t(ior.size),
                    _ptr(d[0]), _ptr(d[1]), _ptr(d[2]), _ptr(tc), _ptr(lg))
        self._check(rc)
        return d[0], d[1], d[2], tc, [int(x) for x in ob], lg


port = _Port()
ref = _Ref()
