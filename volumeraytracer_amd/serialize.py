"""Reader/writer for the reference's debug record/replay files.

Format (/root/reference/src/serialize.h:12-79, image_util.cpp:35-144, image_util.h:236-246):
  vector<T>  = size_t count (8 bytes, little endian) + count raw elements
  scalars    = raw bytes (uint32 4, bool 1)
  debug_scene_instance = RayTraceSceneInstance<Ior>: bound_vec (vector<size_t>), ior (vector<Ior>),
                         translucency (vector<uint32>)
  debug_ray_instance   = RayTraceRayInstance<Dir>: start_position (vector<uint32>), start_direction
                         (vector<Dir>), invscale (vector<float>), minimum_brightness (uint32),
                         iterations (uint32), trace_path (bool), normalize_length (bool)
  RaytraceInstance<Ior,Dir> = the scene fields followed by the ray fields.
The files the reference writes are the integer family (java_binding.cpp:119-124, 189-194:
ior_t / dir_t) unless the float scene was used; the element types are not recorded in the file, so
the caller names them.
"""
import struct

import numpy as np


def _read_vec(f, dtype):
    (n,) = struct.unpack("<Q", _need(f, 8))
    dt = np.dtype(dtype).newbyteorder("<")
    return np.frombuffer(_need(f, n * dt.itemsize), dtype=dt).copy()


def _need(f, n):
    b = f.read(n)
    if len(b) != n:
        raise RuntimeError("Bad input Stream")      # serialize.h:17
    return b


def _write_vec(f, a, dtype):
    a = np.ascontiguousarray(a, dtype=np.dtype(dtype).newbyteorder("<"))
    f.write(struct.pack("<Q", a.size))
    f.write(a.tobytes())


def read_scene_instance(path_or_file, ior_dtype=np.uint32):
    f = open(path_or_file, "rb") if isinstance(path_or_file, (str, bytes)) else path_or_file
    try:
        return dict(bound_vec=_read_vec(f, np.uint64), ior=_read_vec(f, ior_dtype), translucency=_read_vec(f, np.uint32))
    finally:
        if isinstance(path_or_file, (str, bytes)):
            f.close()


def write_scene_instance(path_or_file, bound_vec, ior, translucency):
    f = open(path_or_file, "wb") if isinstance(path_or_file, (str, bytes)) else path_or_file
    try:
        ior = np.asarray(ior)
        _write_vec(f, bound_vec, np.uint64)
        _write_vec(f, ior, np.float32 if ior.dtype == np.float32 else np.uint32)
        _write_vec(f, translucency, np.uint32)
    finally:
        if isinstance(path_or_file, (str, bytes)):
            f.close()


def read_ray_instance(path_or_file, dir_dtype=np.int16):
    f = open(path_or_file, "rb") if isinstance(path_or_file, (str, bytes)) else path_or_file
    try:
        out = dict(start_position=_read_vec(f, np.uint32), start_direction=_read_vec(f, dir_dtype),
                   invscale=_read_vec(f, np.float32))
        out["minimum_brightness"], out["iterations"] = struct.unpack("<II", _need(f, 8))
        out["trace_path"], out["normalize_length"] = (bool(b) for b in struct.unpack("<BB", _need(f, 2)))
        return out
    finally:
        if isinstance(path_or_file, (str, bytes)):
            f.close()


def write_ray_instance(path_or_file, start_position, start_direction, invscale, minimum_brightness, iterations,
                       trace_path, normalize_length):
    f = open(path_or_file, "wb") if isinstance(path_or_file, (str, bytes)) else path_or_file
    try:
        sd = np.asarray(start_direction)
        _write_vec(f, start_position, np.uint32)
        _write_vec(f, sd, np.float32 if sd.dtype == np.float32 else np.int16)
        _write_vec(f, invscale, np.float32)
        f.write(struct.pack("<IIBB", int(minimum_brightness), int(iterations), int(bool(trace_path)),
                            int(bool(normalize_length))))
    finally:
        if isinstance(path_or_file, (str, bytes)):
            f.close()


def read_raytrace_instance(path, ior_dtype=np.uint32, dir_dtype=np.int16):
    """RaytraceInstance<Ior,Dir>: scene fields then ray fields in one file (image_util.cpp:59-73)."""
    with open(path, "rb") as f:
        out = read_scene_instance(f, ior_dtype)
        out.update(read_ray_instance(f, dir_dtype))
    return out
