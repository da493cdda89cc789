"""Replay driver == the reference's raytrace_test CLI (/root/reference/src/raytrace_test.cpp:33-72)
on the MI355X engine.

    python -m volumeraytracer_amd.replay debug_scene_instance debug_ray_instance [--float]

Reads the two dump files the reference writes with Options::_write_instance
(java_binding.cpp:119-124, 189-194), builds the scene, traces the rays and prints the begin / end
positions like the reference tool (raytrace_test.cpp:69-70)."""
import argparse
import sys

import numpy as np

from . import capi, serialize


def replay(scene_path, ray_path, float_family=False):
    ior_dt, dir_dt = (np.float32, np.float32) if float_family else (np.uint32, np.int16)
    sc = serialize.read_scene_instance(scene_path, ior_dt)
    ry = serialize.read_ray_instance(ray_path, dir_dt)
    with capi.Scene([int(b) for b in sc["bound_vec"]], sc["ior"], sc["translucency"]) as scene:
        out = scene.trace_rays(ry["start_position"], ry["start_direction"], ry["invscale"], ry["iterations"],
                               ry["trace_path"], ry["minimum_brightness"], ry["normalize_length"])
    return sc, ry, out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("scene")
    ap.add_argument("rays")
    ap.add_argument("--float", action="store_true", help="float ior / float directions (default: ior_t / dir_t)")
    a = ap.parse_args(argv)
    _, ry, out = replay(a.scene, a.rays, a.float)
    print("begin " + " ".join(str(int(x)) for x in ry["start_position"]))      # raytrace_test.cpp:69
    print("end " + " ".join(str(int(x)) for x in out["end_position"]))         # raytrace_test.cpp:70
    return 0


if __name__ == "__main__":
    sys.exit(main())
