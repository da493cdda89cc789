#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference CPU build.

Run in the build container only (needs /root/reference, compiled by `make -C oracle ref`
into oracle/_ref/libvrt_ref.so).  The reference cannot travel to the GPU box, so the
vectors are committed as small .npz fixtures (inputs + the reference's outputs).

    python tests/golden/make_golden.py

Fixtures:
  F1_scaling_{f32,i16}   the reference's own scaling_test (cuda_volume_raytracer_test.h:10-52):
                         1000x10x10 ior ramp 1->2, two rays, invscale 2, cap 1e6 -- scene level.
  F2_seam_<field>_<dir>  24^3 smooth field, 768 random rays, seam level, all four type combos,
                         three invscale settings; plus path tracing for 32 rays x 48 iterations.
  F3_opaque_<field>      opaque cores (translucency 0 inside a radius): pins the w>0 termination.
  F4_caps                iteration caps 1, 2, 3, 17 on the F2 rays: pins end_iteration/remaining.
  L4_scene_{f32,i16}     20^3 ior volume through RaytraceScene (scene build + ray front end + trace),
                         with the reference's derivative arrays for the scene-build check.
  F5_lut_{f32,i16}       BASELINE config 5 semantics (SURVEY §0 reading A) at 32^3: uint8 density + 256-entry
                         (ior, translucency) transfer function, expanded on the host and fed to the
                         reference; opaque cores terminate rays through the live w>0 test.
  F6_config1             BASELINE config 1 (64^3, 256x256 orthographic, int16 family via the
                         python_binding.cpp instantiation): SHA-256 of the outputs + summary stats.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from volumeraytracer_amd import workloads as W  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
KEYS = ("end_position", "end_direction", "end_iteration", "remaining_light")


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    print("wrote", name, {k: (v.shape, str(v.dtype)) for k, v in arrs.items() if hasattr(v, "shape")})


def scaling_inputs(dt_ior, dt_dir):
    """cuda_volume_raytracer_test.h:10-28"""
    b, n = [1000, 10, 10], 100
    unit = 1.0 if dt_ior == np.float32 else 0x10000
    ior = np.zeros(1000 * n, dt_ior)
    ior[:10 * n] = unit
    ior[-10 * n:] = unit * 2
    for i in range(10, 990):
        if dt_ior == np.float32:
            v = np.float32(1.0) + np.float32(1.0) * np.float32(i) / np.float32(979)
        else:
            v = 0x10000 + (0x10000 * i) // 979
        ior[i * n:(i + 1) * n] = v
    tr = np.full(1000 * n, 0xFFFFFFFF, np.uint32)
    sp = np.array([0x10000, 0x40000, 0x40000, 0x10000 * 1000 - 0x30000, 0x40000, 0x40000], np.uint32)
    xdir = 16.0 if dt_dir == np.float32 else 0x10 * 0x100
    sd = np.array([xdir, 0, 0, -xdir, 0, 0], dt_dir)
    return b, ior, tr, sp, sd


def lut_fixture(ref):
    N = 32
    dens, lut_ior_f, lut_tr = W.uint8_density_and_lut(N)
    dens = dens.reshape(-1)
    rng = np.random.RandomState(31)
    pv = 1.0 + rng.rand(640, 3) * (N - 2.001)
    sp = np.round(pv * 65536).astype(np.uint32).reshape(-1)
    d = rng.randn(640, 3)
    d /= np.linalg.norm(d, axis=1)[:, None]
    for tag in ("f32", "i16"):
        if tag == "f32":
            lut_ior = lut_ior_f
            sd = d.astype(np.float32).reshape(-1)
        else:
            lut_ior = np.round(lut_ior_f.astype(np.float64) * 65536).astype(np.uint32)
            sd = np.round(d * 256).astype(np.int16).reshape(-1)
        ior = lut_ior[dens]                     # the expansion the reference needs
        tr = lut_tr[dens]
        r = ref.scene_trace([N] * 3, ior, tr, sp, sd, [1, 1, 1], 100000, False)
        save("F5_lut_" + tag, bounds=np.array([N] * 3), density=dens, lut_ior=lut_ior, lut_translucency=lut_tr,
             start_position=sp, start_direction=sd, invscale=np.float32([1, 1, 1]), iterations=np.uint32(100000),
             **{k: r[k] for k in KEYS})


def main():
    assert oracle.ref.available(), "build the reference first: make -C oracle ref"
    ref = oracle.ref
    lut_fixture(ref)
    if "--only-lut" in sys.argv:
        return

    # ---- F1
    for tag, dti, dtd in (("f32", np.float32, np.float32), ("i16", np.uint32, np.int16)):
        b, ior, tr, sp, sd = scaling_inputs(dti, dtd)
        r = ref.scene_trace(b, ior, tr, sp, sd, [2, 2, 2], 1000000, False)
        save("F1_scaling_" + tag, bounds=np.array(b), ior=ior, translucency=tr, start_position=sp,
             start_direction=sd, invscale=np.float32([2, 2, 2]), iterations=np.uint32(1000000),
             **{k: r[k] for k in KEYS})

    # ---- F2 / F4
    N = 24
    for ftag, fdt in (("f32", np.float32), ("i16", np.int16)):
        bounds, dx, dy, dz, tc = W.seam_field(N, fdt)
        for dtag, ddt in (("f32", np.float32), ("i16", np.int16)):
            sp, sd = W.random_rays(N, 768, ddt)
            arrs = dict(bounds=np.array(bounds), dx=dx, dy=dy, dz=dz, transl_cropped=tc,
                        start_position=sp, start_direction=sd)
            for j, inv in enumerate(([1, 1, 1], [2, 2, 2], [0.5, 1.0, 1.5])):
                r = ref.seam_trace(bounds, dx, dy, dz, tc, sp, sd, inv, 100000, False)
                arrs["invscale%d" % j] = np.float32(inv)
                for k in KEYS:
                    arrs["%s_%d" % (k, j)] = r[k]
            r = ref.seam_trace(bounds, dx, dy, dz, tc, sp[:96], sd[:96], [1, 1, 1], 48, True)
            for k in KEYS + ("path",):
                arrs[k + "_path48"] = r[k]
            for cap in (1, 2, 3, 17):
                r = ref.seam_trace(bounds, dx, dy, dz, tc, sp, sd, [1, 1, 1], cap, False)
                for k in KEYS:
                    arrs["%s_cap%d" % (k, cap)] = r[k]
            save("F2_seam_%s_%s" % (ftag, dtag), **arrs)

    # ---- F3
    for ftag, fdt in (("f32", np.float32), ("i16", np.int16)):
        bounds, dx, dy, dz, tc = W.seam_field(N, fdt, opaque_radius=4.0)
        sp, sd = W.random_rays(N, 768, np.float32, seed=99)
        r = ref.seam_trace(bounds, dx, dy, dz, tc, sp, sd, [1, 1, 1], 100000, False)
        save("F3_opaque_" + ftag, bounds=np.array(bounds), dx=dx, dy=dy, dz=dz, transl_cropped=tc,
             start_position=sp, start_direction=sd, invscale=np.float32([1, 1, 1]),
             iterations=np.uint32(100000), **{k: r[k] for k in KEYS})

    # ---- L4 scene level
    Ns = 20
    for tag, dti, dtd in (("f32", np.float32, np.float32), ("i16", np.uint32, np.int16)):
        ior = W.ior_volume(Ns, dtype=dti).reshape(-1)
        rng = np.random.RandomState(7)
        tr = np.full(Ns ** 3, 0xFFFFFFFF, np.uint32)
        tr[rng.rand(Ns ** 3) < 0.02] = 0x10000000           # a few opaque voxels
        pv = 1.0 + rng.rand(512, 3) * (Ns - 2.001)
        sp = np.round(pv * 65536).astype(np.uint32).reshape(-1)
        d = rng.randn(512, 3)
        d /= np.linalg.norm(d, axis=1)[:, None]
        sd = d.astype(np.float32).reshape(-1) if dtd == np.float32 else np.round(d * 256).astype(np.int16).reshape(-1)
        r = ref.scene_trace([Ns] * 3, ior, tr, sp, sd, [1, 1, 1], 100000, False)
        rp = ref.scene_trace([Ns] * 3, ior, tr, sp[:48], sd[:48], [1, 1, 1], 40, True)
        bx, by, bz, tc, ob, lg = ref.scene_build([Ns] * 3, ior, tr)
        save("L4_scene_" + tag, bounds=np.array([Ns] * 3), ior=ior, translucency=tr, start_position=sp,
             start_direction=sd, invscale=np.float32([1, 1, 1]), iterations=np.uint32(100000),
             ref_dx=bx, ref_dy=by, ref_dz=bz, ref_transl_cropped=tc, ref_ior_log=lg,
             path_path40=rp["path"], end_position_path40=rp["end_position"],
             **{k: r[k] for k in KEYS})

    # ---- F6: BASELINE config 1 shape through the python_binding instantiation (int16 family)
    N6, R6 = 64, 256
    ior = W.ior_volume(N6, dtype=np.uint32).reshape(-1)
    tr = np.full(N6 ** 3, 0xFFFFFFFF, np.uint32)
    t = np.linspace(1.5, N6 - 2.5, R6)
    yy, zz = np.meshgrid(t, t, indexing="ij")
    pv = np.stack([np.full(R6 * R6, 1.5), yy.reshape(-1), zz.reshape(-1)], 1)
    sp = np.round(pv * 65536).astype(np.uint32).reshape(-1)
    sd = np.zeros((R6 * R6, 3), np.int16)
    sd[:, 0] = 0x100
    sd = sd.reshape(-1)
    r = ref.scene_trace([N6] * 3, ior, tr, sp, sd, [1, 1, 1], 100000, False)
    summary = {
        "config": "64^3 uint32 ior (workloads.ior_volume seed 1234), 256x256 ortho rays +x from x=1.5, int16 dirs, invscale 1, cap 100000",
        "sum_end_iteration": int(r["end_iteration"].astype(np.int64).sum()),
        "mean_end_iteration": float(r["end_iteration"].mean()),
    }
    for k in KEYS:
        summary["sha256_" + k] = hashlib.sha256(np.ascontiguousarray(r[k]).tobytes()).hexdigest()
    with open(os.path.join(OUT, "F6_config1.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print("wrote F6_config1.json", summary["sum_end_iteration"])


if __name__ == "__main__":
    main()
