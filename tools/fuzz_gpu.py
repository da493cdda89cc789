#!/usr/bin/env python3
"""Randomised differential test of the GPU march against the CPU oracle (development aid, run through gpurun).

    python3 tools/fuzz_gpu.py --seconds 240 [--seed 1]

Every case draws a field shape (thin, odd, tiny, up to ~60 voxels per axis), a field type, a direction type, a ray set (random,
axis-parallel sheets, a pinhole fan; with zero / tiny / huge / non-finite directions mixed in), invscale, the iteration cap, and
the engine options (kernel, ray order, schedule, LDS segment size); the C-ABI result must equal the oracle's bit for bit.
Prints one line per failure with everything needed to reproduce it, and a summary."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from volumeraytracer_amd import capi  # noqa: E402


def make_case(rng, big=False):
    kind = rng.randint(0, 5)
    if big:                                          # long marches, many wavefronts: schedules, many rounds per ray
        B = [int(rng.randint(40, 140)) for _ in range(3)]
        if kind == 0:
            B[rng.randint(0, 3)] = int(rng.randint(3, 12))
    elif kind == 0:
        B = [int(rng.randint(2, 9)), int(rng.randint(2, 60)), int(rng.randint(2, 60))]
    elif kind == 1:
        B = [int(rng.randint(2, 60)), int(rng.randint(2, 9)), int(rng.randint(2, 60))]
    elif kind == 2:
        B = [int(rng.randint(2, 60)), int(rng.randint(2, 60)), int(rng.randint(2, 9))]
    else:
        B = [int(rng.randint(3, 48)) for _ in range(3)]
    n = B[0] * B[1] * B[2]
    fdt = np.float32 if rng.rand() < 0.5 else np.int16
    amp = 10.0 ** (rng.uniform(0, 3.0) if big else rng.uniform(0, 4.3))
    x, y, z = np.meshgrid(np.arange(B[0]), np.arange(B[1]), np.arange(B[2]), indexing="ij")
    d = []
    for k in range(3):
        smooth = np.sin(x * rng.uniform(0.05, 0.6) + rng.rand()) * np.cos(y * rng.uniform(0.05, 0.6)) * np.sin(z * rng.uniform(0.05, 0.6) + 1.0)
        v = smooth * amp + rng.randn(*x.shape) * amp * rng.choice([0.0, 0.01, 0.3])
        d.append((v.astype(np.float32) if fdt == np.float32 else np.clip(np.round(v), -32768, 32767).astype(np.int16)).reshape(-1))
    tc = np.full(n, 0xFFFFFFFF, np.uint32)
    if rng.rand() < 0.5:
        tc[rng.rand(n) < rng.choice([0.001, 0.02, 0.2])] = rng.randint(0, 2 ** 31)
    m = int(rng.choice([4096, 20000, 60000])) if big else int(rng.choice([1, 7, 63, 64, 65, 200, 1000, 4096, 9000]))
    mode = rng.randint(0, 3)
    hi = np.maximum(np.array(B) - 1.001, 0.01)
    if mode == 0:
        pv = rng.rand(m, 3) * hi
        dirs = rng.randn(m, 3)
    elif mode == 1:                                  # an axis-parallel sheet entering through one face
        ax = rng.randint(0, 3)
        pv = rng.rand(m, 3) * hi
        pv[:, ax] = rng.choice([0.0, 0.3, hi[ax]])
        dirs = np.zeros((m, 3))
        dirs[:, ax] = 1.0 if pv[0, ax] < hi[ax] else -1.0
        dirs += rng.randn(m, 3) * rng.choice([0.0, 0.01, 0.2])
    else:                                            # pinhole fan
        eye = -np.array(B, float) * rng.uniform(0.2, 1.5) * np.eye(3)[rng.randint(0, 3)] + hi / 2
        tgt = rng.rand(m, 3) * hi
        dirs = tgt - eye
        pv = tgt
    nrm = np.linalg.norm(dirs, axis=1)
    dirs = dirs / np.where(nrm > 0, nrm, 1.0)[:, None]
    scale = 10.0 ** rng.uniform(-0.3, 0.3, m) if rng.rand() < 0.7 else 10.0 ** rng.uniform(-6, 2, m)
    dirs = dirs * scale[:, None]
    sp = np.round(pv * 65536).astype(np.uint32).reshape(-1)
    ddt = np.float32 if rng.rand() < 0.7 else np.int16
    if ddt == np.float32:
        sd = dirs.astype(np.float32)
        if rng.rand() < 0.3:
            k = rng.randint(0, m, max(1, m // 50))
            sd[k] = rng.choice([0.0, np.inf, -np.inf, np.nan, 3e38, 1e-30, 1e-20], size=(k.size, 1)).astype(np.float32) * np.ones(3, np.float32)
            sd[k, rng.randint(0, 3)] = 0.0
        sd = sd.reshape(-1)
    else:
        sd = np.clip(np.round(dirs * 256.0), -32768, 32767).astype(np.int16).reshape(-1)
    inv = [1.0, 1.0, 1.0] if rng.rand() < 0.5 else [float(10.0 ** rng.uniform(-1, 1)) for _ in range(3)]
    iters = int(rng.choice([300, 2000, 100000])) if big else int(rng.choice([1, 2, 3, 17, 100, 1000, 5000]))
    opts = dict(kernel=int(rng.choice([0, 1, 2])), sort_rays=int(rng.choice([0, -1, 1, 2, 3])), schedule=int(rng.choice([0, 1, 2, 3])),
                slab_lds_bytes=int(rng.choice([0, 0, 4096, 6144, 10240, 16384])))
    return B, d, tc, sp, sd, inv, iters, opts, fdt, ddt


def scene_cases(rng, seconds, seed):
    """--scenes: random volume extents (1 .. 140 output voxels per axis, so that every row shift, partial tiles and partial
    slabs occur), array and LUT sources of both families; the field the tiled scene build leaves in HBM must equal the
    oracle's scene build bit for bit (or both must refuse the scene)."""
    t0, cases, fails, vox = time.time(), 0, 0, 0
    while time.time() - t0 < seconds:
        dims = [int(rng.randint(3, 40)), int(rng.randint(3, 80)), int(rng.randint(3, 143))]
        if rng.rand() < 0.3:
            rng.shuffle(dims)
        kind = str(rng.choice(["f32", "f32_lut", "u32", "u32_lut"]))
        nvox = dims[0] * dims[1] * dims[2]
        is_f = kind.startswith("f32")
        lut_ior = (1.0 + rng.rand(256) * 0.8).astype(np.float32) if is_f else np.round((1.0 + rng.rand(256) * 0.8) * 65536).astype(np.uint32)
        lut_tr = np.where(rng.rand(256) < 0.8, 0xFFFFFFFF, rng.randint(0, 2 ** 31, 256)).astype(np.uint32)
        dens = None
        if kind.endswith("lut"):
            dens = np.clip(np.cumsum(rng.randint(-1, 2, nvox)), 0, 255).astype(np.uint8)
            ior, tr = lut_ior[dens], lut_tr[dens]
        else:
            base = 1.0 + 0.3 * rng.rand(nvox)
            ior = base.astype(np.float32) if is_f else np.round(base * 65536).astype(np.uint32)
            tr = np.where(rng.rand(nvox) < 0.9, 0xFFFFFFFF, rng.randint(0, 2 ** 31, nvox)).astype(np.uint32)
        bad = []
        try:
            hx, hy, hz, tc, _ = oracle.port.scene_build(list(dims), ior, tr)
            want = np.asarray(oracle.port.interleave(hx, hy, hz, tc))
        except Exception:                                   # noqa: BLE001  (the oracle raises what the reference throws)
            want = None
        try:
            sc = capi.Scene.from_lut(list(dims), dens, lut_ior, lut_tr) if dens is not None else capi.Scene(list(dims), ior, tr)
            with sc:
                got = sc.engine().download_field()
            if want is None:
                bad.append("the oracle refuses the scene, the GPU builds it")
            elif not np.array_equal(got.reshape(-1).view(np.uint8), want.reshape(-1).view(np.uint8)):
                bad.append("field differs")
        except capi.VrtError as e:
            if want is not None:
                bad.append("exception %r" % (e,))
        cases += 1
        vox += nvox
        if bad:
            fails += 1
            print("FAIL scene %d: %s dims=%s kind=%s" % (cases, bad, dims, kind), flush=True)
    print("scene fuzz: %d scenes, %d failures, %.3g voxels checked in %.0f s (seed %d)" % (cases, fails, vox, time.time() - t0, seed))
    sys.exit(1 if fails else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", action="store_true", help="fields of 40..140 voxels per axis, 4096..60000 rays, long marches")
    ap.add_argument("--scenes", action="store_true", help="fuzz the scene build (row f-1) instead of the march")
    args = ap.parse_args()
    rng = np.random.RandomState(args.seed)
    if args.scenes:
        scene_cases(rng, args.seconds, args.seed)
    t0, cases, fails, steps = time.time(), 0, 0, 0
    while time.time() - t0 < args.seconds:
        state = rng.get_state()[1][:4].tolist()
        B, d, tc, sp, sd, inv, iters, opts, fdt, ddt = make_case(rng, args.big)
        field = oracle.port.interleave(d[0], d[1], d[2], tc)
        want = oracle.port.trace(field, B, sp, sd, inv, iters)
        # round 3: the handle sometimes spans several engines (band cut + merge, on one device), the ray order is sometimes
        # the radix sort of rounds 1 and 2 instead of the tile-count kernels
        reps = int(rng.choice([0, 0, 2, 3]))
        order = str(rng.choice(["tiles", "tiles", "radix"]))
        os.environ.pop("VRT_DEBUG_REPLICAS", None)
        if reps:
            os.environ["VRT_DEBUG_REPLICAS"] = str(reps)
        os.environ["VRT_RAY_ORDER"] = order
        opts_desc = dict(opts, replicas=reps, order=order)
        try:
            with capi.Engine.from_host(B, d, tc) as eng:
                got = eng.trace(sp, sd, inv, iters, options=capi.default_options(**opts))
            bad = [k for k in ("end_position", "end_iteration", "remaining_light") if not np.array_equal(got[k], want[k])]
            a, b = got["end_direction"], want["end_direction"]
            if not np.array_equal(a.view(np.uint8), b.view(np.uint8)):
                bad.append("end_direction")
        except Exception as e:                              # noqa: BLE001
            bad = ["exception %r" % (e,)]
        cases += 1
        steps += int(want["end_iteration"].astype(np.int64).sum())
        if bad:
            fails += 1
            print("FAIL case %d: %s  B=%s field=%s dir=%s rays=%d inv=%s iters=%d opts=%s rng-head=%s" %
                  (cases, bad, B, fdt.__name__, ddt.__name__, sp.size // 3, inv, iters, opts_desc, state), flush=True)
        if cases % 200 == 0:
            print("... %d cases, %d failures, %.3g ray-steps, %.0f s" % (cases, fails, steps, time.time() - t0), flush=True)
    print("fuzz: %d cases, %d failures, %.3g ray-steps checked in %.0f s (seed %d)" % (cases, fails, steps, time.time() - t0, args.seed))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
