"""GPU (-m gpu): randomised differential test of the C-ABI march against the CPU oracle (the case generator of
tools/fuzz_gpu.py with fixed seeds): random field shapes (thin, odd, tiny), both field and direction types, random / sheet /
pinhole ray sets with zero, tiny, huge and non-finite directions mixed in, random invscale, iteration caps from 1 to 100000
and every combination of kernel, ray order, schedule and LDS segment size the options expose.  Bit equality on every ray."""
import os
import sys

import numpy as np
import pytest

import oracle
from volumeraytracer_amd import capi

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import fuzz_gpu  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("big,seed,cases", [(False, 11, 150), (True, 12, 24)])
def test_random_cases_match_the_oracle(big, seed, cases):
    rng = np.random.RandomState(seed)
    steps = 0
    for i in range(cases):
        B, d, tc, sp, sd, inv, iters, opts, fdt, ddt = fuzz_gpu.make_case(rng, big)
        field = oracle.port.interleave(d[0], d[1], d[2], tc)
        want = oracle.port.trace(field, B, sp, sd, inv, iters)
        with capi.Engine.from_host(B, d, tc) as eng:
            got = eng.trace(sp, sd, inv, iters, options=capi.default_options(**opts))
        ctx = "case %d: B=%s field=%s dir=%s rays=%d inv=%s iters=%d opts=%s" % (i, B, fdt.__name__, ddt.__name__, sp.size // 3, inv, iters, opts)
        for k in ("end_position", "end_iteration", "remaining_light"):
            assert np.array_equal(got[k], want[k]), (k, ctx)
        assert np.array_equal(got["end_direction"].view(np.uint8), want["end_direction"].view(np.uint8)), ctx
        steps += int(want["end_iteration"].astype(np.int64).sum())
    assert steps > (5e7 if big else 1e6)            # the cases really march


def _scene_case(rng):
    """Random scene-layer case: a non-cubic ior volume (float or 16.16 fixed point), a few opaque voxels, rays inside."""
    N = [int(rng.randint(5, 44)) for _ in range(3)]
    if rng.rand() < 0.3:
        N[rng.randint(0, 3)] = int(rng.randint(4, 8))
    x, y, z = np.meshgrid(np.arange(N[0]), np.arange(N[1]), np.arange(N[2]), indexing="ij")
    bump = np.zeros(x.shape)
    for _ in range(rng.randint(1, 4)):
        c = rng.rand(3) * np.array(N)
        s2 = (rng.uniform(0.15, 0.5) * min(N)) ** 2
        bump += rng.uniform(0.1, 0.6) * np.exp(-((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) / s2)
    ior = 1.0 + bump
    integer = rng.rand() < 0.5
    ior = np.round(ior * 65536).astype(np.uint32).reshape(-1) if integer else ior.astype(np.float32).reshape(-1)
    n = N[0] * N[1] * N[2]
    tr = np.full(n, 0xFFFFFFFF, np.uint32)
    if rng.rand() < 0.5:
        tr[rng.rand(n) < rng.choice([0.002, 0.03])] = rng.randint(0, 1 << 30)
    m = int(rng.choice([1, 64, 333, 2000, 6000]))
    pv = 1.0 + rng.rand(m, 3) * (np.array(N) - 2.001)
    sp = np.round(pv * 65536).astype(np.uint32).reshape(-1)
    d = rng.randn(m, 3)
    d /= np.linalg.norm(d, axis=1)[:, None]
    sd = np.round(d * 256).astype(np.int16).reshape(-1) if integer else d.astype(np.float32).reshape(-1)
    inv = [1.0, 1.0, 1.0] if rng.rand() < 0.6 else [float(2.0 ** rng.randint(-1, 2)) for _ in range(3)]
    iters = int(rng.choice([5, 60, 100000]))
    paths = bool(iters <= 60 and rng.rand() < 0.5)
    opts = dict(kernel=int(rng.choice([0, 1, 2])), sort_rays=int(rng.choice([0, -1, 1])), schedule=int(rng.choice([0, 1, 2, 3])))
    return N, ior, tr, sp, sd, inv, iters, paths, opts


def test_random_scenes_match_the_oracle():
    """The scene layer end to end (GPU scene build, ray prologue in place, march, +0x10000 epilogue, optional paths) on random
    non-cubic volumes of both type families against the oracle's restatement of RaytraceScene (image_util.cpp:501-772)."""
    rng = np.random.RandomState(33)
    for i in range(40):
        N, ior, tr, sp, sd, inv, iters, paths, opts = _scene_case(rng)
        want = oracle.port.scene_trace(N, ior, tr, sp, sd, inv, iters, trace_paths=paths)
        with capi.Scene(N, ior, tr) as sc:
            got = sc.trace_rays(sp, sd, inv, iters, trace_paths=paths, options=capi.default_options(**opts))
        ctx = "scene case %d: N=%s ior=%s rays=%d inv=%s iters=%d paths=%s opts=%s" % (i, N, ior.dtype, sp.size // 3, inv, iters, paths, opts)
        for k in ("end_position", "end_iteration", "remaining_light"):
            assert np.array_equal(got[k], want[k]), (k, ctx)
        assert np.array_equal(got["end_direction"].view(np.uint8), want["end_direction"].view(np.uint8)), ctx
        if paths:
            assert np.array_equal(got["path"], want["path"]), ctx
