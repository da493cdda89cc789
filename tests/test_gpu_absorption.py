"""GPU (-m gpu): the OPT-IN absorption mode (vrt_options.absorption = 1) -- the branch the reference
compiles out (cuda_volume_raytracer.cu:337-341; translucency is a DummyArray at every call site).
PARITY UNPINNED by construction: no reference build can execute it, so the checker is the oracle's
restatement of those lines (oracle/vrt_oracle.c, trace_one).  What IS pinned: with a fully translucent
volume the mode is a no-op and must reproduce the default mode bit for bit."""
import numpy as np
import pytest

import oracle
from volumeraytracer_amd import capi, workloads as W

pytestmark = pytest.mark.gpu
KEYS = ("end_position", "end_direction", "end_iteration", "remaining_light")


def _same(a, b):
    return a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.parametrize("kernel", [capi.KERNEL_DIRECT, capi.KERNEL_SLAB])
@pytest.mark.parametrize("fdt", [np.float32, np.int16])
def test_absorption_vs_restatement(fdt, kernel):
    N = 48
    bounds, dx, dy, dz, _ = W.seam_field(N, fdt)
    rng = np.random.RandomState(5)
    n = dx.size
    # smooth-ish absorber: most voxels lose 1/200 of full scale per step, some are nearly opaque
    tr = np.full(n, 0xFFFFFFFF - 0x01400000, np.uint32)
    tr[rng.rand(n) < 0.02] = 0x20000000
    tr[rng.rand(n) < 0.30] = 0xFFFFFFFF
    field = oracle.port.interleave(dx, dy, dz, tr)
    sp, sd = W.perspective_rays(N, 96)
    on = capi.default_options(absorption=1, kernel=kernel)
    with capi.Engine.from_host(bounds, [dx, dy, dz], tr, options=capi.default_options(absorption=1)) as eng:
        for minb in (0, 0x10000000, 0x80000000, 0xFFFFFFFF):
            want = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], 100000, absorb_translucency=tr, min_brightness=minb)
            got = eng.trace(sp, sd, [1, 1, 1], 100000, min_brightness=minb, options=on)
            for k in KEYS:
                assert _same(got[k], want[k]), (k, hex(minb))
            if kernel == capi.KERNEL_SLAB:              # persistent schedule + longest-first order carry the brightness too
                got = eng.trace(sp, sd, [1, 1, 1], 100000, min_brightness=minb,
                                options=capi.default_options(absorption=1, kernel=kernel, schedule=2, sort_rays=3))
                for k in KEYS:
                    assert _same(got[k], want[k]), (k, hex(minb), "persistent")
        # brightness only ever decreases, and a higher cut-off never lets a ray go further
        a = eng.trace(sp, sd, [1, 1, 1], 100000, min_brightness=0x10000000, options=on)
        b = eng.trace(sp, sd, [1, 1, 1], 100000, min_brightness=0x80000000, options=on)
        assert np.all(b["end_iteration"] <= a["end_iteration"])
        assert np.all(a["remaining_light"] <= 0xFFFFFFFF)
        # without the trace-time flag the engine behaves exactly like the reference: min_brightness ignored
        plain = eng.trace(sp, sd, [1, 1, 1], 100000, min_brightness=0x80000000, options=capi.default_options(kernel=kernel))
        ref = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], 100000)
        for k in KEYS:
            assert _same(plain[k], ref[k]), k


def test_absorption_noop_on_translucent_volume_and_errors():
    N = 32
    bounds, dx, dy, dz, tc = W.seam_field(N, np.float32, opaque_radius=3.0)
    tc_clear = np.where(tc == 0, 0, 0xFFFFFFFF).astype(np.uint32)     # opaque cores stay (w channel), rest fully translucent
    sp, sd = W.random_rays(N, 2000)
    with capi.Engine.from_host(bounds, [dx, dy, dz], tc_clear, options=capi.default_options(absorption=1)) as eng:
        plain = eng.trace(sp, sd, [1, 1, 1], 100000)
        ab = eng.trace(sp, sd, [1, 1, 1], 100000, min_brightness=0, options=capi.default_options(absorption=1))
        for k in KEYS[:3]:
            assert _same(plain[k], ab[k]), k
    with capi.Engine.from_host(bounds, [dx, dy, dz], tc_clear) as eng2:          # created without the flag
        with pytest.raises(capi.VrtError):
            eng2.trace(sp, sd, [1, 1, 1], 100, options=capi.default_options(absorption=1))


def test_absorption_through_scene_layer():
    """vrt_scene_create with absorption keeps the CROPPED translucency (image_util.cpp:519)."""
    N = 28
    ior = W.ior_volume(N, dtype=np.float32).reshape(-1)
    rng = np.random.RandomState(2)
    tr = np.where(rng.rand(N ** 3) < 0.5, 0xFFFFFFFF, 0xFFFFFFFF - 0x02000000).astype(np.uint32)
    pv = 1.0 + rng.rand(700, 3) * (N - 2.001)
    sp = np.round(pv * 65536).astype(np.uint32).reshape(-1)
    d = rng.randn(700, 3)
    d /= np.linalg.norm(d, axis=1)[:, None]
    sd = d.astype(np.float32).reshape(-1)
    dx, dy, dz, tc, ob = oracle.port.scene_build([N] * 3, ior, tr)
    field = oracle.port.interleave(dx, dy, dz, tc)
    pos, dd = oracle.port.ray_prologue([N] * 3, ior, sp, sd)
    want = oracle.port.trace(field, ob, pos, dd, [1, 1, 1], 100000, absorb_translucency=tc, min_brightness=0x30000000)
    on = capi.default_options(absorption=1)
    with capi.Scene([N] * 3, ior, tr, options=on) as sc:
        got = sc.trace_rays(sp, sd, [1, 1, 1], 100000, minimum_brightness=0x30000000, options=on)
    assert np.array_equal(got["end_position"], want["end_position"] + np.uint32(0x10000))
    assert np.array_equal(got["end_iteration"], want["end_iteration"])
    assert np.array_equal(got["remaining_light"], want["remaining_light"])
