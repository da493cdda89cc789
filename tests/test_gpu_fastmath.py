"""GPU (-m gpu): the opt-in tolerance mode (vrt_options.fast_math = 1).  The default path is bit-exact and every other test
holds it to that; this file holds the tolerance mode to north_star's bound -- positions and directions within 1e-5
relative, end_iteration equal on >= 99.9 % of the rays and off by one otherwise (SURVEY section 7's policy) -- against the
oracle, on the seam level for both field types and both direction types, and checks that the flag really selects another
kernel result (it is not silently ignored) while leaving the default untouched."""
import numpy as np
import pytest

import oracle
from volumeraytracer_amd import capi, workloads as W

pytestmark = pytest.mark.gpu


def _check(got, want, N, ddt):
    ei_g, ei_w = got["end_iteration"].astype(np.int64), want["end_iteration"].astype(np.int64)
    assert np.abs(ei_g - ei_w).max() <= 1
    same = ei_g == ei_w
    assert same.mean() >= 0.999, same.mean()
    pg = got["end_position"].reshape(-1, 3).astype(np.int64)
    pw = want["end_position"].reshape(-1, 3).astype(np.int64)
    pg[pg >= 2 ** 31] -= 2 ** 32
    pw[pw >= 2 ** 31] -= 2 ** 32
    # 1e-5 relative to the extent of the field in position units (16.16 fixed point), on rays that took the same number of steps
    tol = 1e-5 * (N << 16) + 1
    assert np.abs(pg - pw)[same].max() <= tol, (np.abs(pg - pw)[same].max(), tol)
    dg = got["end_direction"].reshape(-1, 3).astype(np.float64)
    dw = want["end_direction"].reshape(-1, 3).astype(np.float64)
    scale = np.linalg.norm(dw, axis=1).max()
    dtol = 1e-5 * scale + (1.0 if ddt == np.int16 else 0.0)           # int16 directions: one unit of 1/256 from the final rounding
    assert np.abs(dg - dw)[same].max() <= dtol
    assert np.array_equal(got["remaining_light"], want["remaining_light"])


@pytest.mark.parametrize("fdt,ddt", [(np.float32, np.float32), (np.int16, np.int16), (np.float32, np.int16), (np.int16, np.float32)])
def test_fast_math_within_tolerance(fdt, ddt):
    N = 72
    bounds, dx, dy, dz, tc = W.seam_field(N, fdt, seed=17, opaque_radius=5.0)
    field = oracle.port.interleave(dx, dy, dz, tc)
    with capi.Engine.from_host(bounds, [dx, dy, dz], tc) as eng:
        for sp, sd in (W.perspective_rays(N, 160, dir_dtype=ddt, seed=17), W.random_rays(N, 20000, ddt, seed=5, vol_seed=17)):
            want = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], 100000)
            exact = eng.trace(sp, sd, [1, 1, 1], 100000, options=capi.default_options(kernel=capi.KERNEL_SLAB))
            for k in ("end_position", "end_iteration", "remaining_light"):
                assert np.array_equal(exact[k], want[k]), k            # the default stays bit-exact
            for sched in (1, 2, 3):
                fast = eng.trace(sp, sd, [1, 1, 1], 100000, options=capi.default_options(kernel=capi.KERNEL_SLAB, fast_math=1, schedule=sched))
                _check(fast, want, N, ddt)
            # the flag is honoured: over thousands of rays some last-place bits must differ from the exact result
            assert not (np.array_equal(fast["end_position"], want["end_position"]) and
                        np.array_equal(fast["end_direction"].view(np.uint8), want["end_direction"].view(np.uint8)))
            # kernels without a tolerance variant ignore it and stay exact
            direct = eng.trace(sp, sd, [1, 1, 1], 100000, options=capi.default_options(kernel=capi.KERNEL_DIRECT, fast_math=1))
            assert np.array_equal(direct["end_position"], want["end_position"])


def test_fast_math_non_unit_invscale_and_long_marches():
    """invscale != 1 (the other copy of the proven loop) and marches of thousands of steps (errors accumulate)."""
    N = 140
    bounds, dx, dy, dz, tc = W.seam_field(N, np.float32, seed=3)
    field = oracle.port.interleave(dx, dy, dz, tc)
    sp, sd = W.perspective_rays(N, 128, seed=3)
    for inv in ([1.0, 1.0, 1.0], [0.5, 0.5, 0.5], [0.7, 1.0, 1.3]):
        want = oracle.port.trace(field, bounds, sp, sd, inv, 100000)
        assert want["end_iteration"].astype(np.int64).mean() > 400
        with capi.Engine.from_host(bounds, [dx, dy, dz], tc) as eng:
            fast = eng.trace(sp, sd, inv, 100000, options=capi.default_options(kernel=capi.KERNEL_SLAB, fast_math=1))
        _check(fast, want, N, np.float32)
