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
