"""GPU (-m gpu): the HIP engine, called through the C ABI (include/vrt.h), against
  (a) the golden vectors produced by the real reference (tests/golden/*.npz), and
  (b) the oracle (oracle/vrt_oracle.c) on seeded inputs at sizes it finishes in seconds,
  (c) size-independent properties at BASELINE config sizes.
Bar: bit-exact (positions/iterations are integers; directions are fp32 computed with the
reference CPU build's operation order -- stronger than the 1e-5 relative tolerance north_star allows).
Nothing here reads /root/reference."""
import os

import numpy as np
import pytest

import oracle
from volumeraytracer_amd import capi, workloads as W

pytestmark = pytest.mark.gpu
KEYS = ("end_position", "end_direction", "end_iteration", "remaining_light")
KERNELS = [capi.KERNEL_DIRECT, capi.KERNEL_SLAB]


def _same(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def _assert_same(got, want, keys=KEYS, ctx=""):
    for k in keys:
        if not _same(got[k], want[k]):
            bad = np.flatnonzero(got[k].reshape(-1) != want[k].reshape(-1))
            raise AssertionError("%s %s: %d of %d differ, first at %d: got %r want %r" % (
                ctx, k, bad.size, got[k].size, bad[0], got[k].reshape(-1)[bad[0]], want[k].reshape(-1)[bad[0]]))


def _opts(kernel, sort=0, schedule=0):
    return capi.default_options(kernel=kernel, sort_rays=sort, schedule=schedule)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def test_device_present():
    assert capi.device_count() >= 1, "no HIP device: the GPU tests must not pass vacuously"


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("ftag", ["f32", "i16"])
@pytest.mark.parametrize("dtag", ["f32", "i16"])
def test_golden_seam(golden_dir, ftag, dtag, kernel):
    g = _load(golden_dir, "F2_seam_%s_%s" % (ftag, dtag))
    with capi.Engine.from_host(g["bounds"], [g["dx"], g["dy"], g["dz"]], g["transl_cropped"]) as eng:
        for sort in (-1, 1):
            for j in range(3):
                got = eng.trace(g["start_position"], g["start_direction"], g["invscale%d" % j], 100000,
                                options=_opts(kernel, sort))
                _assert_same(got, {k: g["%s_%d" % (k, j)] for k in KEYS}, ctx="inv%d sort%d" % (j, sort))
            for cap in (1, 2, 3, 17):
                got = eng.trace(g["start_position"], g["start_direction"], [1, 1, 1], cap, options=_opts(kernel, sort))
                _assert_same(got, {k: g["%s_cap%d" % (k, cap)] for k in KEYS}, ctx="cap%d" % cap)
        got = eng.trace(g["start_position"][:96], g["start_direction"][:96], [1, 1, 1], 48, trace_paths=True,
                        options=_opts(kernel))
        _assert_same(got, {k: g[k + "_path48"] for k in KEYS + ("path",)}, keys=KEYS + ("path",), ctx="path")


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("ftag", ["f32", "i16"])
def test_golden_opaque(golden_dir, ftag, kernel):
    g = _load(golden_dir, "F3_opaque_" + ftag)
    with capi.Engine.from_host(g["bounds"], [g["dx"], g["dy"], g["dz"]], g["transl_cropped"]) as eng:
        got = eng.trace(g["start_position"], g["start_direction"], g["invscale"], int(g["iterations"]),
                        options=_opts(kernel))
        _assert_same(got, g)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("tag", ["f32", "i16"])
def test_golden_scaling_seam(golden_dir, tag, kernel):
    """The reference's scaling_test (46718 +- 100 steps): scene build + ray front end by the
    oracle (host-side rows, SURVEY f-1/f-2), the march itself on the GPU."""
    g = _load(golden_dir, "F1_scaling_" + tag)
    dx, dy, dz, tc, ob = oracle.port.scene_build(g["bounds"], g["ior"], g["translucency"])
    pos, d = oracle.port.ray_prologue(g["bounds"], g["ior"], g["start_position"], g["start_direction"])
    with capi.Engine.from_host(ob, [dx, dy, dz], tc) as eng:
        got = eng.trace(pos, d, g["invscale"], int(g["iterations"]), options=_opts(kernel))
    got["end_position"] = got["end_position"] + np.uint32(0x10000)
    _assert_same(got, g)
    assert abs(int(got["end_iteration"][0]) - 46718) <= 100 and abs(int(got["end_iteration"][1]) - 46718) <= 100


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("fdt", [np.float32, np.int16])
@pytest.mark.parametrize("ddt", [np.float32, np.int16])
def test_vs_oracle_coherent(fdt, ddt, kernel):
    """Coherent bundles (the bench's ray generators) at a size the oracle does in seconds."""
    N = 72
    bounds, dx, dy, dz, tc = W.seam_field(N, fdt, opaque_radius=6.0)
    field = oracle.port.interleave(dx, dy, dz, tc)
    with capi.Engine.from_host(bounds, [dx, dy, dz], tc) as eng:
        for gen, R in ((W.ortho_rays, 150), (W.perspective_rays, 131)):
            sp, sd = gen(N, R, dir_dtype=ddt)
            for inv in ([1, 1, 1], [2, 2, 2]):
                want = oracle.port.trace(field, bounds, sp, sd, inv, 100000)
                got = eng.trace(sp, sd, inv, 100000, options=_opts(kernel))
                _assert_same(got, want, ctx="%s inv=%s" % (gen.__name__, inv))


@pytest.mark.parametrize("kernel", KERNELS)
def test_vs_oracle_incoherent(kernel):
    """Random starts and directions: every wavefront is incoherent (slab kernel's global-gather fallback)."""
    N = 40
    bounds, dx, dy, dz, tc = W.seam_field(N, np.float32, seed=3, opaque_radius=4.0)
    field = oracle.port.interleave(dx, dy, dz, tc)
    sp, sd = W.random_rays(N, 5000, seed=17, vol_seed=3)
    want = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], 100000)
    with capi.Engine.from_host(bounds, [dx, dy, dz], tc) as eng:
        for sort in (-1, 1, 2, 3):
            got = eng.trace(sp, sd, [1, 1, 1], 100000, options=_opts(kernel, sort))
            _assert_same(got, want, ctx="sort%d" % sort)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("fdt", [np.float32, np.int16])
@pytest.mark.parametrize("ddt", [np.float32, np.int16])
def test_longest_first_wavefront_order(fdt, ddt, kernel):
    """sort_rays = 3: tile order + wavefronts launched longest-first from the coarse-march estimate (the
    auto mode above 24 wavefronts per CU).  Only the launch order may change -- results stay bit-identical,
    including a ragged last wavefront, NaN / inf directions and rays that start outside the field."""
    N = 48
    bounds, dx, dy, dz, tc = W.seam_field(N, fdt, opaque_radius=5.0)
    field = oracle.port.interleave(dx, dy, dz, tc)
    sp, sd = W.perspective_rays(N, 75, dir_dtype=ddt)                     # 5625 rays = 87 wavefronts + 57 rays
    sp, sd = sp.copy(), sd.copy()
    sp.reshape(-1, 3)[3::211] = 0xFFF00000                                 # outside the field
    if ddt == np.float32:
        sd.reshape(-1, 3)[5::173, 1] = np.nan
        sd.reshape(-1, 3)[7::191, 0] = np.inf
        sd.reshape(-1, 3)[11::197] = 0.0
    for inv, cap in (([1, 1, 1], 100000), ([0.5, 1, 2], 700)):
        want = oracle.port.trace(field, bounds, sp, sd, inv, cap)
        with capi.Engine.from_host(bounds, [dx, dy, dz], tc) as eng:
            got = eng.trace(sp, sd, inv, cap, options=_opts(kernel, 3))
            _assert_same(got, want, ctx="inv=%s cap=%d" % (inv, cap))


@pytest.mark.parametrize("kernel", KERNELS)
def test_edge_cases(kernel):
    N = 20
    bounds, dx, dy, dz, tc = W.seam_field(N, np.float32)
    field = oracle.port.interleave(dx, dy, dz, tc)
    with capi.Engine.from_host(bounds, [dx, dy, dz], tc) as eng:
        # empty ray set
        got = eng.trace(np.zeros(0, np.uint32), np.zeros(0, np.float32), [1, 1, 1], 10, options=_opts(kernel))
        assert got["end_iteration"].size == 0
        # ragged counts (not a multiple of the 64-lane wavefront / 256-thread block)
        for n in (1, 63, 65, 257):
            sp, sd = W.random_rays(N, n, seed=n)
            want = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], 1000)
            _assert_same(eng.trace(sp, sd, [1, 1, 1], 1000, options=_opts(kernel)), want, ctx="n=%d" % n)
        # rays that start outside the field, on the last valid cell, and with zero direction
        B = bounds[0]
        sp = np.array([0xFFFF0000, 0x20000, 0x20000,
                       (B - 2) * 0x10000 + 0xFFFF, (B - 2) * 0x10000 + 0xFFFF, (B - 2) * 0x10000 + 0xFFFF,
                       (B - 1) * 0x10000, 0x20000, 0x20000,
                       0x28000, 0x28000, 0x28000], np.uint32)
        sd = np.array([1, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0], np.float32)
        want = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], 50)
        _assert_same(eng.trace(sp, sd, [1, 1, 1], 50, options=_opts(kernel)), want, ctx="edges")
        # iterations == 0 is rejected, never executed
        with pytest.raises(capi.VrtError):
            eng.trace(sp, sd, [1, 1, 1], 0, options=_opts(kernel))


def test_path_chunking_and_caps():
    """Path tracing in several device chunks (budget-bounded), iteration cap hit by every ray."""
    N = 24
    bounds, dx, dy, dz, tc = W.seam_field(N, np.int16)
    field = oracle.port.interleave(dx, dy, dz, tc)
    sp, sd = W.ortho_rays(N, 40, dir_dtype=np.int16)
    want = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], 30, trace_paths=True)
    with capi.Engine.from_host(bounds, [dx, dy, dz], tc) as eng:
        got = eng.trace(sp, sd, [1, 1, 1], 30, trace_paths=True)
    _assert_same(got, want, keys=KEYS + ("path",))
    assert np.all(got["end_iteration"] == 30)


def test_path_tracing_in_many_chunks(monkeypatch):
    """vrt_trace cuts path tracing into ray chunks bounded by a device path budget (2 GiB by default, like
    the reference's 0x8000-ray chunks, .cu:798-840); shrink the budget so 300 rays need 7 chunks."""
    N = 24
    bounds, dx, dy, dz, tc = W.seam_field(N, np.float32, opaque_radius=3.0)
    field = oracle.port.interleave(dx, dy, dz, tc)
    sp, sd = W.random_rays(N, 300, seed=8)
    cap = 70
    want = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], cap, trace_paths=True)
    monkeypatch.setenv("VRT_PATH_BUDGET_BYTES", str(45 * cap * 12))      # 45 rays per chunk
    with capi.Engine.from_host(bounds, [dx, dy, dz], tc) as eng:
        got = eng.trace(sp, sd, [1, 1, 1], cap, trace_paths=True, options=capi.default_options(loglevel=-1))
    _assert_same(got, want, keys=KEYS + ("path",))


def test_create_from_device_and_trace_device():
    """Device-pointer entry points (what bench.py times): same bits as the host path."""
    torch = pytest.importorskip("torch")
    N = 64
    bounds, dx, dy, dz, tc = W.seam_field(N, np.float32)
    field = oracle.port.interleave(dx, dy, dz, tc)
    sp, sd = W.perspective_rays(N, 128)
    want = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], 100000)
    dev = torch.device("cuda:0")
    t = [torch.from_numpy(a).to(dev) for a in (dx, dy, dz)]
    ttc = torch.from_numpy(tc.view(np.int32)).to(dev)
    n = sp.size // 3
    eng = capi.Engine.from_device(bounds, [x.data_ptr() for x in t], ttc.data_ptr(), capi.VRT_F32, dx.size)
    dsp = torch.from_numpy(sp.view(np.int32)).to(dev)
    dsd = torch.from_numpy(sd).to(dev)
    ep = torch.empty(3 * n, dtype=torch.int32, device=dev)
    ed = torch.empty(3 * n, dtype=torch.float32, device=dev)
    ei = torch.empty(n, dtype=torch.int32, device=dev)
    rl = torch.empty(n, dtype=torch.int32, device=dev)
    for kernel in KERNELS:
        eng.trace_device(capi.VRT_F32, n, dsp.data_ptr(), dsd.data_ptr(), ep.data_ptr(), ed.data_ptr(), ei.data_ptr(),
                         rl.data_ptr(), [1, 1, 1], 100000, stream=torch.cuda.current_stream().cuda_stream,
                         options=_opts(kernel))
        torch.cuda.synchronize()
        got = dict(end_position=ep.cpu().numpy().view(np.uint32), end_direction=ed.cpu().numpy(),
                   end_iteration=ei.cpu().numpy().view(np.uint32), remaining_light=rl.cpu().numpy().view(np.uint32))
        _assert_same(got, want, ctx="kernel %d" % kernel)
    eng.close()


def test_config2_properties():
    """BASELINE config 2 size (256^3 fp32, 512x512 perspective): both kernels agree bit-for-bit
    with each other on every ray and with the oracle on a strided sample; ordering invariance."""
    N, R = 256, 512
    bounds, dx, dy, dz, tc = W.seam_field(N, np.float32)
    sp, sd = W.perspective_rays(N, R)
    with capi.Engine.from_host(bounds, [dx, dy, dz], tc) as eng:
        a = eng.trace(sp, sd, [1, 1, 1], 100000, options=_opts(capi.KERNEL_DIRECT, -1))
        b = eng.trace(sp, sd, [1, 1, 1], 100000, options=_opts(capi.KERNEL_SLAB, 1))
        _assert_same(a, b, ctx="direct vs slab")
        # permutation invariance: a shuffled ray order gives the same per-ray results
        perm = np.random.RandomState(0).permutation(R * R)
        c = eng.trace(sp.reshape(-1, 3)[perm].reshape(-1), sd.reshape(-1, 3)[perm].reshape(-1), [1, 1, 1], 100000,
                      options=_opts(capi.KERNEL_SLAB, 0))
        assert np.array_equal(c["end_position"].reshape(-1, 3), b["end_position"].reshape(-1, 3)[perm])
        assert np.array_equal(c["end_iteration"], b["end_iteration"][perm])
    field = oracle.port.interleave(dx, dy, dz, tc)
    idx = np.arange(0, R * R, 97)
    want = oracle.port.trace(field, bounds, sp.reshape(-1, 3)[idx].reshape(-1), sd.reshape(-1, 3)[idx].reshape(-1),
                             [1, 1, 1], 100000)
    assert np.array_equal(want["end_position"].reshape(-1, 3), b["end_position"].reshape(-1, 3)[idx])
    assert np.array_equal(want["end_direction"].reshape(-1, 3).view(np.uint32),
                          b["end_direction"].reshape(-1, 3)[idx].view(np.uint32))
    assert np.array_equal(want["end_iteration"], b["end_iteration"][idx])
    # all rays of this transparent scene leave the field: physical invariant |dir| == ior at exit (ior -> 1)
    steps = b["end_iteration"].astype(np.int64)
    assert steps.min() > 100 and steps.max() < 100000


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("fdt", [np.float32, np.int16])
def test_torture_rare_paths(fdt, kernel):
    """Rare arithmetic paths: collapsed / tiny / huge directions (division by ~0, steps beyond 2^30 ->
    the x86 cvttss2si INT_MIN rule), large gradients, anisotropic invscale, mixed opaque voxels."""
    rng = np.random.RandomState(123)
    B = [18, 22, 20]
    n = B[0] * B[1] * B[2]
    if fdt == np.float32:
        d = [(rng.randn(n) * 3000).astype(np.float32) for _ in range(3)]
    else:
        d = [rng.randint(-20000, 20000, n).astype(np.int16) for _ in range(3)]
    tc = np.where(rng.rand(n) < 0.97, 0xFFFFFFFF, rng.randint(0, 2 ** 31, n)).astype(np.uint32)
    field = oracle.port.interleave(d[0], d[1], d[2], tc)
    m = 4096
    pv = rng.rand(m, 3) * (np.array(B) - 1.001)
    sp = np.round(pv * 65536).astype(np.uint32).reshape(-1)
    dirs = rng.randn(m, 3).astype(np.float32)
    scale = np.float32(10.0) ** rng.uniform(-9, 3, m).astype(np.float32)       # 1e-9 .. 1e3
    dirs *= scale[:, None]
    dirs[::97] = 0.0                                                           # exactly collapsed directions
    dirs[5::101, 1:] = 0.0
    dirs[7::113, 0] = np.inf                                                   # one infinite component: step = inf * 0 = NaN there only
    dirs[11::127, 2] = -np.inf
    dirs[13::131, 1] = np.nan                                                  # NaN direction component
    dirs[17::137] = 3.0e38                                                     # |dir|^2 overflows
    sd = dirs.reshape(-1)
    with capi.Engine.from_host(B, d, tc) as eng:
        for inv in ([1, 1, 1], [7.5, 0.25, 3.0], [1e-3, 1e3, 1.0]):
            want = oracle.port.trace(field, B, sp, sd, inv, 3000)
            for sort in (-1, 1):
                got = eng.trace(sp, sd, inv, 3000, options=_opts(kernel, sort))
                _assert_same(got, want, ctx="inv=%s sort=%d" % (inv, sort))
        # int16 directions including zeros and extremes
        sdi = rng.randint(-32768, 32768, 3 * m).astype(np.int16)
        sdi[::50] = 0
        want = oracle.port.trace(field, B, sp, sdi, [1, 2, 0.5], 3000)
        _assert_same(eng.trace(sp, sdi, [1, 2, 0.5], 3000, options=_opts(kernel)), want, ctx="i16 dirs")


@pytest.mark.gpu
def test_short_division_is_exact_for_every_admitted_float():
    """march_step replaces the reference's IEEE divide 0x42000000p0f / dot (.cu:346) by rcp + one fma
    refinement inside an exponent window; the device sweeps every float bit pattern (both signs, all
    exponents, all 2^23 mantissas) and compares with the IEEE quotient.  Integer-exact: 0 mismatches."""
    bad, admitted = capi.selftest_ilen2(0)
    assert bad == 0
    assert admitted == (251 - 32) * (1 << 23)          # positive d with biased exponent 32..250, nothing else


@pytest.mark.gpu
def test_position_update_is_exact_for_every_step_below_2_31():
    """Inside the step window the march adds roundf(s) to the position through v_cvt_rpi_i32_f32(|s|) with the sign folded
    into the add (vrt_march.h: add_round): the device sweeps every float with |s| < 2^31 against std::round + cvttss2si
    (.cu:347).  Integer-exact: 0 mismatches."""
    bad, checked = capi.selftest_round(0)
    assert bad == 0
    assert checked == 2 * 158 * (1 << 23)


@pytest.mark.gpu
def test_staging_division_by_multiplication_is_exact():
    """Staging splits a brick's voxel number v < 1024 into (x, y, z) by (v * m) >> 22 with m from a float reciprocal
    (vrt_slab.h: stage_magic): the device checks quotient and remainder for every divisor 2..1024 and every dividend."""
    bad, checked = capi.selftest_stage_division(0)
    assert bad == 0
    assert checked == 1023 * 1024


@pytest.mark.gpu
@pytest.mark.parametrize("fdt", [np.float32, np.int16])
@pytest.mark.parametrize("ddt", [np.float32, np.int16])
def test_persistent_schedule(fdt, ddt):
    """schedule = 2: a grid of resident wavefronts, each drawing groups of 64 rays from a counter until it runs
    out (the auto mode for launches larger than the GPU holds).  Results must not depend on which wavefront
    marches which group: coherent + incoherent rays, ragged last group, NaN / inf directions, all sort modes,
    more groups than resident wavefronts cannot be forced at this size, so the counter path is what is tested."""
    N = 48
    bounds, dx, dy, dz, tc = W.seam_field(N, fdt, opaque_radius=5.0)
    field = oracle.port.interleave(dx, dy, dz, tc)
    sp1, sd1 = W.perspective_rays(N, 75, dir_dtype=ddt)
    sp2, sd2 = W.random_rays(N, 3001, seed=5, vol_seed=1234)
    if ddt == np.int16:
        sd2 = np.clip(np.round(sd2 * 256.0), -32768, 32767).astype(np.int16)
    sp, sd = np.concatenate([sp1, sp2]), np.concatenate([sd1, sd2])
    if ddt == np.float32:
        sd.reshape(-1, 3)[5::173, 1] = np.nan
        sd.reshape(-1, 3)[7::191, 0] = np.inf
        sd.reshape(-1, 3)[11::197] = 0.0
    for inv, cap in (([1, 1, 1], 100000), ([0.5, 1, 2], 300)):
        want = oracle.port.trace(field, bounds, sp, sd, inv, cap)
        with capi.Engine.from_host(bounds, [dx, dy, dz], tc) as eng:
            for sort in (-1, 1, 3):
                got = eng.trace(sp, sd, inv, cap, options=_opts(capi.KERNEL_SLAB, sort, schedule=2))
                _assert_same(got, want, ctx="inv=%s cap=%d sort=%d" % (inv, cap, sort))


@pytest.mark.gpu
@pytest.mark.parametrize("fdt,ddt", [(np.float32, np.float32), (np.int16, np.int16)])
@pytest.mark.parametrize("slice_steps", [16, 100, 1024])
def test_time_sliced_schedule(fdt, ddt, slice_steps, monkeypatch):
    """schedule = 3: persistent wavefronts draw (slice, group) tickets; a group marches `slice_steps` wave-steps, parks
    its ray state in global memory (write-through stores, flag, vrt_slab.h) and is resumed by whichever wavefront
    draws its next ticket.  Slices far shorter than the rays' lives (16 steps: a few hundred hand-offs per group) and
    the production length must both reproduce the oracle bit for bit, with rays stopped by opacity, by the field
    boundary and by a tight iteration cap, NaN / inf / zero directions and a launch order by caller / tile / longest
    first."""
    monkeypatch.setenv("VRT_SLICE_STEPS", str(slice_steps))
    N = 40
    bounds, dx, dy, dz, tc = W.seam_field(N, fdt, opaque_radius=5.0)
    field = oracle.port.interleave(dx, dy, dz, tc)
    sp1, sd1 = W.perspective_rays(N, 75, dir_dtype=ddt)
    sp2, sd2 = W.random_rays(N, 3001, seed=5, vol_seed=1234)
    if ddt == np.int16:
        sd2 = np.clip(np.round(sd2 * 256.0), -32768, 32767).astype(np.int16)
    sp, sd = np.concatenate([sp1, sp2]), np.concatenate([sd1, sd2])
    if ddt == np.float32:
        sd.reshape(-1, 3)[5::173, 1] = np.nan
        sd.reshape(-1, 3)[7::191, 0] = np.inf
        sd.reshape(-1, 3)[11::197] = 0.0
    for inv, cap in (([1, 1, 1], 100000), ([0.5, 1, 2], 300)):
        want = oracle.port.trace(field, bounds, sp, sd, inv, cap)
        with capi.Engine.from_host(bounds, [dx, dy, dz], tc) as eng:
            for sort in (-1, 1, 3):
                got = eng.trace(sp, sd, inv, cap, options=_opts(capi.KERNEL_SLAB, sort, schedule=3))
                _assert_same(got, want, ctx="slice=%d inv=%s cap=%d sort=%d" % (slice_steps, inv, cap, sort))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 63, 64, 65, 129, 257])
def test_tiny_launches_with_forced_order_and_schedule(n):
    """Ray counts around the wavefront size with every launch-order / schedule feature forced on: one group
    (the persistent schedule then falls back to the static launch), a ragged second group, a few groups."""
    N = 24
    bounds, dx, dy, dz, tc = W.seam_field(N, np.float32, opaque_radius=3.0)
    field = oracle.port.interleave(dx, dy, dz, tc)
    sp, sd = W.random_rays(N, n, seed=n, vol_seed=1234)
    want = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], 5000)
    with capi.Engine.from_host(bounds, [dx, dy, dz], tc) as eng:
        for sort in (-1, 1, 2, 3):
            for schedule in (1, 2, 3):
                got = eng.trace(sp, sd, [1, 1, 1], 5000, options=_opts(capi.KERNEL_SLAB, sort, schedule=schedule))
                _assert_same(got, want, ctx="n=%d sort=%d schedule=%d" % (n, sort, schedule))


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", KERNELS)
def test_step_window_boundary(kernel):
    """The step's fast path (short division + integer rounding) is taken iff |dir|^2 lies in a window whose lower
    edge is (2C max|inv| / 2^31)^2 rounded up (vrt_march.h, StepWindow): inside it every doubled step component
    is below 2^31 by construction (v_cvt_i32_f32 would saturate beyond), outside the general rounding runs.  Rays
    whose |dir| straddles that edge within a few ulps, aligned with the axis of the largest invscale (worst
    case) and in random directions, for several invscales -- any mis-set edge shows as a position mismatch."""
    rng = np.random.RandomState(77)
    B = [14, 15, 16]
    n = B[0] * B[1] * B[2]
    d = [(rng.randn(n) * 1e-4).astype(np.float32) for _ in range(3)]       # tiny gradients: |dir| stays where it starts
    tc = np.full(n, 0xFFFFFFFF, np.uint32)
    field = oracle.port.interleave(d[0], d[1], d[2], tc)
    m = 6000
    pv = 1.0 + rng.rand(m, 3) * (np.array(B) - 3.0)
    sp = np.round(pv * 65536).astype(np.uint32).reshape(-1)
    with capi.Engine.from_host(B, d, tc) as eng:
        for inv in ([1, 1, 1], [3.0, 0.5, 1.0], [0.25, 0.25, 1000.0], [1e-3, 2e-3, 1e-3]):
            edge = 2.0 * 1107296256.0 * max(inv) / 2147483648.0 / 65536.0   # |dir| (caller scale) at the window edge
            axis = int(np.argmax(inv))
            u = rng.randn(m, 3)
            u /= np.linalg.norm(u, axis=1)[:, None]
            u[: m // 2] = 0.0
            u[: m // 2, axis] = np.where(rng.rand(m // 2) < 0.5, 1.0, -1.0)      # worst case: along the max-inv axis
            mag = edge * (1.0 + rng.uniform(-3e-5, 3e-5, m))                      # within +-3e-5 of the edge (margin is 2^-17)
            mag[::7] = edge * (1.0 + rng.uniform(-1e-7, 1e-7, mag[::7].size))     # and within an ulp or two
            sd = (u * mag[:, None]).astype(np.float32).reshape(-1)
            want = oracle.port.trace(field, B, sp, sd, inv, 6)
            got = eng.trace(sp, sd, inv, 6, options=_opts(kernel))
            _assert_same(got, want, ctx="inv=%s" % (inv,))


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", KERNELS)
def test_quotient_at_the_edge_of_overflow(kernel):
    """|dir|^2 so small that 0x42000000p0f / |dir|^2 lies within a factor 4 of FLT_MAX (internal |dir| around 2^-48, i.e.
    caller-scale 2^-64): the step (dir * inv) * ilen must be formed exactly as the reference writes it (.cu:346-347) --
    a component that is zero or denormal gives 0 or a small step while the others saturate to INT_MIN (cvttss2si), and
    any re-association (such as doubling ilen first, which overflows one binade earlier) turns that component into
    inf * 0 = NaN -> INT_MIN as well."""
    rng = np.random.RandomState(31)
    B = [12, 13, 14]
    n = B[0] * B[1] * B[2]
    d = [(rng.randn(n) * 1e-30).astype(np.float32) for _ in range(3)]
    tc = np.full(n, 0xFFFFFFFF, np.uint32)
    field = oracle.port.interleave(d[0], d[1], d[2], tc)
    m = 8192
    pv = 1.0 + rng.rand(m, 3) * (np.array(B) - 3.0)
    sp = np.round(pv * 65536).astype(np.uint32).reshape(-1)
    u = rng.randn(m, 3)
    u /= np.linalg.norm(u, axis=1)[:, None]
    mag = 2.0 ** rng.uniform(-66.5, -62.0, m)                 # the quotient overflows below about 2^-64.5
    dirs = (u * mag[:, None]).astype(np.float32)
    dirs[::3, 0] = 0.0                                        # exactly zero components
    dirs[1::5, 1] = (rng.randn(dirs[1::5].shape[0]) * 1e-42).astype(np.float32)   # denormal components
    dirs[2::7, 2] *= np.float32(1e-12)                        # tiny but normal
    sd = dirs.reshape(-1)
    with capi.Engine.from_host(B, d, tc) as eng:
        for inv in ([1, 1, 1], [2.0, 0.5, 1.0]):
            want = oracle.port.trace(field, B, sp, sd, inv, 5)
            got = eng.trace(sp, sd, inv, 5, options=_opts(kernel))
            _assert_same(got, want, ctx="inv=%s" % (inv,))


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", KERNELS)
def test_denormal_scale_gradients_and_invscale(kernel):
    """Gradient values and invscales so small that the interpolation's 2^-48 scaling produces denormals, with
    directions tiny enough for those bits to reach the output: the scaling must stay a separate, correctly
    rounded multiplication (folding it into invscale was tried; it is exact only away from denormals, needed a
    field scan plus a per-step uniform branch, and measured 1-3 % slower, so it was dropped -- this test stays)."""
    rng = np.random.RandomState(9)
    B = [12, 13, 14]
    n = B[0] * B[1] * B[2]
    tc = np.full(n, 0xFFFFFFFF, np.uint32)
    m = 3000
    pv = 1.0 + rng.rand(m, 3) * (np.array(B) - 3.0)
    sp = np.round(pv * 65536).astype(np.uint32).reshape(-1)
    cases = []
    # (gradient scale, direction scale, invscale)
    cases.append((1e-30, 1e-36, [1.0, 1.0, 1.0]))          # field values ~1e-30: denormal after scaling
    cases.append((1e-41, 1e-40, [1.0, 2.0, 0.5]))          # denormal field values
    cases.append((1.0, 1e-3, [1e-25, 1e-25, 1e-25]))       # invscale < 2^-78
    cases.append((1e-12, 1e-9, [1.0, 1.0, 1.0]))           # small but normal throughout
    # the other end: the reference multiplies by weights of up to 2^48 in total BEFORE its final 2^-48, so values beyond
    # 2^128 / 2^48 overflow to infinity inside the interpolation; LDS bricks carry the 2^-48 up front and must not be
    # used for such voxels (prescale_ok, vrt_slab.h) -- well above, around and just below that edge
    cases.append((3e27, 1.0, [1.0, 1.0, 1.0]))
    cases.append((4e23, 1.0, [1.0, 1.0, 1.0]))
    cases.append((1e23, 1.0, [1.0, 0.5, 2.0]))
    for gs, ds, inv in cases:
        d = [(rng.randn(n) * gs).astype(np.float32) for _ in range(3)]
        field = oracle.port.interleave(d[0], d[1], d[2], tc)
        sd = (rng.randn(m, 3) * ds).astype(np.float32).reshape(-1)
        want = oracle.port.trace(field, B, sp, sd, inv, 40)
        with capi.Engine.from_host(B, d, tc) as eng:
            got = eng.trace(sp, sd, inv, 40, options=_opts(kernel))
        _assert_same(got, want, ctx="gs=%g ds=%g inv=%s" % (gs, ds, inv))
