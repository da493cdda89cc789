"""GPU (-m gpu): BASELINE config 3 at full size (512^3 fp32 field, 1024x1024 perspective rays, the bench
workload) through size-independent properties: the two kernels agree bit-for-bit on every ray under
different ray orders, device-pointer and host-pointer entry points agree, and an evenly strided
sample of rays matches the oracle bit-for-bit."""
import numpy as np
import pytest

import oracle
from volumeraytracer_amd import capi, distributed as D

pytestmark = pytest.mark.gpu


def test_config3_full_size_properties():
    torch = pytest.importorskip("torch")
    import bench
    cfg = bench.CONFIGS[3]
    dev = torch.device("cuda", 0)
    bounds, d, tr = bench.synth_field_device(cfg["N"], cfg["field"], dev)
    torch.cuda.synchronize()
    eng = capi.Engine.from_device(bounds, [x.data_ptr() for x in d], tr.data_ptr(), capi.VRT_F32, d[0].numel())
    sp, sd = bench.make_rays(cfg, 0, 1)
    n = sp.size // 3
    d_sp = torch.from_numpy(sp.view(np.int32)).to(dev)
    d_sd = torch.from_numpy(sd).to(dev)
    stream = torch.cuda.current_stream().cuda_stream

    def run(kernel, sort):
        rec = D.alloc_record(n, dev)
        ep, ed, ei, rl = D.record_views(rec, n)
        eng.trace_device(capi.VRT_F32, n, d_sp.data_ptr(), d_sd.data_ptr(), ep.data_ptr(), ed.data_ptr(), ei.data_ptr(),
                         rl.data_ptr(), [1, 1, 1], 100000, stream=stream,
                         options=capi.default_options(kernel=kernel, sort_rays=sort))
        torch.cuda.synchronize()
        return rec.cpu().numpy()

    a = run(capi.KERNEL_SLAB, 1)
    assert np.array_equal(a, run(capi.KERNEL_DIRECT, -1)), "slab (tile order) vs direct (caller order)"
    assert np.array_equal(a, run(capi.KERNEL_SLAB, 2)), "longest-first order"
    assert np.array_equal(a, run(capi.KERNEL_SLAB, -1)), "slab in caller order"
    ep, ed, ei, rl = a[:3 * n], a[3 * n:6 * n], a[6 * n:7 * n], a[7 * n:]
    assert np.all(rl.view(np.uint32) == 0xFFFFFFFF)
    steps = ei.view(np.uint32).astype(np.int64)
    assert steps.min() > 500 and steps.max() < 100000          # transparent scene: every ray leaves the field
    B = np.array(bounds, np.int64)
    cell = ep.view(np.uint32).reshape(-1, 3).astype(np.int64) >> 16
    assert np.all(np.any((cell >= B - 1) | (cell > 60000), axis=1)), "every ray ended outside the field"

    # host-pointer entry point on a band of rays == device-pointer results
    band = slice(0, 3 * 4096)
    h = eng.trace(sp[band], sd[band], [1, 1, 1], 100000)
    assert np.array_equal(h["end_position"], ep.view(np.uint32)[band])
    assert np.array_equal(h["end_iteration"], ei.view(np.uint32)[:4096])

    # oracle on an evenly strided sample
    host = [x.cpu().numpy() for x in d]
    field = oracle.port.interleave(host[0], host[1], host[2], tr.cpu().numpy().view(np.uint32))
    del host
    idx = np.linspace(0, n - 1, 3000).astype(np.int64)
    want = oracle.port.trace(field, bounds, sp.reshape(-1, 3)[idx].reshape(-1), sd.reshape(-1, 3)[idx].reshape(-1),
                             [1, 1, 1], 100000)
    assert np.array_equal(want["end_position"].reshape(-1, 3), ep.view(np.uint32).reshape(-1, 3)[idx])
    assert np.array_equal(want["end_direction"].view(np.uint32).reshape(-1, 3), ed.view(np.uint32).reshape(-1, 3)[idx])
    assert np.array_equal(want["end_iteration"], ei.view(np.uint32)[idx])
    eng.close()


def test_config4_tiled():
    """BASELINE config 4 as written: 512^3 fp32 field, 2048 x 2048 perspective rays cut into 8 contiguous row bands
    (the tiles the 8 GPUs of a node would take, distributed.row_band).  On the one GPU here every band is marched into
    its own output record, the records are reassembled exactly as rank 0 does after the gather
    (distributed.assemble), and the result must equal the single launch over all 4 194 304 rays bit for bit.
    An evenly strided sample is checked against the oracle and every ray must have left the field."""
    torch = pytest.importorskip("torch")
    import bench
    cfg = bench.CONFIGS[4]
    world = 8
    dev = torch.device("cuda", 0)
    bounds, d, tr = bench.synth_field_device(cfg["N"], cfg["field"], dev)
    torch.cuda.synchronize()
    eng = capi.Engine.from_device(bounds, [x.data_ptr() for x in d], tr.data_ptr(), capi.VRT_F32, d[0].numel())
    stream = torch.cuda.current_stream().cuda_stream

    def march(sp, sd):
        n = sp.size // 3
        d_sp = torch.from_numpy(sp.view(np.int32)).to(dev)
        d_sd = torch.from_numpy(sd).to(dev)
        rec = D.alloc_record(n, dev)
        ep, ed, ei, rl = D.record_views(rec, n)
        eng.trace_device(capi.VRT_F32, n, d_sp.data_ptr(), d_sd.data_ptr(), ep.data_ptr(), ed.data_ptr(), ei.data_ptr(),
                         rl.data_ptr(), [1, 1, 1], 100000, stream=stream)
        torch.cuda.synchronize()
        return rec

    assert bench.lattice_rows(cfg, world, "strong") == 2048            # the fixed lattice, not 8 x 2048 rows
    sp_all, sd_all = bench.make_rays(cfg, 0, 1, scaling="strong")
    n_all = sp_all.size // 3
    assert n_all == 2048 * 2048
    whole = D.assemble([march(sp_all, sd_all)], [n_all])

    records, counts, first = [], [], 0
    for rank in range(world):
        sp, sd = bench.make_rays(cfg, rank, world, scaling="strong")
        n = sp.size // 3
        assert np.array_equal(sp, sp_all[3 * first:3 * (first + n)]) and np.array_equal(sd, sd_all[3 * first:3 * (first + n)])
        records.append(march(sp, sd))
        counts.append(n)
        first += n
    assert first == n_all and len(set(counts)) == 1
    tiled = D.assemble(records, counts)
    for k in ("end_position", "end_direction_bits", "end_iteration", "remaining_light"):
        assert np.array_equal(tiled[k], whole[k]), "8 row bands vs one launch: %s" % k

    assert np.all(whole["remaining_light"] == 0xFFFFFFFF)
    steps = whole["end_iteration"].astype(np.int64)
    assert steps.min() > 500 and steps.max() < 100000
    B = np.array(bounds, np.int64)
    cell = whole["end_position"].reshape(-1, 3).astype(np.int64) >> 16
    assert np.all(np.any((cell >= B - 1) | (cell > 60000), axis=1)), "every ray ended outside the field"

    host = [x.cpu().numpy() for x in d]
    field = oracle.port.interleave(host[0], host[1], host[2], tr.cpu().numpy().view(np.uint32))
    del host
    idx = np.linspace(0, n_all - 1, 3000).astype(np.int64)
    want = oracle.port.trace(field, bounds, sp_all.reshape(-1, 3)[idx].reshape(-1), sd_all.reshape(-1, 3)[idx].reshape(-1),
                             [1, 1, 1], 100000)
    assert np.array_equal(want["end_position"].reshape(-1, 3), whole["end_position"].reshape(-1, 3)[idx])
    assert np.array_equal(want["end_direction"].view(np.uint32).reshape(-1, 3), whole["end_direction_bits"].reshape(-1, 3)[idx])
    assert np.array_equal(want["end_iteration"], whole["end_iteration"][idx])
    eng.close()


def test_config5_full_size():
    """BASELINE config 5 as written and at its real size: 1024^3 uint8 density + 256-entry transfer function -> scene
    build on the device (8.5 GB int16 field) -> 1024 x 1024 orthographic rays, opaque cores (the early-out stress case).
      * the default launch (auto schedule: the engine's tuner runs the whole-group and the time-sliced schedule in turn
        and settles) gives the same bits on every call, equal to the direct kernel in caller order, to the slab kernel in
        caller order and to the explicit schedules;
      * roughly half of the rays stop on the opacity test (w > 0) inside the field, before the cap; the others leave it;
      * two blocks of 40 x 40 rays (through an opaque core / near the rim) are checked bit for bit against the oracle.  The
        oracle's own scene build of the whole 1024^3 volume would take minutes on the host, so it builds only the
        sub-volume those rays visit: their paths are traced on the GPU (path output), the bounding box of every
        position of every sampled ray plus the stencil's and the interpolation's reach is cut out of the density, and
        the oracle builds and marches that (stencil, field and march are translation invariant in whole voxels)."""
    torch = pytest.importorskip("torch")
    import bench
    cfg = bench.CONFIGS[5]
    N, R = cfg["N"], cfg["R"]
    dev = torch.device("cuda", 0)
    eng, scene, bounds, _, _, (dens, lut_ior, lut_tr) = bench.build_engine(cfg, dev)
    assert eng.field_bytes == 8 * (N - 2) ** 3
    sp, sd = bench.make_rays(cfg, 0, 1)
    n = sp.size // 3
    assert n == R * R
    d_sp = torch.from_numpy(sp.view(np.int32)).to(dev)
    d_sd = torch.from_numpy(sd).to(dev)
    stream = torch.cuda.current_stream().cuda_stream

    def run(**kw):
        rec = D.alloc_record(n, dev)
        ep, ed, ei, rl = D.record_views(rec, n)
        eng.trace_device(capi.VRT_F32, n, d_sp.data_ptr(), d_sd.data_ptr(), ep.data_ptr(), ed.data_ptr(), ei.data_ptr(),
                         rl.data_ptr(), [1, 1, 1], 100000, stream=stream, options=capi.default_options(**kw))
        eng.status()
        return rec.cpu().numpy(), eng.last_schedule()

    a, _ = run()
    seen = set()
    for call in range(6):                                          # trials 2, 3, 2, 3, then the settled choice
        b, sched = run()
        seen.add(sched)
        assert np.array_equal(a, b), "auto schedule, call %d (schedule %d)" % (call, sched)
    assert seen >= {2, 3}, seen                                    # both persistent schedules really ran
    assert np.array_equal(a, run(kernel=capi.KERNEL_DIRECT, sort_rays=-1)[0]), "slab (tile order) vs direct (caller order)"
    assert np.array_equal(a, run(kernel=capi.KERNEL_SLAB, sort_rays=-1, schedule=1)[0]), "slab in caller order, static schedule"
    assert np.array_equal(a, run(kernel=capi.KERNEL_SLAB, schedule=3)[0]), "time-sliced schedule"

    ep, ed, ei, rl = a[:3 * n].view(np.uint32), a[3 * n:6 * n].view(np.uint32), a[6 * n:7 * n].view(np.uint32), a[7 * n:].view(np.uint32)
    assert np.all(rl == 0xFFFFFFFF)
    B = np.array(bounds, np.int64)
    cell = ep.reshape(-1, 3).astype(np.int64) >> 16
    inside = np.all(cell < B - 1, axis=1)                          # stopped inside the field: only the w > 0 test does that
    assert ei.max() < 100000, "no ray ran into the cap"
    frac = float(inside.mean())
    assert 0.25 < frac < 0.75, "share of rays stopped by the opacity test: %.3f" % frac

    # ---- oracle on two blocks of rays, through the sub-volume they visit
    iters_p = 8192
    assert int(ei.max()) < iters_p
    dens3 = dens.reshape(N, N, N)
    rows = np.arange(R)
    hit_rows = inside.reshape(R, R)
    r_hit, c_hit = np.argwhere(hit_rows)[len(np.argwhere(hit_rows)) // 2]       # a ray that hits a core: block around it
    blocks = [(int(np.clip(r_hit - 20, 0, R - 40)), int(np.clip(c_hit - 20, 0, R - 40))), (60, 930)]
    checked = 0
    for (r0, c0) in blocks:
        idx = (rows[r0:r0 + 40, None] * R + rows[None, c0:c0 + 40]).reshape(-1)
        m = idx.size
        bsp, bsd = sp.reshape(-1, 3)[idx].reshape(-1).copy(), sd.reshape(-1, 3)[idx].reshape(-1).copy()
        t_sp = torch.from_numpy(bsp.view(np.int32)).to(dev)
        t_sd = torch.from_numpy(bsd).to(dev)
        rec = D.alloc_record(m, dev)
        pep, ped, pei, prl = D.record_views(rec, m)
        path = torch.zeros(m * iters_p * 3, dtype=torch.int32, device=dev)
        eng.trace_device(capi.VRT_F32, m, t_sp.data_ptr(), t_sd.data_ptr(), pep.data_ptr(), ped.data_ptr(), pei.data_ptr(),
                         prl.data_ptr(), [1, 1, 1], iters_p, d_path=path.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        assert np.array_equal(pei.cpu().numpy().view(np.uint32), ei[idx]), "the path run marches the same steps"
        P = path.cpu().numpy().view(np.uint32).astype(np.int64).reshape(m * iters_p, 3)
        del path
        P[P >= 2 ** 31] -= 2 ** 32                                 # positions that wrapped below zero
        c = P >> 16
        cmin, cmax = c.min(axis=0), c.max(axis=0)
        del P, c
        lo = np.maximum(cmin - 1, 0)
        hi_field = np.minimum(cmax + 2, B - 1)                     # last field voxel the sub-field must hold
        sub = np.ascontiguousarray(dens3[lo[0]:hi_field[0] + 3, lo[1]:hi_field[1] + 3, lo[2]:hi_field[2] + 3])
        sdims = [int(x) for x in sub.shape]
        assert sub.size < 200e6, sdims                             # the whole point: far smaller than 1024^3
        hx, hy, hz, tc, ob = oracle.port.scene_build(sdims, lut_ior[sub.reshape(-1)], lut_tr[sub.reshape(-1)])
        field = oracle.port.interleave(hx, hy, hz, tc)
        del hx, hy, hz
        shift = (lo.astype(np.uint32) << 16)
        ssp = (bsp.reshape(-1, 3) - shift[None, :]).reshape(-1)
        want = oracle.port.trace(field, ob, ssp, bsd, [1, 1, 1], 100000)
        assert np.array_equal((want["end_position"].reshape(-1, 3) + shift[None, :]).astype(np.uint32), ep.reshape(-1, 3)[idx]), (r0, c0)
        assert np.array_equal(want["end_direction"].view(np.uint32).reshape(-1, 3), ed.reshape(-1, 3)[idx]), (r0, c0)
        assert np.array_equal(want["end_iteration"], ei[idx]), (r0, c0)
        checked += m
    assert checked >= 3000
    scene.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fdt", [np.float32, np.int16])
def test_wide_field_stages_through_64_bit_addresses(fdt):
    """The slab kernel stages a brick through 32-bit byte offsets beside a scalar base whenever the widest brick of the
    field spans less than 4 GiB (vrt_slab.h: off32) -- true for every BASELINE config.  A field whose x-slices are large
    enough takes the other branch (64-bit address per voxel): 6 x 2900 x 2900 int16 / 6 x 1500 x 1400 fp32, random rays in
    all directions against the oracle, both kernels."""
    B = [6, 2900, 2900] if fdt == np.int16 else [6, 1500, 1400]
    assert 160 * B[1] * B[2] * (8 if fdt == np.int16 else 16) >= 1 << 32
    n = B[0] * B[1] * B[2]
    rng = np.random.RandomState(5)
    # smooth-ish field: a few plane waves, so that rays bend but bricks stay compact
    y = np.arange(B[1], dtype=np.float32)[:, None]
    z = np.arange(B[2], dtype=np.float32)[None, :]
    d = []
    for k in range(3):
        plane = (np.sin(y * (0.011 + 0.003 * k)) * np.cos(z * (0.013 - 0.002 * k))).astype(np.float32)
        vol = np.stack([plane * (1.0 + 0.1 * x) for x in range(B[0])]) * (40.0 if fdt == np.float32 else 60.0)
        d.append((vol.astype(np.float32) if fdt == np.float32 else np.round(vol).astype(np.int16)).reshape(-1))
    tc = np.full(n, 0xFFFFFFFF, np.uint32)
    tc[rng.randint(0, n, 2000)] = 7                                  # a few opaque-ish voxels (w > 0 bricks)
    m = 20000
    pv = np.stack([rng.uniform(0.5, B[0] - 1.5, m), rng.uniform(1, B[1] - 2, m), rng.uniform(1, B[2] - 2, m)], axis=1)
    sp = np.round(pv * 65536).astype(np.uint32).reshape(-1)
    u = rng.randn(m, 3)
    u[:, 0] *= 0.02                                                  # mostly along y / z: long paths inside the thin field
    u /= np.linalg.norm(u, axis=1)[:, None]
    sd = u.astype(np.float32).reshape(-1)
    field = oracle.port.interleave(d[0], d[1], d[2], tc)
    want = oracle.port.trace(field, B, sp, sd, [1, 1, 1], 4000)
    assert want["end_iteration"].astype(np.int64).mean() > 200      # the rays really march
    with capi.Engine.from_host(B, d, tc) as eng:
        for kernel in (capi.KERNEL_SLAB, capi.KERNEL_DIRECT):
            got = eng.trace(sp, sd, [1, 1, 1], 4000, options=capi.Options(kernel=kernel))
            for k in ("end_position", "end_iteration", "remaining_light"):
                assert np.array_equal(got[k], want[k]), (k, kernel)
            assert np.array_equal(got["end_direction"].view(np.uint32), want["end_direction"].view(np.uint32)), kernel
