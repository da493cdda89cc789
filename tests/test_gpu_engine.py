"""GPU (-m gpu): the engine object behind the C ABI beyond a single plain call -- the things the reference's
dispatcher does around the march (cuda_volume_raytracer.cu:676-686 replicate, :804-912 chunk fan-out over devices)
and what the MI355X engine adds (streams, chunk pipeline, device-pointer scene entry).  Every variant must give the
bits of the plain single call, which test_gpu_parity.py pins to the oracle and the reference's golden vectors; the
plain call is pinned to the oracle here once more so that this file stands on its own."""
import numpy as np
import pytest

import oracle
from volumeraytracer_amd import capi, workloads as W

pytestmark = pytest.mark.gpu
KEYS = ("end_position", "end_direction", "end_iteration", "remaining_light")


def _same(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def _scene(N=56, fdt=np.float32, ddt=np.float32, nrays=20000, opaque=4.0):
    bounds, dx, dy, dz, tc = W.seam_field(N, fdt, seed=31, opaque_radius=opaque)
    sp, sd = W.random_rays(N, nrays, ddt, seed=99, vol_seed=31)
    field = oracle.port.interleave(dx, dy, dz, tc)
    want = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], 100000)
    return bounds, (dx, dy, dz), tc, sp, sd, want


def test_two_streams_one_engine():
    """Two device-pointer calls in flight on two streams of ONE handle (different ray sets, so a shared permutation or
    group counter would show): the handle's workspaces are per handle, the second call must wait for the first on the
    device (vrt.h, vrt_trace_device) -- both results equal the oracle."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    bounds, d, tc, sp, sd, want = _scene(nrays=40000)
    half = (sp.size // 3 // 2) * 3
    sets = [(sp[:half], sd[:half]), (sp[half:], sd[half:])]
    with capi.Engine.from_host(bounds, list(d), tc) as eng:
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs = []
        for rep in range(3):                                   # back to back, alternating streams, nothing synchronised between
            for (p, q), st in zip(sets, streams):
                n = p.size // 3
                t_sp = torch.from_numpy(p.view(np.int32)).to(dev)
                t_sd = torch.from_numpy(q).to(dev)
                o = [torch.empty(3 * n, dtype=torch.int32, device=dev), torch.empty(3 * n, dtype=torch.float32, device=dev),
                     torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)]
                torch.cuda.synchronize()
                outs.append((o, t_sp, t_sd, n))
        k = 0
        for rep in range(3):
            for st in streams:
                o, t_sp, t_sd, n = outs[k]
                k += 1
                eng.trace_device(capi.VRT_F32, n, t_sp.data_ptr(), t_sd.data_ptr(), o[0].data_ptr(), o[1].data_ptr(),
                                 o[2].data_ptr(), o[3].data_ptr(), [1, 1, 1], 100000, stream=st.cuda_stream,
                                 options=capi.default_options(kernel=capi.KERNEL_SLAB, sort_rays=3, schedule=2))
        torch.cuda.synchronize()
        n0 = half // 3
        for k, (o, _, _, n) in enumerate(outs):
            sl = slice(0, n0) if k % 2 == 0 else slice(n0, None)
            assert np.array_equal(o[0].cpu().numpy().view(np.uint32).reshape(-1, 3), want["end_position"].reshape(-1, 3)[sl]), k
            assert np.array_equal(o[1].cpu().numpy().view(np.uint32).reshape(-1, 3),
                                  want["end_direction"].view(np.uint32).reshape(-1, 3)[sl]), k
            assert np.array_equal(o[2].cpu().numpy().view(np.uint32), want["end_iteration"][sl]), k


@pytest.mark.parametrize("fdt,ddt", [(np.float32, np.float32), (np.int16, np.int16)])
def test_host_chunk_pipeline(fdt, ddt):
    """vrt_trace cut into 1, 2, 4 and 7 chunks pipelined over the engine's lanes: same bits, whatever the cut."""
    bounds, d, tc, sp, sd, want = _scene(fdt=fdt, ddt=ddt, nrays=30011)
    with capi.Engine.from_host(bounds, list(d), tc) as eng:
        for lanes in (1, 2, 4, 7):
            got = eng.trace(sp, sd, [1, 1, 1], 100000, options=capi.default_options(host_lanes=lanes))
            for k in KEYS:
                assert _same(got[k], want[k]), (lanes, k)


def test_multi_engine_handle_on_one_device(monkeypatch):
    """The multi-device path of a handle (field replicated, one contiguous band of rays per engine, results written
    straight into the caller's arrays) with all engines on device 0 (VRT_DEBUG_REPLICAS): band cut + merge must
    reproduce the single-engine result bit for bit.  A one-bit device mask is today's single-device handle."""
    bounds, d, tc, sp, sd, want = _scene(nrays=25013)
    monkeypatch.setenv("VRT_DEBUG_REPLICAS", "3")
    with capi.Engine.from_host(bounds, list(d), tc) as eng:
        assert eng.replica_count == 3
        got = eng.trace(sp, sd, [1, 1, 1], 100000)
        for k in KEYS:
            assert _same(got[k], want[k]), k
        few = eng.trace(sp[:3 * 50], sd[:3 * 50], [1, 1, 1], 100000)         # fewer rays than engines x 64: one band
        assert _same(few["end_iteration"], want["end_iteration"][:50])
        assert _same(few["end_position"], want["end_position"][:150]) and _same(few["end_direction"], want["end_direction"][:150])
        pth = eng.trace(sp[:3 * 300], sd[:3 * 300], [1, 1, 1], 37, trace_paths=True)
    monkeypatch.delenv("VRT_DEBUG_REPLICAS")
    with capi.Engine.from_host(bounds, list(d), tc, options=capi.default_options(devices=1)) as eng1:
        assert eng1.replica_count == 1
        got = eng1.trace(sp, sd, [1, 1, 1], 100000)
        for k in KEYS:
            assert _same(got[k], want[k]), k
        pth1 = eng1.trace(sp[:3 * 300], sd[:3 * 300], [1, 1, 1], 37, trace_paths=True)
    for k in KEYS + ("path",):
        assert _same(pth[k], pth1[k]), k
    with capi.Engine.from_host(bounds, list(d), tc, options=capi.default_options(devices=-1)) as eng_all:
        assert eng_all.replica_count == capi.device_count()                   # every visible device (1 on the test box)
        got = eng_all.trace(sp, sd, [1, 1, 1], 100000)
        for k in KEYS:
            assert _same(got[k], want[k]), k


@pytest.mark.parametrize("dti,dtd", [(np.float32, np.float32), (np.uint32, np.int16)])
def test_scene_device_entry(dti, dtd):
    """vrt_scene_trace_device (device buffers, asynchronous) == vrt_scene_trace (host buffers) == the oracle's scene
    layer; the caller's start arrays are left untouched; a ray outside the volume surfaces in status()."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    N = 40
    ior = W.ior_volume(N, seed=9, dtype=dti).reshape(-1)
    rng = np.random.RandomState(5)
    tr = np.full(N ** 3, 0xFFFFFFFF, np.uint32)
    tr[rng.rand(N ** 3) < 0.01] = 7
    n = 5000
    pv = 1.0 + rng.rand(n, 3) * (N - 2.001)
    sp = np.round(pv * 65536).astype(np.uint32).reshape(-1)
    dd = rng.randn(n, 3)
    dd /= np.linalg.norm(dd, axis=1)[:, None]
    sd = dd.astype(np.float32).reshape(-1) if dtd == np.float32 else np.round(dd * 256).astype(np.int16).reshape(-1)
    want = oracle.port.scene_trace([N] * 3, ior, tr, sp, sd, [1, 1, 1], 100000)
    dtype = capi.VRT_F32 if dtd == np.float32 else capi.VRT_I16
    with capi.Scene([N] * 3, ior, tr) as sc:
        host = sc.trace_rays(sp, sd, [1, 1, 1], 100000)
        for k in KEYS:
            assert _same(host[k], want[k]), k
        t_sp = torch.from_numpy(sp.view(np.int32)).to(dev)
        t_sd = torch.from_numpy(sd).to(dev)
        ep = torch.empty(3 * n, dtype=torch.int32, device=dev)
        ed = torch.empty(3 * n, dtype=t_sd.dtype, device=dev)
        ei = torch.empty(n, dtype=torch.int32, device=dev)
        rl = torch.empty(n, dtype=torch.int32, device=dev)
        st = torch.cuda.Stream()
        torch.cuda.synchronize()
        sc.trace_rays_device(dtype, n, t_sp.data_ptr(), t_sd.data_ptr(), ep.data_ptr(), ed.data_ptr(), ei.data_ptr(),
                             rl.data_ptr(), [1, 1, 1], 100000, stream=st.cuda_stream)
        sc.status()
        assert np.array_equal(ep.cpu().numpy().view(np.uint32), want["end_position"])
        assert _same(ed.cpu().numpy(), want["end_direction"])
        assert np.array_equal(ei.cpu().numpy().view(np.uint32), want["end_iteration"])
        assert np.array_equal(rl.cpu().numpy().view(np.uint32), want["remaining_light"])
        assert np.array_equal(t_sp.cpu().numpy().view(np.uint32), sp) and _same(t_sd.cpu().numpy(), sd)
        bad = sp.copy()
        bad[3 * 17] = 0x8000                                        # ray 17 below 1.0 (image_util.cpp:690-695)
        t_bad = torch.from_numpy(bad.view(np.int32)).to(dev)
        sc.trace_rays_device(dtype, n, t_bad.data_ptr(), t_sd.data_ptr(), ep.data_ptr(), ed.data_ptr(), ei.data_ptr(),
                             rl.data_ptr(), [1, 1, 1], 100000, stream=st.cuda_stream)
        with pytest.raises(capi.VrtError) as e:
            sc.status()
        assert e.value.code == -7 and "ray 17 " in str(e.value)
        sc.status()                                                 # reported once


def test_march_clock_is_reported():
    bounds, d, tc, sp, sd, want = _scene(nrays=8192)
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    n = sp.size // 3
    with capi.Engine.from_host(bounds, list(d), tc) as eng:
        t_sp = torch.from_numpy(sp.view(np.int32)).to(dev)
        t_sd = torch.from_numpy(sd).to(dev)
        o = [torch.empty(3 * n, dtype=torch.int32, device=dev), torch.empty(3 * n, dtype=torch.float32, device=dev),
             torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)]
        eng.trace_device(capi.VRT_F32, n, t_sp.data_ptr(), t_sd.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(),
                         o[3].data_ptr(), [1, 1, 1], 100000, options=capi.default_options(kernel=capi.KERNEL_SLAB))
        torch.cuda.synchronize()
        ghz = eng.last_march_clock_ghz()
        assert 0.3 < ghz < 3.0, ghz
        assert np.array_equal(o[2].cpu().numpy().view(np.uint32), want["end_iteration"])


def test_schedule_auto_tuning_keeps_results():
    """vrt_options.schedule = 0 on a launch larger than the GPU holds: the engine alternates the whole-group and the
    time-sliced persistent schedule over its first calls, times them with HIP events and keeps the faster
    (vrt_engine.hip, Workspace::Tuner).  Every call along the way -- trials, undecided, decided -- gives the same bits;
    a change of the ray count starts a new trial series."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    N = 72
    bounds, dx, dy, dz, tc = W.seam_field(N, np.float32, seed=3)
    sp, sd = W.perspective_rays(N, 640)                       # 409 600 rays = 6400 groups > 4096 resident wavefronts
    field = oracle.port.interleave(dx, dy, dz, tc)
    idx = np.linspace(0, sp.size // 3 - 1, 4000).astype(np.int64)
    want = oracle.port.trace(field, bounds, sp.reshape(-1, 3)[idx].reshape(-1), sd.reshape(-1, 3)[idx].reshape(-1), [1, 1, 1], 100000)
    n = sp.size // 3
    with capi.Engine.from_host(bounds, [dx, dy, dz], tc) as eng:
        t_sp = torch.from_numpy(sp.view(np.int32)).to(dev)
        t_sd = torch.from_numpy(sd).to(dev)
        first = None
        for call in range(9):
            m = n if call < 7 else n - 64 * 100                # calls 7, 8: another shape -> another trial series
            o = [torch.zeros(3 * m, dtype=torch.int32, device=dev), torch.zeros(3 * m, dtype=torch.float32, device=dev),
                 torch.zeros(m, dtype=torch.int32, device=dev), torch.zeros(m, dtype=torch.int32, device=dev)]
            eng.trace_device(capi.VRT_F32, m, t_sp.data_ptr(), t_sd.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(),
                             o[3].data_ptr(), [1, 1, 1], 100000)
            torch.cuda.synchronize()
            got = [x.cpu().numpy() for x in o]
            if first is None:
                first = got
                assert np.array_equal(got[0].view(np.uint32).reshape(-1, 3)[idx], want["end_position"].reshape(-1, 3))
                assert np.array_equal(got[1].view(np.uint32).reshape(-1, 3)[idx], want["end_direction"].view(np.uint32).reshape(-1, 3))
                assert np.array_equal(got[2].view(np.uint32)[idx], want["end_iteration"])
            else:
                for a, b in zip(got, first):
                    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)[:a.size]), "call %d" % call


def _device_call(eng, torch, dev, sp, sd, iterations=100000, path=False, stream=None, options=None, dir_type=None):
    n = sp.size // 3
    t_sp = torch.from_numpy(sp.view(np.int32)).to(dev)
    t_sd = torch.from_numpy(sd).to(dev)
    o = [torch.zeros(3 * n, dtype=torch.int32, device=dev), torch.zeros(3 * n, dtype=t_sd.dtype, device=dev),
         torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)]
    pth = torch.zeros(iterations * 3 * n, dtype=torch.int32, device=dev) if path else None
    torch.cuda.synchronize()
    eng.trace_device(dir_type if dir_type is not None else (capi.VRT_F32 if sd.dtype == np.float32 else capi.VRT_I16), n,
                     t_sp.data_ptr(), t_sd.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(),
                     [1, 1, 1], iterations, d_path=pth.data_ptr() if path else 0,
                     stream=stream.cuda_stream if stream is not None else 0, options=options)
    return o, pth, (t_sp, t_sd)


def _device_result(o, pth=None):
    r = dict(end_position=o[0].cpu().numpy().view(np.uint32), end_direction=o[1].cpu().numpy(),
             end_iteration=o[2].cpu().numpy().view(np.uint32), remaining_light=o[3].cpu().numpy().view(np.uint32))
    if pth is not None:
        r["path"] = pth.cpu().numpy().view(np.uint32)
    return r


@pytest.mark.parametrize("fdt,ddt", [(np.float32, np.float32), (np.int16, np.int16)])
def test_device_pointer_call_on_multi_engine_handle(monkeypatch, fdt, ddt):
    """vrt_trace_device on a handle that spans several engines (the reference fans EVERY trace_rays_cu call over all
    devices, cuda_volume_raytracer.cu:804-912): one band per engine, the arrays stay in the caller's device memory and
    every engine reads / writes its band in place, forked from and joined into the caller's stream.  Three engines on
    device 0 (VRT_DEBUG_REPLICAS) must give the single engine's bits, also on a side stream, also with path tracing,
    also when the call is too small to be spread."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    bounds, d, tc, sp, sd, want = _scene(fdt=fdt, ddt=ddt, nrays=25013)
    monkeypatch.setenv("VRT_DEBUG_REPLICAS", "3")
    with capi.Engine.from_host(bounds, list(d), tc) as eng:
        assert eng.replica_count == 3
        st = torch.cuda.Stream()
        for stream in (None, st, st):
            o, _, keep = _device_call(eng, torch, dev, sp, sd, stream=stream)
            eng.status()                                            # waits for every band, reports scheduler failures
            got = _device_result(o)
            for k in KEYS:
                assert _same(got[k], want[k]), k
        o, _, keep = _device_call(eng, torch, dev, sp[:3 * 50], sd[:3 * 50])   # fewer rays than engines x 64: one band
        torch.cuda.synchronize()
        assert _same(_device_result(o)["end_iteration"], want["end_iteration"][:50])
        o, pth, keep = _device_call(eng, torch, dev, sp[:3 * 900], sd[:3 * 900], iterations=37, path=True)
        torch.cuda.synchronize()
        multi = _device_result(o, pth)
    monkeypatch.delenv("VRT_DEBUG_REPLICAS")
    with capi.Engine.from_host(bounds, list(d), tc) as eng1:
        assert eng1.replica_count == 1
        o, pth, keep = _device_call(eng1, torch, dev, sp[:3 * 900], sd[:3 * 900], iterations=37, path=True)
        torch.cuda.synchronize()
        single = _device_result(o, pth)
    for k in KEYS + ("path",):
        assert _same(multi[k], single[k]), k


@pytest.mark.parametrize("dti,dtd", [(np.float32, np.float32), (np.uint32, np.int16)])
def test_multi_engine_scene(monkeypatch, dti, dtd):
    """A scene created over several devices (vrt_options.devices = -1; three engines on device 0 under
    VRT_DEBUG_REPLICAS): field AND ior volume replicated, host-pointer and device-pointer trace_rays cut into bands --
    same bits as the oracle's scene layer."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    N = 40
    ior = W.ior_volume(N, seed=9, dtype=dti).reshape(-1)
    rng = np.random.RandomState(5)
    tr = np.full(N ** 3, 0xFFFFFFFF, np.uint32)
    tr[rng.rand(N ** 3) < 0.01] = 7
    n = 6000
    pv = 1.0 + rng.rand(n, 3) * (N - 2.001)
    sp = np.round(pv * 65536).astype(np.uint32).reshape(-1)
    dd = rng.randn(n, 3)
    dd /= np.linalg.norm(dd, axis=1)[:, None]
    sd = dd.astype(np.float32).reshape(-1) if dtd == np.float32 else np.round(dd * 256).astype(np.int16).reshape(-1)
    want = oracle.port.scene_trace([N] * 3, ior, tr, sp, sd, [1, 1, 1], 100000)
    monkeypatch.setenv("VRT_DEBUG_REPLICAS", "3")
    with capi.Scene([N] * 3, ior, tr, options=capi.default_options(devices=-1)) as sc:
        assert sc.engine().replica_count == (3 if capi.device_count() == 1 else capi.device_count())
        host = sc.trace_rays(sp, sd, [1, 1, 1], 100000)
        for k in KEYS:
            assert _same(host[k], want[k]), k
        t_sp = torch.from_numpy(sp.view(np.int32)).to(dev)
        t_sd = torch.from_numpy(sd).to(dev)
        o = [torch.zeros(3 * n, dtype=torch.int32, device=dev), torch.zeros(3 * n, dtype=t_sd.dtype, device=dev),
             torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)]
        torch.cuda.synchronize()
        sc.trace_rays_device(capi.VRT_F32 if dtd == np.float32 else capi.VRT_I16, n, t_sp.data_ptr(), t_sd.data_ptr(),
                             o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), [1, 1, 1], 100000)
        sc.status()
        got = _device_result(o)
        for k in KEYS:
            assert _same(got[k], want[k]), k


def test_scene_errors_are_sticky_and_leave_outputs_untouched():
    """(a) ray errors of device-pointer scene calls accumulate until status() is asked -- a later good call
    must not wipe an earlier call's error; (b) the host-pointer call reports a ray outside the volume BEFORE anything
    is traced, as the reference throws before the march (image_util.cpp:690-695): the output arrays stay as they were."""
    import ctypes as C
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    N = 32
    ior = W.ior_volume(N, seed=4, dtype=np.float32).reshape(-1)
    tr = np.full(N ** 3, 0xFFFFFFFF, np.uint32)
    rng = np.random.RandomState(2)
    n = 3000
    sp = np.round((1.0 + rng.rand(n, 3) * (N - 2.001)) * 65536).astype(np.uint32).reshape(-1)
    dd = rng.randn(n, 3)
    sd = (dd / np.linalg.norm(dd, axis=1)[:, None]).astype(np.float32).reshape(-1)
    bad = sp.copy()
    bad[3 * 41 + 1] = 0x8000
    with capi.Scene([N] * 3, ior, tr) as sc:
        t_sd = torch.from_numpy(sd).to(dev)
        o = [torch.zeros(3 * n, dtype=torch.int32, device=dev), torch.zeros(3 * n, dtype=torch.float32, device=dev),
             torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)]
        for arr in (bad, sp):                                       # the bad call first, then a good one, no status in between
            t_sp = torch.from_numpy(arr.view(np.int32)).to(dev)
            torch.cuda.synchronize()
            sc.trace_rays_device(capi.VRT_F32, n, t_sp.data_ptr(), t_sd.data_ptr(), o[0].data_ptr(), o[1].data_ptr(),
                                 o[2].data_ptr(), o[3].data_ptr(), [1, 1, 1], 100000)
        with pytest.raises(capi.VrtError) as e:
            sc.status()
        assert e.value.code == -7 and "ray 41 " in str(e.value)
        sc.status()                                                 # reported once, then clear
        # host-pointer call with the bad ray: error, outputs untouched
        L = capi.load()
        ep = np.full(3 * n, 0xABABABAB, np.uint32)
        ed = np.full(3 * n, 7.0, np.float32)
        ei = np.full(n, 0xCDCDCDCD, np.uint32)
        rl = np.full(n, 0x12121212, np.uint32)
        inv = np.ones(3, np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = L.vrt_scene_trace(sc._h, capi.VRT_F32, n, p(bad), p(sd), p(ep), p(ed), p(ei), p(rl), None, p(inv), 0, 100000, 1, None)
        assert rc == -7 and b"ray 41:" in L.vrt_last_error()
        assert (ep == 0xABABABAB).all() and (ed == 7.0).all() and (ei == 0xCDCDCDCD).all() and (rl == 0x12121212).all()


def test_scheduler_give_up_is_reported(monkeypatch):
    """The time-sliced schedule bounds every wait; a wavefront that hits the bound leaves rays unfinished, and that must
    never pass silently (VERDICT r2 / ADVICE r2).  VRT_DEBUG_SPIN_LIMIT=1 forces the give-up branch: the device-pointer
    call then fails in vrt_trace_status (once), the host-pointer call fails itself, and the same engine works again
    without the limit."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    N = 64
    bounds, dx, dy, dz, tc = W.seam_field(N, np.float32, seed=12)
    sp, sd = W.random_rays(N, 128000, np.float32, seed=77, vol_seed=12)       # 2000 groups of very unequal length
    field = oracle.port.interleave(dx, dy, dz, tc)
    idx = np.linspace(0, sp.size // 3 - 1, 3000).astype(np.int64)
    want = oracle.port.trace(field, bounds, sp.reshape(-1, 3)[idx].reshape(-1), sd.reshape(-1, 3)[idx].reshape(-1), [1, 1, 1], 100000)
    opts = capi.default_options(kernel=capi.KERNEL_SLAB, schedule=3)
    with capi.Engine.from_host(bounds, [dx, dy, dz], tc) as eng:
        monkeypatch.setenv("VRT_DEBUG_SPIN_LIMIT", "1")
        monkeypatch.setenv("VRT_SLICE_STEPS", "64")
        o, _, keep = _device_call(eng, torch, dev, sp, sd, options=opts)
        assert eng.last_schedule() == 3
        with pytest.raises(capi.VrtError) as e:
            eng.status()
        assert e.value.code == -3 and "gave up waiting" in str(e.value)
        eng.status()                                                # sticky until reported, then clear
        with pytest.raises(capi.VrtError) as e:
            eng.trace(sp, sd, [1, 1, 1], 100000, options=opts)
        assert e.value.code == -3 and "gave up waiting" in str(e.value)
        monkeypatch.delenv("VRT_DEBUG_SPIN_LIMIT")
        o, _, keep = _device_call(eng, torch, dev, sp, sd, options=opts)
        eng.status()
        got = _device_result(o)
        assert np.array_equal(got["end_position"].reshape(-1, 3)[idx], want["end_position"].reshape(-1, 3))
        assert np.array_equal(got["end_direction"].view(np.uint32).reshape(-1, 3)[idx], want["end_direction"].view(np.uint32).reshape(-1, 3))
        assert np.array_equal(got["end_iteration"][idx], want["end_iteration"])
        # the schedule a launch used is observable (bench.py picks its counter profile by it)
        for sched, expect in ((1, 1), (2, 2), (3, 3)):
            o, _, keep = _device_call(eng, torch, dev, sp, sd, options=capi.default_options(kernel=capi.KERNEL_SLAB, schedule=sched))
            assert eng.last_schedule() == expect
        o, _, keep = _device_call(eng, torch, dev, sp, sd, options=capi.default_options(kernel=capi.KERNEL_DIRECT))
        torch.cuda.synchronize()
        assert eng.last_schedule() == 0


@pytest.mark.parametrize("fdt,ddt", [(np.float32, np.float32), (np.int16, np.int16)])
def test_tile_order_kernels_on_uneven_ray_sets(monkeypatch, fdt, ddt):
    """The hand-written tile order (count per tile, scan, scatter, in-tile ranks: vrt_sort.h) must yield a permutation
    whatever the ray set looks like -- every ray marched exactly once, results at the rays' own indices: random rays in
    all six dominant directions and several depth bins (tiles holding rays of several kinds), 30 000 rays crowded
    into one spot plus a few outliers that stretch the tile grid (segments far longer than the 512 the in-tile pass
    orders: the copy-through branch), many identical rays, and a set just above the sort threshold.  Same bits as the
    oracle, and as the radix-sort order of rounds 1 and 2."""
    N = 56
    bounds, dx, dy, dz, tc = W.seam_field(N, fdt, seed=31, opaque_radius=4.0)
    field = oracle.port.interleave(dx, dy, dz, tc)
    rng = np.random.RandomState(123)
    sp_r, sd_r = W.random_rays(N, 20000, ddt, seed=7, vol_seed=31)
    # crowded: starts within a quarter voxel of one point, directions in a narrow cone; plus 5 far-away outliers
    c = np.array([20.3, 31.7, 25.1])
    pv = np.concatenate([c + (rng.rand(30000, 3) - 0.5) * 0.25, 2.0 + rng.rand(5, 3) * (N - 6.0)])
    dd = np.array([1.0, 0.05, -0.02]) + rng.randn(30005, 3) * 0.01
    dd /= np.linalg.norm(dd, axis=1)[:, None]
    sp_c = np.round((pv - 1.0) * 65536).astype(np.uint32).reshape(-1)
    sd_c = dd.astype(np.float32).reshape(-1) if ddt == np.float32 else np.round(dd * 256).astype(np.int16).reshape(-1)
    sp_i, sd_i = np.tile(sp_r[:3], 5000), np.tile(sd_r[:3], 5000)              # 5000 identical rays
    sets = {"random": (sp_r, sd_r), "crowded": (sp_c, sd_c), "identical": (sp_i, sd_i), "threshold": (sp_r[:3 * 4097], sd_r[:3 * 4097])}
    with capi.Engine.from_host(bounds, [dx, dy, dz], tc) as eng:
        for name, (sp, sd) in sets.items():
            want = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], 100000)
            for order in ("tiles", "radix"):
                monkeypatch.setenv("VRT_RAY_ORDER", order)
                for kern in (capi.KERNEL_SLAB, capi.KERNEL_DIRECT):
                    got = eng.trace(sp, sd, [1, 1, 1], 100000, options=capi.default_options(kernel=kern, sort_rays=1))
                    for k in KEYS:
                        assert _same(got[k], want[k]), (name, order, kern, k)
