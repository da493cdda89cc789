"""GPU (-m gpu): the scene layer -- derivative stencil + interleave and
the ray front end on the device -- against the reference's golden vectors and the oracle.
Bit-exact, both families."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
from volumeraytracer_amd import capi, cuda_raytrace, workloads as W

pytestmark = pytest.mark.gpu
KEYS = ("end_position", "end_direction", "end_iteration", "remaining_light")


def _same(a, b):
    return a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.parametrize("tag", ["f32", "i16"])
def test_scene_golden(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "L4_scene_%s.npz" % tag))
    with capi.Scene(g["bounds"], g["ior"], g["translucency"]) as sc:
        out = sc.trace_rays(g["start_position"], g["start_direction"], g["invscale"], int(g["iterations"]))
        for k in KEYS:
            assert _same(out[k], g[k]), k
        outp = sc.trace_rays(g["start_position"][:48], g["start_direction"][:48], g["invscale"], 40, trace_paths=True)
        assert _same(outp["path"], g["path_path40"])
        assert _same(outp["end_position"], g["end_position_path40"])


@pytest.mark.parametrize("tag", ["f32", "i16"])
def test_scaling_test_fully_on_gpu(golden_dir, tag):
    """The reference's own scaling_test (cuda_volume_raytracer_test.h:10-52) end to end on the GPU."""
    g = np.load(os.path.join(golden_dir, "F1_scaling_%s.npz" % tag))
    with capi.Scene(g["bounds"], g["ior"], g["translucency"]) as sc:
        out = sc.trace_rays(g["start_position"], g["start_direction"], g["invscale"], int(g["iterations"]))
    for k in KEYS:
        assert _same(out[k], g[k]), k
    assert abs(int(out["end_iteration"][0]) - 46718) <= 100 and abs(int(out["end_iteration"][1]) - 46718) <= 100


def test_config1_python_binding_signature(golden_dir):
    """BASELINE config 1 (64^3, 256x256 orthographic) through the cuda_raytrace() mirror of python_binding.cpp."""
    summary = json.load(open(os.path.join(golden_dir, "F6_config1.json")))
    N, R = 64, 256
    ior = W.ior_volume(N, dtype=np.uint32).reshape(-1)
    tr = np.full(N ** 3, 0xFFFFFFFF, np.uint32)
    t = np.linspace(1.5, N - 2.5, R)
    yy, zz = np.meshgrid(t, t, indexing="ij")
    pv = np.stack([np.full(R * R, 1.5), yy.reshape(-1), zz.reshape(-1)], 1)
    sp = np.round(pv * 65536).astype(np.uint32).reshape(-1)
    sd = np.zeros((R * R, 3), np.int16)
    sd[:, 0] = 0x100
    res = cuda_raytrace.cuda_raytrace([N, N, N], ior, tr, sp, sd.reshape(-1), [1.0, 1.0, 1.0], 0, 100000, False)
    assert isinstance(res, tuple) and len(res) == 4 and isinstance(res[0], list)
    dts = (np.uint32, np.int16, np.uint32, np.uint32)
    for k, arr, dt in zip(KEYS, res, dts):
        assert hashlib.sha256(np.asarray(arr, dtype=dt).tobytes()).hexdigest() == summary["sha256_" + k], k


@pytest.mark.parametrize("dti,dtd", [(np.float32, np.float32), (np.uint32, np.int16)])
def test_scene_vs_oracle_random(dti, dtd):
    N = 40
    ior = W.ior_volume(N, seed=9, dtype=dti).reshape(-1)
    rng = np.random.RandomState(21)
    tr = np.full(N ** 3, 0xFFFFFFFF, np.uint32)
    tr[rng.rand(N ** 3) < 0.01] = 7
    pv = 1.0 + rng.rand(3000, 3) * (N - 2.001)
    sp = np.round(pv * 65536).astype(np.uint32).reshape(-1)
    d = rng.randn(3000, 3)
    d /= np.linalg.norm(d, axis=1)[:, None]
    sd = d.astype(np.float32).reshape(-1) if dtd == np.float32 else np.round(d * 256).astype(np.int16).reshape(-1)
    want = oracle.port.scene_trace([N] * 3, ior, tr, sp, sd, [1, 1, 1], 100000)
    with capi.Scene([N] * 3, ior, tr) as sc:
        got = sc.trace_rays(sp, sd, [1, 1, 1], 100000)
    for k in KEYS:
        assert _same(got[k], want[k]), k


def test_scene_errors():
    N = 12
    ior = W.ior_volume(N, dtype=np.float32).reshape(-1)
    tr = np.full(N ** 3, 0xFFFFFFFF, np.uint32)
    with pytest.raises(capi.VrtError) as e:                  # image_util.cpp:512-515
        capi.Scene([N, N, N + 1], ior, tr)
    assert "imagesizes doesn't match" in str(e.value)
    bad = ior.copy()
    bad[5] = 0.0
    with pytest.raises(capi.VrtError) as e:                  # image_util.cpp:605-608
        capi.Scene([N] * 3, bad, tr)
    assert "refraction-index underflow" in str(e.value)
    with capi.Scene([N] * 3, ior, tr) as sc:
        sp = np.array([0x20000, 0x20000, 0x20000, 0x8000, 0x20000, 0x20000], np.uint32)   # 2nd ray below 1.0
        sd = np.array([1, 0, 0, 1, 0, 0], np.float32)
        with pytest.raises(capi.VrtError) as e:              # image_util.cpp:690-695
            sc.trace_rays(sp, sd, [1, 1, 1], 10)
        assert e.value.code == -7 and "ray 1:" in str(e.value) and "is not in 0 to" in str(e.value)
        with pytest.raises(capi.VrtError):                   # float scene takes float directions only
            sc.trace_rays(sp[:3], np.array([256, 0, 0], np.int16), [1, 1, 1], 10)


@pytest.mark.parametrize("tag", ["f32", "i16"])
def test_lut_scene_golden(golden_dir, tag):
    """BASELINE config 5 semantics (row f-3): uint8 density + 256-entry transfer function, expanded on
    the device; the golden outputs come from the reference on the host-expanded arrays."""
    g = np.load(os.path.join(golden_dir, "F5_lut_%s.npz" % tag))
    with capi.Scene.from_lut(g["bounds"], g["density"], g["lut_ior"], g["lut_translucency"]) as sc:
        out = sc.trace_rays(g["start_position"], g["start_direction"], g["invscale"], int(g["iterations"]))
    for k in KEYS:
        assert _same(out[k], g[k]), k
    # and the expanded-array scene gives the same bits as the LUT scene
    ior = g["lut_ior"][g["density"]]
    tr = g["lut_translucency"][g["density"]]
    with capi.Scene(g["bounds"], ior, tr) as sc2:
        out2 = sc2.trace_rays(g["start_position"], g["start_direction"], g["invscale"], int(g["iterations"]))
    for k in KEYS:
        assert _same(out2[k], g[k]), k


def test_replay_reference_dump_files(golden_dir, capsys):
    """Row f-4: replay the reference-written debug_scene_instance / debug_ray_instance on the GPU
    (the raytrace_test.cpp CLI) and compare with the reference's outputs for those rays."""
    from volumeraytracer_amd import replay
    g = np.load(os.path.join(golden_dir, "L4_scene_i16.npz"))
    sc, ry, out = replay.replay(os.path.join(golden_dir, "debug_scene_instance"),
                                os.path.join(golden_dir, "debug_ray_instance"))
    assert np.array_equal(out["end_position"], g["end_position"][:192])
    assert np.array_equal(out["end_direction"], g["end_direction"][:192])
    assert np.array_equal(out["end_iteration"], g["end_iteration"][:64])
    assert replay.main([os.path.join(golden_dir, "debug_scene_instance"), os.path.join(golden_dir, "debug_ray_instance")]) == 0
    text = capsys.readouterr().out
    assert "begin " in text and ("end " + " ".join(str(int(x)) for x in g["end_position"][:192])) in text


@pytest.mark.parametrize("kind", ["f32", "f32_lut", "u32", "u32_lut"])
# the last three extents have (Ny - 2) * (Nz - 2) a multiple of four with field rows that are not: the tiled build then shifts each
# row's runs by (y * row) & 3 voxels (all four shifts with 33-voxel rows; two and three tiles per row with 70 and 129)
@pytest.mark.parametrize("dims", [(3, 3, 3), (9, 11, 35), (37, 10, 70), (41, 36, 33), (12, 10, 35), (5, 8, 72), (6, 10, 131)])
def test_scene_build_field_bits(monkeypatch, kind, dims):
    """Row f-1: the field the scene build leaves in HBM, voxel for voxel, against the oracle's scene build (which is
    pinned to the reference's, tests/test_oracle_vs_reference.py) -- for the LDS-tiled kernel (default) and the
    per-voxel kernel of round 2, at extents that are not multiples of the tile (4 x 64 outputs, 32 planes per workgroup),
    for array and LUT sources of both families.  The field scan fused into the tiled kernel must give the numbers of the
    separate scan."""
    rng = np.random.RandomState(hash((kind,) + dims) % (2 ** 31))
    nvox = dims[0] * dims[1] * dims[2]
    is_f = kind.startswith("f32")
    dens = rng.randint(0, 256, nvox).astype(np.uint8)
    if is_f:
        lut_ior = (1.0 + rng.rand(256) * 0.8).astype(np.float32)
    else:
        lut_ior = np.round((1.0 + rng.rand(256) * 0.8) * 65536).astype(np.uint32)
    lut_tr = np.where(rng.rand(256) < 0.8, 0xFFFFFFFF, rng.randint(0, 2 ** 31, 256)).astype(np.uint32)
    if kind.endswith("lut"):
        # smooth-ish density so that the integer family's int16 derivative does not overflow
        dens = (np.clip(np.cumsum(rng.randint(-1, 2, nvox)), 0, 255)).astype(np.uint8)
        ior, tr = lut_ior[dens], lut_tr[dens]
    else:
        base = 1.0 + 0.3 * rng.rand(nvox)
        ior = base.astype(np.float32) if is_f else np.round(base * 65536).astype(np.uint32)
        tr = np.where(rng.rand(nvox) < 0.9, 0xFFFFFFFF, rng.randint(0, 2 ** 31, nvox)).astype(np.uint32)
    try:
        hx, hy, hz, tc, _ = oracle.port.scene_build(list(dims), ior, tr)
    except Exception as ex:                                  # the oracle raises what the reference throws (overflow): same on the GPU
        with pytest.raises(capi.VrtError):
            capi.Scene(list(dims), ior, tr)
        return
    want = oracle.port.interleave(hx, hy, hz, tc)
    fields, scans = {}, {}
    for mode in ("tiled", "pervoxel"):
        monkeypatch.setenv("VRT_SCENE_BUILD", mode)
        sc = capi.Scene.from_lut(list(dims), dens, lut_ior, lut_tr) if kind.endswith("lut") else capi.Scene(list(dims), ior, tr)
        with sc:
            eng = sc.engine()
            fields[mode] = eng.download_field()
            scans[mode] = eng.field_scan()
    for mode in fields:
        assert np.array_equal(fields[mode].reshape(-1).view(np.uint8), np.asarray(want).reshape(-1).view(np.uint8)), mode
    assert np.array_equal(scans["tiled"][0].view(np.uint32), scans["pervoxel"][0].view(np.uint32)) and scans["tiled"][1] == scans["pervoxel"][1]

