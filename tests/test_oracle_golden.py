"""CPU: the oracle (oracle/vrt_oracle.c) against the golden vectors the REAL reference produced
(tests/golden/make_golden.py) and against the reference's own scaling_test known answers
(/root/reference/src/cuda_volume_raytracer_test.h:48-52).  Bit-exact everywhere."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle

KEYS = ("end_position", "end_direction", "end_iteration", "remaining_light")


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _same(a, b):
    return a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.parametrize("tag", ["f32", "i16"])
def test_scaling_known_answers(golden_dir, tag):
    g = _load(golden_dir, "F1_scaling_" + tag)
    out = oracle.port.scene_trace(g["bounds"], g["ior"], g["translucency"], g["start_position"],
                                  g["start_direction"], g["invscale"], int(g["iterations"]))
    for k in KEYS:
        assert _same(out[k], g[k]), k
    # the reference's own asserts: end_iteration == 46718 +- 100, |end_dir/start_dir| == ior(end_pos)
    assert abs(int(out["end_iteration"][0]) - 46718) <= 100
    assert abs(int(out["end_iteration"][1]) - 46718) <= 100
    sd, ed = g["start_direction"].astype(np.float64), out["end_direction"].astype(np.float64)
    tol = 1e-5 + (1 / 256 if tag == "i16" else 0)
    assert abs(ed[0] / sd[0] - 2.0) < 2.0 * tol + 3e-3     # ray 0 ends in the ior = 2 slab
    assert abs(ed[3] / sd[3] - 1.0) < tol + 3e-3           # ray 1 ends in the ior = 1 slab
    assert np.all(out["remaining_light"] == 0xFFFFFFFF)


@pytest.mark.parametrize("ftag", ["f32", "i16"])
@pytest.mark.parametrize("dtag", ["f32", "i16"])
def test_seam_golden(golden_dir, ftag, dtag):
    g = _load(golden_dir, "F2_seam_%s_%s" % (ftag, dtag))
    field = oracle.port.interleave(g["dx"], g["dy"], g["dz"], g["transl_cropped"])
    for j in range(3):
        out = oracle.port.trace(field, g["bounds"], g["start_position"], g["start_direction"],
                                g["invscale%d" % j], 100000)
        for k in KEYS:
            assert _same(out[k], g["%s_%d" % (k, j)]), (k, j)
    out = oracle.port.trace(field, g["bounds"], g["start_position"][:96], g["start_direction"][:96],
                            [1, 1, 1], 48, trace_paths=True)
    for k in KEYS + ("path",):
        assert _same(out[k], g[k + "_path48"]), k
    for cap in (1, 2, 3, 17):
        out = oracle.port.trace(field, g["bounds"], g["start_position"], g["start_direction"], [1, 1, 1], cap)
        for k in KEYS:
            assert _same(out[k], g["%s_cap%d" % (k, cap)]), (k, cap)
        assert out["end_iteration"].max() <= cap


@pytest.mark.parametrize("ftag", ["f32", "i16"])
def test_opaque_termination(golden_dir, ftag):
    g = _load(golden_dir, "F3_opaque_" + ftag)
    field = oracle.port.interleave(g["dx"], g["dy"], g["dz"], g["transl_cropped"])
    out = oracle.port.trace(field, g["bounds"], g["start_position"], g["start_direction"], g["invscale"],
                            int(g["iterations"]))
    for k in KEYS:
        assert _same(out[k], g[k]), k
    # some rays must actually have stopped inside the field on the w > 0 test
    B = g["bounds"].astype(np.int64)
    ep = out["end_position"].reshape(-1, 3).astype(np.int64) >> 16
    inside = np.all((ep >= 0) & (ep < B - 1), axis=1) & (out["end_iteration"] < int(g["iterations"]))
    assert inside.sum() > 10


@pytest.mark.parametrize("tag", ["f32", "i16"])
def test_scene_level_golden(golden_dir, tag):
    g = _load(golden_dir, "L4_scene_" + tag)
    dx, dy, dz, tc, ob = oracle.port.scene_build(g["bounds"], g["ior"], g["translucency"])
    assert _same(dx, g["ref_dx"]) and _same(dy, g["ref_dy"]) and _same(dz, g["ref_dz"])
    assert _same(tc, g["ref_transl_cropped"])
    out = oracle.port.scene_trace(g["bounds"], g["ior"], g["translucency"], g["start_position"],
                                  g["start_direction"], g["invscale"], int(g["iterations"]))
    for k in KEYS:
        assert _same(out[k], g[k]), k
    outp = oracle.port.scene_trace(g["bounds"], g["ior"], g["translucency"], g["start_position"][:48],
                                   g["start_direction"][:48], g["invscale"], 40, trace_paths=True)
    assert _same(outp["path"], g["path_path40"])
    assert _same(outp["end_position"], g["end_position_path40"])


def test_config1_hashes(golden_dir):
    """BASELINE config 1 (64^3, 256x256 orthographic, the python_binding.cpp int16 instantiation)."""
    from volumeraytracer_amd import workloads as W
    with open(os.path.join(golden_dir, "F6_config1.json")) as f:
        summary = json.load(f)
    N, R = 64, 256
    ior = W.ior_volume(N, dtype=np.uint32).reshape(-1)
    tr = np.full(N ** 3, 0xFFFFFFFF, np.uint32)
    t = np.linspace(1.5, N - 2.5, R)
    yy, zz = np.meshgrid(t, t, indexing="ij")
    pv = np.stack([np.full(R * R, 1.5), yy.reshape(-1), zz.reshape(-1)], 1)
    sp = np.round(pv * 65536).astype(np.uint32).reshape(-1)
    sd = np.zeros((R * R, 3), np.int16)
    sd[:, 0] = 0x100
    out = oracle.port.scene_trace([N] * 3, ior, tr, sp, sd.reshape(-1), [1, 1, 1], 100000)
    assert int(out["end_iteration"].astype(np.int64).sum()) == summary["sum_end_iteration"]
    for k in KEYS:
        assert hashlib.sha256(np.ascontiguousarray(out[k]).tobytes()).hexdigest() == summary["sha256_" + k], k


def test_edge_cases():
    from volumeraytracer_amd import workloads as W
    bounds, dx, dy, dz, tc = W.seam_field(12, np.float32)
    field = oracle.port.interleave(dx, dy, dz, tc)
    # empty ray set
    out = oracle.port.trace(field, bounds, np.zeros(0, np.uint32), np.zeros(0, np.float32), [1, 1, 1], 10)
    assert out["end_iteration"].size == 0
    # a ray that starts outside the field takes no step: end_iteration == 1, state unchanged
    sp = np.array([0xFFFF0000, 0x20000, 0x20000], np.uint32)
    sd = np.array([1.0, 0.0, 0.0], np.float32)
    out = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], 100)
    assert out["end_iteration"][0] == 1 and np.array_equal(out["end_position"], sp)
    assert np.array_equal(out["end_direction"], sd)
    # iterations == 0 is rejected (the reference underflows)
    with pytest.raises(RuntimeError):
        oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], 0)


def test_w_channel_extremes():
    """w = (0x7FFFFFFF - translucency) / 0x10000 with int64 arithmetic (cuda_volume_raytracer.cu:656-659)."""
    tr = np.array([0, 1, 0x7FFFFFFF, 0x80000000, 0x8000FFFF, 0x80010000, 0xFFFFFFFF], np.uint32)
    z = np.zeros(tr.size, np.float32)
    f = oracle.port.interleave(z, z, z, tr)[:, 3]
    assert f.tolist() == [32767.0, 32767.0, 0.0, 0.0, -1.0, -1.0, -32768.0]
    zi = np.zeros(tr.size, np.int16)
    fi = oracle.port.interleave(zi, zi, zi, tr)[:, 3]
    assert fi.tolist() == [32767, 32767, 0, 0, -1, -1, -32768]


@pytest.mark.parametrize("tag", ["f32", "i16"])
def test_lut_expansion_golden(golden_dir, tag):
    """Config 5 semantics: the oracle on the host-expanded (ior, translucency) arrays."""
    g = _load(golden_dir, "F5_lut_" + tag)
    ior = g["lut_ior"][g["density"]]
    tr = g["lut_translucency"][g["density"]]
    out = oracle.port.scene_trace(g["bounds"], ior, tr, g["start_position"], g["start_direction"], g["invscale"],
                                  int(g["iterations"]))
    for k in KEYS:
        assert _same(out[k], g[k]), k
    assert (out["end_iteration"] < 20).sum() > 10      # opaque cores stop rays early (w > 0)
