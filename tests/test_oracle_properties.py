"""CPU: size-independent properties of the oracle and of the host-side helpers (hypothesis where it pays)."""
import io

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import oracle
from volumeraytracer_amd import serialize, workloads as W


@pytest.fixture(scope="module")
def scene():
    bounds, dx, dy, dz, tc = W.seam_field(28, np.float32, opaque_radius=3.0)
    return bounds, oracle.port.interleave(dx, dy, dz, tc)


def test_thread_count_independence(scene):
    """The reference's results are thread-count independent; so are the oracle's."""
    bounds, field = scene
    sp, sd = W.random_rays(28, 1500)
    a = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], 100000, nthreads=1)
    b = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], 100000, nthreads=8)
    for k in ("end_position", "end_direction", "end_iteration"):
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8))


def test_path_layout_invariants(scene):
    """cuda_volume_raytracer.cu:333-358: path[cap-1] is the start, entries count down with the steps,
    everything below the last step is filled with the end position."""
    bounds, field = scene
    sp, sd = W.random_rays(28, 200, seed=2)
    cap = 60
    out = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], cap, trace_paths=True)
    path = out["path"].reshape(-1, cap, 3)
    ep = out["end_position"].reshape(-1, 3)
    start = sp.reshape(-1, 3)
    for i in range(path.shape[0]):
        n = int(out["end_iteration"][i]) - 1          # completed steps
        assert np.array_equal(path[i, cap - 1], start[i])
        assert np.array_equal(path[i, cap - 1 - n], ep[i])
        assert np.all(path[i, :cap - 1 - n] == ep[i])
        assert 1 <= out["end_iteration"][i] <= cap


def test_ray_independence_and_order(scene):
    """Rays do not interact: any subset / permutation gives the same per-ray results."""
    bounds, field = scene
    sp, sd = W.random_rays(28, 400, seed=3)
    full = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], 100000)
    perm = np.random.RandomState(0).permutation(400)[:137]
    sub = oracle.port.trace(field, bounds, sp.reshape(-1, 3)[perm].reshape(-1), sd.reshape(-1, 3)[perm].reshape(-1),
                            [1, 1, 1], 100000)
    assert np.array_equal(sub["end_position"].reshape(-1, 3), full["end_position"].reshape(-1, 3)[perm])
    assert np.array_equal(sub["end_iteration"], full["end_iteration"][perm])


def test_iteration_cap_is_a_prefix(scene):
    """Stopping at cap k is the state after min(k-1, n) steps of the uncapped march."""
    bounds, field = scene
    sp, sd = W.random_rays(28, 64, seed=4)
    cap = 25
    full = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], 1000, trace_paths=True)
    capped = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], cap)
    fp = full["path"].reshape(-1, 1000, 3)
    for i in range(64):
        n_full = int(full["end_iteration"][i]) - 1
        n = min(cap - 1, n_full)
        assert np.array_equal(capped["end_position"].reshape(-1, 3)[i], fp[i, 999 - n])
        assert capped["end_iteration"][i] == n + 1


@settings(max_examples=30, deadline=None)
@given(st.integers(0, 50), st.integers(1, 4), st.booleans(), st.booleans(), st.integers(0, 2 ** 32 - 1),
       st.integers(0, 2 ** 32 - 1))
def test_ray_instance_roundtrip(n, seed, trace_path, normalize, minb, iters):
    rng = np.random.RandomState(seed)
    sp = rng.randint(0, 2 ** 32, 3 * n, dtype=np.uint64).astype(np.uint32)
    sd = rng.randint(-2 ** 15, 2 ** 15, 3 * n).astype(np.int16)
    inv = rng.rand(3).astype(np.float32)
    buf = io.BytesIO()
    serialize.write_ray_instance(buf, sp, sd, inv, minb, iters, trace_path, normalize)
    buf.seek(0)
    r = serialize.read_ray_instance(buf)
    assert np.array_equal(r["start_position"], sp) and np.array_equal(r["start_direction"], sd)
    assert np.array_equal(r["invscale"], inv)
    assert (r["minimum_brightness"], r["iterations"], r["trace_path"], r["normalize_length"]) == (minb, iters, trace_path, normalize)
    assert buf.tell() == 8 * 3 + 4 * 3 * n + 2 * 3 * n + 12 + 8 + 2


def test_workload_generators_are_in_range():
    for N, R in ((32, 17), (64, 40)):
        for gen in (W.ortho_rays, W.perspective_rays):
            sp, sd = gen(N, R)
            cell = sp.reshape(-1, 3) >> 16
            assert cell.min() >= 0 and cell.max() < N - 3
            assert sd.shape == sp.shape
        a = W.perspective_rays(N, R, rows_total=2 * R, row_range=(0, R))
        b = W.perspective_rays(N, R, rows_total=2 * R, row_range=(R, 2 * R))
        whole = W.perspective_rays(N, R, rows_total=2 * R)
        assert np.array_equal(np.concatenate([a[0], b[0]]), whole[0])     # bands tile the lattice
