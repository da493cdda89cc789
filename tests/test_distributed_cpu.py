"""CPU, world_size 2 over gloo: the N > 1 path of bench.py -- ray sharding into contiguous bands,
per-rank output records, gather to rank 0, reassembly in global ray order.  The march itself is
GPU-only, so each rank's record is filled by the oracle (tests may use it); what is under test is
the partition + collective + ordering logic that runs unchanged over RCCL on the GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from volumeraytracer_amd import distributed as D, workloads as W


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, R, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bounds, dx, dy, dz, tc = W.seam_field(N, np.float32)
        field = oracle.port.interleave(dx, dy, dz, tc)
        rows = R * world
        r0, r1 = D.row_band(rows, rank, world)
        sp, sd = W.perspective_rays(N, R, rows_total=rows, row_range=(r0, r1))
        if rank == 1:                      # uneven shards: rank 1 drops its last 5 rays
            sp, sd = sp[:-15], sd[:-15]
        n = sp.size // 3
        out = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], 100000, nthreads=2)
        rec = D.alloc_record(n, "cpu")
        ep, ed, ei, rl = D.record_views(rec, n)
        ep.copy_(torch.from_numpy(out["end_position"].view(np.int32)))
        ed.copy_(torch.from_numpy(out["end_direction"].view(np.int32)))
        ei.copy_(torch.from_numpy(out["end_iteration"].view(np.int32)))
        rl.copy_(torch.from_numpy(out["remaining_light"].view(np.int32)))
        counts = [None] * world
        dist.all_gather_object(counts, n)
        recs = D.gather_records(rec, counts, dst=0)
        if rank == 0:
            whole = D.assemble(recs, counts)
            q.put({k: v for k, v in whole.items()})
        else:
            assert recs is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_matches_single_process():
    N, R, world = 32, 24, 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, R, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # single-process answer on the same global ray set (rank 1's last 5 rays dropped)
    bounds, dx, dy, dz, tc = W.seam_field(N, np.float32)
    field = oracle.port.interleave(dx, dy, dz, tc)
    sp, sd = W.perspective_rays(N, R, rows_total=R * world)
    sp, sd = sp[:-15], sd[:-15]
    want = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], 100000)
    assert np.array_equal(got["end_position"], want["end_position"])
    assert np.array_equal(got["end_direction_bits"], want["end_direction"].view(np.uint32))
    assert np.array_equal(got["end_iteration"], want["end_iteration"])
    assert np.all(got["remaining_light"] == 0xFFFFFFFF)


def _pipe_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1000 + 37 * rank                                   # uneven shards
        counts = [None] * world
        dist.all_gather_object(counts, n)
        pipe = D.RecordGather(n, counts, "cpu")
        steps = 5
        for k in range(steps):
            buf = pipe.buffer()
            assert buf.numel() == D.RECORD_WORDS * n
            buf.copy_(torch.arange(buf.numel(), dtype=torch.int32) + 100000 * rank + 7 * k)   # "march" of step k
            pipe.launch()
        recs = pipe.finish()
        if rank == 0:
            q.put([r.clone().numpy() for r in recs])
        else:
            assert recs is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_double_buffered_gather_pipeline():
    """RecordGather: the gather of step k overlaps step k+1 (two buffers); after finish() rank 0 holds every
    rank's record of the LAST step, trimmed to that rank's ray count."""
    world, steps = 2, 5
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(world):
        n = 1000 + 37 * r
        want = np.arange(D.RECORD_WORDS * n, dtype=np.int32) + 100000 * r + 7 * (steps - 1)
        assert np.array_equal(got[r], want)


def test_ray_range_partition():
    for n in (0, 1, 7, 64, 1000, 1048576):
        for world in (1, 2, 3, 8):
            ranges = [D.ray_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            for a, b in zip(ranges, ranges[1:]):
                assert a[1] == b[0]
            sizes = [e - b for b, e in ranges]
            assert max(sizes) - min(sizes) <= 1


def test_balanced_bands():
    rng = np.random.RandomState(3)
    for world in (2, 3, 4, 8):
        for rows in (world, 17, 1024):
            w = rng.rand(rows) ** 3 * 1000.0
            bands = D.balanced_bands(w, world)
            assert bands[0][0] == 0 and bands[-1][1] == rows
            assert all(b[1] > b[0] for b in bands)
            assert all(bands[i][1] == bands[i + 1][0] for i in range(world - 1))
            if rows >= 64 * world:                      # fine-grained rows: every band within a few rows' work of the mean
                work = np.array([w[a:b].sum() for a, b in bands])
                assert work.max() - work.min() <= 4 * w.max()
    assert D.balanced_bands(np.zeros(10), 2) == [(0, 5), (5, 10)]
    # a skewed profile (short rays at the rim, long in the middle) gives wide outer bands and narrow inner ones
    prof = np.concatenate([np.linspace(1, 10, 512), np.linspace(10, 1, 512)])
    b = D.balanced_bands(prof, 4)
    assert (b[0][1] - b[0][0]) > (b[1][1] - b[1][0]) and (b[3][1] - b[3][0]) > (b[2][1] - b[2][0])


def test_strong_scaling_bands_tile_the_fixed_lattice():
    """bench.py --scaling strong (BASELINE config 4: a FIXED R x R lattice cut into N row bands): the bands of ranks
    0..N-1, concatenated in rank order, are exactly the rays of the single-GPU job, for every N the driver runs;
    weak scaling instead grows the lattice to (R*N) x R with R rows per rank."""
    import bench
    cfg = dict(bench.CONFIGS[4], N=48, R=64)                       # config 4's shape at a size the CPU handles
    sp1, sd1 = bench.make_rays(cfg, 0, 1, scaling="strong")
    assert sp1.size // 3 == cfg["R"] ** 2
    for world in (2, 4, 8):
        assert bench.lattice_rows(cfg, world, "strong") == cfg["R"]
        parts = [bench.make_rays(cfg, r, world, scaling="strong") for r in range(world)]
        assert np.array_equal(np.concatenate([p[0] for p in parts]), sp1)
        assert np.array_equal(np.concatenate([p[1] for p in parts]), sd1)
        assert bench.lattice_rows(cfg, world, "weak") == cfg["R"] * world
        w = [bench.make_rays(cfg, r, world, scaling="weak") for r in range(world)]
        assert all(p[0].size // 3 == cfg["R"] ** 2 for p in w)


def test_sample_gathered_covers_every_rank():
    """bench.py's N > 1 self-check samples the gathered records evenly over the whole job; every rank's band must be hit
    (a band swapped or shifted by the gather then cannot pass), indices stay inside the rank's count."""
    for counts in ([1000, 1000], [5, 100000, 7, 3], [0, 64, 64], [1 << 20] * 8):
        pick = D.sample_gathered(counts, 4096)
        assert [r for r, _ in pick] == list(range(len(counts)))
        for (r, idx), c in zip(pick, counts):
            assert (idx.size > 0) == (c > 0)
            assert idx.size == 0 or (idx.min() >= 0 and idx.max() < c and np.all(np.diff(idx) > 0))
        assert sum(idx.size for _, idx in pick) >= min(4096, sum(counts)) - 1
