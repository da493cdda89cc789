"""Multi-GPU plumbing for trace_rays: one process per GPU, field replicated, rays sharded.

The reference's multi-device strategy is an OpenMP thread per CUDA device pulling 0x8000-ray
chunks, with the field replicated and every result copied back through host memory
(/root/reference/src/cuda_volume_raytracer.cu:676-686, 804-912); it has no collective.  Rays are
independent and the field is read-only, so here each rank owns a contiguous range of rays (row
bands of the ray lattice keep entry tiles coherent), marches them with its own engine, and the
per-ray output records (32 B/ray) are gathered to rank 0 with one torch.distributed gather --
RCCL over xGMI with backend "nccl", gloo on CPU for the tests.  No other exchange exists on the
path.
"""
import numpy as np
import torch

RECORD_WORDS = 8   # end_pos 3 + end_dir 3 + end_iter 1 + remaining_light 1 (32-bit words per ray)


def ray_range(nrays, rank, world):
    """Contiguous, balanced range [begin, end) of rays owned by `rank` (first ranks take the remainder)."""
    base, rem = divmod(int(nrays), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def row_band(rows_total, rank, world):
    """Contiguous band of lattice rows owned by `rank` (whole rows keep a rank's rays a compact slab)."""
    return ray_range(rows_total, rank, world)


def balanced_bands(row_work, world):
    """Contiguous row bands of (nearly) equal work: row_work[i] = measured ray-steps of lattice row i (any
    non-negative weights).  Returns [(begin, end)] * world covering all rows, every band at least one row.
    SURVEY 8(e): balance by ray count first, by measured steps if the counts leave the ranks skewed."""
    w = np.maximum(np.asarray(row_work, dtype=np.float64), 0.0)
    rows = w.size
    assert rows >= world
    csum = np.concatenate([[0.0], np.cumsum(w)])
    total = csum[-1]
    if total <= 0.0:
        return [ray_range(rows, r, world) for r in range(world)]
    cuts = [0]
    for r in range(1, world):
        c = int(np.searchsorted(csum, total * r / world, side="left"))     # first row index whose prefix reaches the share
        if c > 0 and abs(csum[c - 1] - total * r / world) <= abs(csum[min(c, rows)] - total * r / world):
            c -= 1
        c = max(c, cuts[-1] + 1)                                            # at least one row per band ...
        c = min(c, rows - (world - r))                                      # ... for the ranks still to come as well
        cuts.append(c)
    cuts.append(rows)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def record_views(buf, n):
    """Views into one rank's contiguous output record [end_pos 3n | end_dir 3n | end_iter n | light n]
    (int32 storage; end_dir holds float32 or int16-widened bits as written by the engine)."""
    assert buf.numel() == RECORD_WORDS * n
    return buf[0:3 * n], buf[3 * n:6 * n], buf[6 * n:7 * n], buf[7 * n:8 * n]


def alloc_record(n, device):
    return torch.empty(RECORD_WORDS * n, dtype=torch.int32, device=device)


def gather_records(local, counts, dst=0, group=None):
    """Gather every rank's record to `dst`.  `counts[r]` = number of rays of rank r.  Ranks may own
    different counts: records are padded to the largest one for the collective and trimmed on dst.
    Returns the list of per-rank records on dst, None elsewhere."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nmax = max(counts)
    send = local
    if local.numel() != RECORD_WORDS * nmax:
        send = torch.zeros(RECORD_WORDS * nmax, dtype=local.dtype, device=local.device)
        send[:local.numel()] = local
    recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return [recv[r][:RECORD_WORDS * counts[r]] for r in range(world)]


class RecordGather:
    """Double-buffered gather of per-rank records to `dst`: the gather of step k runs on the collective's own
    stream while step k+1 marches into the other buffer, so the only exchange on the path costs no step time.

        pipe = RecordGather(n_local, counts, device)
        for each step:
            buf = pipe.buffer()          # waits until the gather that last used this buffer is done
            ... march into record_views(buf, n_local) on the current stream ...
            pipe.launch()                # async gather of buf (ordered after the current stream's work)
        records = pipe.finish()          # all gathers complete; dst gets the last step's records, others None
    """

    def __init__(self, n_local, counts, device, dst=0, group=None, depth=2):
        import torch.distributed as dist
        self.dist, self.group, self.dst = dist, group, dst
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.n, self.counts, self.nmax = int(n_local), list(counts), max(counts)
        self.bufs = [torch.zeros(RECORD_WORDS * self.nmax, dtype=torch.int32, device=device) for _ in range(depth)]
        self.recv = [[torch.empty_like(self.bufs[0]) for _ in range(self.world)] if self.rank == dst else None
                     for _ in range(depth)]
        self.work = [None] * depth
        self.k = 0
        self.last = None

    def buffer(self):
        i = self.k % len(self.bufs)
        if self.work[i] is not None:
            self.work[i].wait()
            self.work[i] = None
        return self.bufs[i][:RECORD_WORDS * self.n]

    def launch(self):
        i = self.k % len(self.bufs)
        self.work[i] = self.dist.gather(self.bufs[i], self.recv[i], dst=self.dst, group=self.group, async_op=True)
        self.last = i
        self.k += 1

    def finish(self):
        for i, w in enumerate(self.work):
            if w is not None:
                w.wait()
                self.work[i] = None
        if self.rank != self.dst or self.last is None:
            return None
        return [self.recv[self.last][r][:RECORD_WORDS * self.counts[r]] for r in range(self.world)]


def sample_gathered(counts, nsample):
    """Evenly strided sample of the whole job's rays for the check of the gathered records (bench.py, N > 1): global ray
    indices 0, s, 2s, ... over sum(counts), split by owning rank.  Returns [(rank, local index array)] for every rank
    (possibly empty arrays); every rank with at least one ray gets at least one sample."""
    counts = [int(c) for c in counts]
    total = sum(counts)
    if total == 0 or nsample <= 0:
        return [(r, np.zeros(0, np.int64)) for r in range(len(counts))]
    g = np.unique(np.linspace(0, total - 1, min(int(nsample), total)).astype(np.int64))
    out, base = [], 0
    for r, c in enumerate(counts):
        loc = g[(g >= base) & (g < base + c)] - base
        if loc.size == 0 and c > 0:
            loc = np.array([c // 2], np.int64)
        out.append((r, loc))
        base += c
    return out


def assemble(records, counts):
    """dst side: concatenate per-rank records back into whole-job arrays in global ray order
    (rank r's rays precede rank r+1's).  Returns dict of numpy arrays (bit patterns, uint32)."""
    ep, ed, ei, rl = [], [], [], []
    for rec, n in zip(records, counts):
        a, b, c, d = record_views(rec, n)
        ep.append(a.cpu().numpy().view(np.uint32))
        ed.append(b.cpu().numpy().view(np.uint32))
        ei.append(c.cpu().numpy().view(np.uint32))
        rl.append(d.cpu().numpy().view(np.uint32))
    return dict(end_position=np.concatenate(ep), end_direction_bits=np.concatenate(ed),
                end_iteration=np.concatenate(ei), remaining_light=np.concatenate(rl))
