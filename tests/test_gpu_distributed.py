"""RCCL (backend "nccl") smoke of the N > 1 plumbing on one GPU: a world of ONE rank is the largest RCCL world a
one-GPU box allows (RCCL refuses two ranks on one device), but it exercises the same calls bench.py makes for
N > 1 -- asynchronous dist.gather on the collective's stream, double-buffered records, waits -- with the engine
marching into the record buffers in between.  Partition / ordering logic for world 2 is covered on CPU (gloo)."""
import os
import socket

import numpy as np
import pytest
import torch

import oracle
from volumeraytracer_amd import capi, distributed as D, workloads as W


@pytest.mark.gpu
def test_rccl_world1_overlapped_gather_matches_oracle():
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        N = 40
        bounds, dx, dy, dz, tc = W.seam_field(N, np.float32, opaque_radius=4.0)
        field = oracle.port.interleave(dx, dy, dz, tc)
        sp, sd = W.perspective_rays(N, 70)
        n = sp.size // 3
        want = oracle.port.trace(field, bounds, sp, sd, [1, 1, 1], 100000)
        d_sp = torch.from_numpy(sp.view(np.int32)).to(dev)
        d_sd = torch.from_numpy(sd).to(dev)
        stream = torch.cuda.current_stream().cuda_stream
        with capi.Engine.from_host(bounds, [dx, dy, dz], tc) as eng:
            pipe = D.RecordGather(n, [n], dev, dst=0)
            for _ in range(5):                                   # more steps than buffers: every buffer is reused
                ep, ed, ei, rl = D.record_views(pipe.buffer(), n)
                eng.trace_device(capi.VRT_F32, n, d_sp.data_ptr(), d_sd.data_ptr(), ep.data_ptr(), ed.data_ptr(),
                                 ei.data_ptr(), rl.data_ptr(), [1, 1, 1], 100000, stream=stream)
                pipe.launch()
            recs = pipe.finish()
            torch.cuda.synchronize()
        got = D.assemble(recs, [n])
        assert np.array_equal(got["end_position"], want["end_position"])
        assert np.array_equal(got["end_direction_bits"], want["end_direction"].view(np.uint32))
        assert np.array_equal(got["end_iteration"], want["end_iteration"])
        assert np.all(got["remaining_light"] == 0xFFFFFFFF)
    finally:
        dist.destroy_process_group()
