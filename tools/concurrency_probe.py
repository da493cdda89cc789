#!/usr/bin/env python3
"""Probe: does one config-3 launch fill the GPU?  Time 1 launch vs 2 identical launches on 2 streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from volumeraytracer_amd import capi, distributed as D

cfg = bench.CONFIGS[int(sys.argv[1]) if len(sys.argv) > 1 else 3]
dev = torch.device("cuda", 0)
bounds, d, tr = bench.synth_field_device(cfg["N"], cfg["field"], dev)
torch.cuda.synchronize()
engs = [capi.Engine.from_device(bounds, [x.data_ptr() for x in d], tr.data_ptr(), capi.VRT_F32, d[0].numel()) for _ in range(2)]
sp, sd = bench.make_rays(cfg, 0, 1)
n = sp.size // 3
d_sp = torch.from_numpy(sp.view(np.int32)).to(dev); d_sd = torch.from_numpy(sd).to(dev)
outs = [D.alloc_record(n, dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
opts = capi.default_options(kernel=capi.KERNEL_SLAB)
def launch(i):
    ep, ed, ei, rl = D.record_views(outs[i], n)
    engs[i].trace_device(capi.VRT_F32, n, d_sp.data_ptr(), d_sd.data_ptr(), ep.data_ptr(), ed.data_ptr(), ei.data_ptr(), rl.data_ptr(),
                         [1, 1, 1], 100000, stream=streams[i].cuda_stream, options=opts)
for k in (1, 2):
    for _ in range(2):
        for i in range(k): launch(i)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        for i in range(k): launch(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%d concurrent launch(es): %.2f ms per round -> %.2f ms per launch-equivalent" % (k, dt * 1e3, dt * 1e3 / k))
