#!/usr/bin/env python3
"""Time vrt_scene_create (log + stencil + interleave) at BASELINE sizes, both families."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from volumeraytracer_amd import capi, workloads as W
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for dt in (np.float32, np.uint32):
    t0 = time.time(); ior = W.ior_volume(N, dtype=dt).reshape(-1); tr = np.full(N ** 3, 0xFFFFFFFF, np.uint32)
    print("host volume synth %.1f s" % (time.time() - t0), flush=True)
    for rep in range(2):
        t0 = time.time()
        sc = capi.Scene([N] * 3, ior, tr)
        dt_s = time.time() - t0
        sc.close()
        print("N=%d %s scene create %.3f s (%.1f ns/voxel)" % (N, np.dtype(dt).name, dt_s, dt_s * 1e9 / N ** 3), flush=True)
