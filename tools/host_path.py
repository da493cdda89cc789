#!/usr/bin/env python3
"""Host-pointer entry point (vrt_trace, what the shim / JNI / pybind11 users call) against the device-resident march:
PCIe-inclusive time of one call on BASELINE config 3, by number of pipelined chunks (vrt_options.host_lanes).

    python3 tools/host_path.py [--config 3]

(The output arrays are allocated once outside the timed region, as a JNI caller's direct buffers are.)"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from volumeraytracer_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    dev = torch.device("cuda:0")
    eng, scene, bounds, d, tr, _ = bench.build_engine(cfg, dev)
    sp, sd = bench.make_rays(cfg, 0, 1)
    n = sp.size // 3
    L = capi.load()
    ep, ed = np.empty(3 * n, np.uint32), np.empty(3 * n, np.float32)
    ei, rl = np.empty(n, np.uint32), np.empty(n, np.uint32)
    inv = np.ones(3, np.float32)
    # device-resident reference time
    d_sp = torch.from_numpy(sp.view(np.int32)).to(dev)
    d_sd = torch.from_numpy(sd).to(dev)
    o = [torch.empty(3 * n, dtype=torch.int32, device=dev), torch.empty(3 * n, dtype=torch.float32, device=dev),
         torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)]
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.trace_device(capi.VRT_F32, n, d_sp.data_ptr(), d_sd.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(),
                         o[3].data_ptr(), [1, 1, 1], 100000)
        torch.cuda.synchronize()
        dev_ms = (time.perf_counter() - t0) * 1e3
    print("device-resident call (ray order + march): %.2f ms (march kernel %.2f ms)" % (dev_ms, eng.last_march_ms()))
    ref = o[2].cpu().numpy().view(np.uint32)
    # the floor: the same bytes as bare copies (24 MB up, 32 MB down), from pageable and from pinned host memory
    for pinned in (False, True):
        h_up = [torch.from_numpy(sp.view(np.int32)), torch.from_numpy(sd)]
        h_dn = [torch.empty_like(x, device="cpu") for x in o]
        if pinned:
            h_up = [x.pin_memory() for x in h_up]
            h_dn = [x.pin_memory() for x in h_dn]
        best_up = best_dn = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d_sp.copy_(h_up[0], non_blocking=True); d_sd.copy_(h_up[1], non_blocking=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for hx, dx in zip(h_dn, o):
                hx.copy_(dx, non_blocking=True)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            best_up, best_dn = min(best_up, (t1 - t0) * 1e3), min(best_dn, (t2 - t1) * 1e3)
        print("bare copies, %s host memory: up %.2f ms (%.1f GB/s), down %.2f ms (%.1f GB/s)" %
              ("pinned" if pinned else "pageable", best_up, 24 * n / 1e6 / best_up, best_dn, 32 * n / 1e6 / best_dn))
    for lanes in (1, 2, 4, 8, 0):
        opts = capi.default_options(host_lanes=lanes)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            capi.check(L.vrt_trace(eng.handle, capi.VRT_F32, n, sp.ctypes.data_as(C.c_void_p), sd.ctypes.data_as(C.c_void_p),
                                   ep.ctypes.data_as(C.c_void_p), ed.ctypes.data_as(C.c_void_p), ei.ctypes.data_as(C.c_void_p),
                                   rl.ctypes.data_as(C.c_void_p), None, inv.ctypes.data_as(C.c_void_p), 0, 100000, C.byref(opts)))
            best = min(best, (time.perf_counter() - t0) * 1e3)
        assert np.array_equal(ei, ref)
        print("host-pointer vrt_trace, %d rays, host_lanes=%d%s: %.2f ms  (+%.2f ms over the device-resident call)"
              % (n, lanes, " (auto)" if lanes == 0 else "", best, best - dev_ms))


if __name__ == "__main__":
    main()
