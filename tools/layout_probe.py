#!/usr/bin/env python3
"""Probe (development aid): how much does the memory-minor axis of the field matter?  The bench rays travel along x, the
memory-MAJOR axis of the field (index = (x * By + y) * Bz + z).  This runs the same physical workload mirrored x <-> z
(field arrays transposed, gradient components and ray components swapped): the rays then travel along the memory-minor axis.

    python3 tools/layout_probe.py --config 5
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from volumeraytracer_amd import capi, distributed as D  # noqa: E402


def time_march(eng, sp, sd, dev, reps=4):
    n = sp.size // 3
    d_sp = torch.from_numpy(np.ascontiguousarray(sp).view(np.int32)).to(dev)
    d_sd = torch.from_numpy(np.ascontiguousarray(sd)).to(dev)
    out = D.alloc_record(n, dev)
    ep, ed, ei, rl = D.record_views(out, n)
    best = 1e9
    for _ in range(reps + 5):
        eng.trace_device(capi.VRT_F32, n, d_sp.data_ptr(), d_sd.data_ptr(), ep.data_ptr(), ed.data_ptr(), ei.data_ptr(),
                         rl.data_ptr(), [1.0, 1.0, 1.0], 100000)
        torch.cuda.synchronize()
        best = min(best, eng.last_march_ms())
    return best, int(ei.to(torch.int64).sum().item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=5)
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    dev = torch.device("cuda:0")
    N = cfg["N"]
    sp, sd = bench.make_rays(cfg, 0, 1)
    sp_t = sp.reshape(-1, 3)[:, ::-1].reshape(-1)
    sd_t = sd.reshape(-1, 3)[:, ::-1].reshape(-1)
    for mirrored in (False, True):
        if cfg.get("lut"):
            dens = bench.synth_density_device(N, dev)
            if mirrored:
                dens = dens.view(N, N, N).permute(2, 1, 0).contiguous().view(-1)
            dens = dens.cpu().numpy()
            dd = np.arange(256, dtype=np.float64)
            lut_ior = np.round((1.0 + dd / 510.0) * 65536).astype(np.uint32)
            lut_tr = np.where(dd < 160, 0xFFFFFFFF, 0).astype(np.uint32)
            scene = capi.Scene.from_lut([N, N, N], dens, lut_ior, lut_tr)
            eng = scene.engine()
            keep = scene
        else:
            bounds, d, tr = bench.synth_field_device(N, cfg["field"], dev, opaque=cfg.get("opaque", False))
            if mirrored:
                B = bounds[0]
                t = lambda a: a.view(B, B, B).permute(2, 1, 0).contiguous().view(-1)
                d = [t(d[2]), t(d[1]), t(d[0])]
                tr = t(tr)
            torch.cuda.synchronize()
            ftype = capi.VRT_F32 if cfg["field"] == "f32" else capi.VRT_I16
            eng = capi.Engine.from_device(bounds, [x.data_ptr() for x in d], tr.data_ptr(), ftype, d[0].numel())
            keep = (d, tr)
        ms, steps = time_march(eng, sp_t if mirrored else sp, sd_t if mirrored else sd, dev)
        print("config %d, rays along the memory-%s axis: march kernel %.3f ms, %d ray-steps" %
              (args.config, "MINOR (z)" if mirrored else "MAJOR (x)", ms, steps), flush=True)
        eng.close() if not cfg.get("lut") else keep.close()
        del keep
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
