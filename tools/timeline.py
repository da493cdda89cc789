#!/usr/bin/env python3
"""Wavefront timeline of one march launch (development aid; needs tools/build_debug.sh's libvrt_timeline.so):

    VRT_LIB=$PWD/tmp_libs/libvrt_timeline.so VRT_DEBUG_STATS=1 python3 tools/timeline.py [--config 3]

Prints how many wavefronts are in flight over the launch (deciles of the kernel's duration), the share of
the launch spent with less than half of the peak in flight (the tail), and the spread of per-wave run times."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from volumeraytracer_amd import capi, distributed as D  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--sort", type=int, default=0)
    ap.add_argument("--dump", default="", help="write the per-wavefront arrays to this .npz")
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    dev = torch.device("cuda:0")
    eng, scene, bounds, d, tr, _ = bench.build_engine(cfg, dev)
    sp, sd = bench.make_rays(cfg, 0, 1)
    n = sp.size // 3
    d_sp = torch.from_numpy(sp.view(np.int32)).to(dev)
    d_sd = torch.from_numpy(sd).to(dev)
    out = D.alloc_record(n, dev)
    o_ep, o_ed, o_ei, o_rl = D.record_views(out, n)
    opts = capi.default_options(sort_rays=args.sort)
    for _ in range(2):
        eng.trace_device(capi.VRT_F32, n, d_sp.data_ptr(), d_sd.data_ptr(), o_ep.data_ptr(), o_ed.data_ptr(),
                         o_ei.data_ptr(), o_rl.data_ptr(), [1.0, 1.0, 1.0], 100000, options=opts)
    torch.cuda.synchronize()
    nw = (n + 63) // 64
    raw4 = eng.debug_raw(64 + 4 * nw)[64:].reshape(nw, 4)
    raw = raw4[:, :2].astype(np.int64)
    lds_steps, dir_steps = (raw4[:, 2] & 0xFFFFFFFF).astype(np.int64), (raw4[:, 2] >> 32).astype(np.int64)
    rounds, max_steps = (raw4[:, 3] & 0xFFFFFFFF).astype(np.int64), (raw4[:, 3] >> 32).astype(np.int64)
    t0 = raw[:, 0].min()
    beg, end = (raw[:, 0] - t0) / 100.0, (raw[:, 1] - t0) / 100.0            # microseconds
    total = end.max()
    print("kernel_ms (events) %.3f, timeline span %.3f ms, %d wavefronts" % (eng.last_march_ms(), total / 1e3, nw))
    edges = np.linspace(0, total, 21)
    ev = np.concatenate([np.stack([beg, np.ones(nw)], 1), np.stack([end, -np.ones(nw)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    inflight = np.cumsum(ev[:, 1])
    # time-weighted mean of in-flight wavefronts per 5 % slice of the launch
    tt = ev[:, 0]
    means = []
    for a, b in zip(edges[:-1], edges[1:]):
        m = (tt >= a) & (tt < b)
        if m.sum() < 2:
            means.append(float("nan"))
            continue
        ts, fl = tt[m], inflight[m]
        w = np.diff(np.append(ts, b))
        means.append(float((fl * w).sum() / max(w.sum(), 1e-9)))
    peak = np.nanmax(means)
    print("in-flight wavefronts per 5%% slice: %s" % " ".join("%.0f" % x for x in means))
    dt = np.diff(np.append(tt, total))
    below = dt[inflight < 0.5 * peak].sum() / total
    print("peak %.0f; share of the launch with < half the peak in flight: %.1f %%" % (peak, 100 * below))
    run = end - beg
    print("per-wave run time us: min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f" %
          (run.min(), np.percentile(run, 10), np.median(run), np.percentile(run, 90), run.max()))
    last = np.argsort(-end)[:12]
    print("last wavefronts to finish: begin_us end_us run_us lds_steps direct_steps rounds max_ray_steps")
    for w in last:
        print("   %8.0f %8.0f %7.0f %6d %6d %5d %6d" % (beg[w], end[w], run[w], lds_steps[w], dir_steps[w], rounds[w], max_steps[w]))
    wsteps = lds_steps + dir_steps
    print("wave-steps: lds %.3e direct %.3e; us per wave-step overall %.3f; corr(run, wave-steps) %.3f; corr(run, max_ray_steps) %.3f" %
          (lds_steps.sum(), dir_steps.sum(), run.sum() / wsteps.sum(), np.corrcoef(run, wsteps)[0, 1], np.corrcoef(run, max_steps)[0, 1]))
    full = end < 0.7 * total                                     # wavefronts that lived entirely at full occupancy
    A = np.stack([lds_steps[full], dir_steps[full], rounds[full], np.ones(full.sum())], 1).astype(np.float64)
    coef, *_ = np.linalg.lstsq(A, run[full], rcond=None)
    print("fit over %d full-occupancy wavefronts: run_us = %.3f*lds_steps + %.3f*direct_steps + %.2f*rounds + %.0f  (resid rms %.0f us)" %
          (full.sum(), coef[0], coef[1], coef[2], coef[3], np.sqrt(np.mean((A @ coef - run[full]) ** 2))))
    order_by_begin = np.argsort(beg)
    q = nw // 8
    print("launch-order octiles: mean max_ray_steps / mean run_us / mean direct share")
    for k in range(8):
        sel = order_by_begin[k * q:(k + 1) * q]
        print("   %d: %6.0f %7.0f %.2f" % (k, max_steps[sel].mean(), run[sel].mean(), dir_steps[sel].sum() / max(wsteps[sel].sum(), 1)))
    if args.dump:
        np.savez_compressed(args.dump, beg=beg, end=end, lds_steps=lds_steps, dir_steps=dir_steps, rounds=rounds, max_steps=max_steps)
    steps = o_ei.cpu().numpy().view(np.uint32).astype(np.int64)
    print("steps per ray: min %d median %d max %d" % (steps.min(), np.median(steps), steps.max()))
    print("ideal (sum of wave run times / peak in flight): %.3f ms" % (run.sum() / peak / 1e3))


if __name__ == "__main__":
    main()
