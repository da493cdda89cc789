#!/usr/bin/env python3
"""Probe (development aid): what would marching in several passes with a re-sort of the surviving rays in between buy?

    python3 tools/multipass_probe.py --config 5 --pass-steps 1500

Pass 1 marches every ray for at most S steps through the public entry point (iterations = S + 1), pass 2 takes the rays that
are still inside the field from their end positions / directions (the engine's tile order is recomputed from those), and so
on.  Prints the march-kernel time of every pass next to the single-pass time, and checks that the summed step counts agree."""
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
    ap.add_argument("--config", type=int, default=5)
    ap.add_argument("--pass-steps", type=int, nargs="+", default=[1500])
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    dev = torch.device("cuda:0")
    eng, scene, bounds, d, tr, _ = bench.build_engine(cfg, dev)
    sp, sd = bench.make_rays(cfg, 0, 1)
    n = sp.size // 3
    B = torch.tensor(bounds, device=dev, dtype=torch.int64)

    def march(d_sp, d_sd, m, iters):
        out = D.alloc_record(m, dev)
        ep, ed, ei, rl = D.record_views(out, m)
        best = 1e9
        for _ in range(args.reps):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.trace_device(capi.VRT_F32, m, d_sp.data_ptr(), d_sd.data_ptr(), ep.data_ptr(), ed.data_ptr(), ei.data_ptr(),
                             rl.data_ptr(), [1.0, 1.0, 1.0], iters, stream=torch.cuda.current_stream().cuda_stream)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return ep, ed, ei, eng.last_march_ms(), best

    d_sp = torch.from_numpy(sp.view(np.int32)).to(dev)
    d_sd = torch.from_numpy(sd).to(dev)
    ep, ed, ei, k_ms, c_ms = march(d_sp, d_sd, n, 100000)
    total = int(ei.to(torch.int64).sum().item())
    print("config %d single pass: kernel %.3f ms, call %.3f ms, %d ray-steps" % (args.config, k_ms, c_ms, total))
    for S in args.pass_steps:
        cur_sp, cur_sd, m, steps, k_sum, c_sum, left, p = d_sp, d_sd, n, 0, 0.0, 0.0, 100000, 0
        line = []
        while m > 0 and p < 8:
            iters = min(S + 1, left) if p < 7 else left
            ep, ed, ei, k_ms, c_ms = march(cur_sp, cur_sd, m, iters)
            # the loop runs iterations - 1 steps at most (.cu:333-335); end_iteration counts the steps taken + 1 unless the cap was hit
            pos = ep.view(torch.int32).to(torch.int64).view(-1, 3) & 0xFFFFFFFF
            inside = ((pos >> 16) < (B - 1)).all(dim=1)
            took = ei.to(torch.int64)
            steps += int(took.sum().item()) - m          # provisional: corrected below for the last pass of each ray
            alive = inside & (took >= iters - 1)
            k_sum += k_ms; c_sum += c_ms
            line.append("pass %d: %d rays, kernel %.3f ms (call %.3f)" % (p, m, k_ms, c_ms))
            idx = alive.nonzero().squeeze(1)
            m = int(idx.numel())
            if m == 0:
                break
            cur_sp = ep.view(torch.int32).view(-1, 3)[idx].contiguous().view(-1)
            cur_sd = ed.view(-1, 3)[idx].contiguous().view(-1)
            left -= iters - 1
            p += 1
        print("S = %d: kernel sum %.3f ms, call sum %.3f ms;  " % (S, k_sum, c_sum) + "; ".join(line))
    eng.close()


if __name__ == "__main__":
    main()
