#!/usr/bin/env python3
"""Where do the slab kernel's wavefronts spend their cycles?  (development aid; needs tools/build_debug.sh)

    VRT_LIB=$PWD/build/ab/libvrt_phases.so VRT_DEBUG_STATS=1 python3 tools/phases.py [--config 3]

Prints, for one march launch of a bench config: the clock the kernel ran at (s_memtime / s_memrealtime), the share of
wavefront lifetime spent planning / staging / marching out of LDS / marching out of global memory, and shader cycles
per wave-step in the two marching modes (wall cycles of the wavefront, i.e. including the time its SIMD gave to the
other resident wavefronts)."""
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
    ap.add_argument("--lds-bytes", type=int, default=0)
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
    opts = capi.default_options(sort_rays=args.sort, slab_lds_bytes=args.lds_bytes)
    for _ in range(3):
        eng.trace_device(capi.VRT_F32, n, d_sp.data_ptr(), d_sd.data_ptr(), o_ep.data_ptr(), o_ed.data_ptr(),
                         o_ei.data_ptr(), o_rl.data_ptr(), [1.0, 1.0, 1.0], 100000, options=opts)
    torch.cuda.synchronize()
    ms = eng.last_march_ms()
    raw = eng.debug_raw(64).astype(np.float64)
    ph = raw[16:21]
    waves, cyc, ticks = raw[24], raw[25], raw[26]
    lds, dr, rounds = raw[27], raw[28], raw[29]
    clock = cyc / max(ticks, 1) * 100e6
    names = ("plan", "stage", "march-lds", "march-global", "other")
    print("config %d: kernel %.3f ms, %d wavefronts, clock %.3f GHz (s_memtime/s_memrealtime)" % (args.config, ms, waves, clock / 1e9))
    print("wavefront lifetime: mean %.3f ms (%.0f %% of the launch)" % (cyc / waves / clock * 1e3, 100 * cyc / waves / clock * 1e3 / ms))
    for nme, v in zip(names, ph):
        print("   %-13s %5.1f %%" % (nme, 100 * v / ph.sum()))
    print("wave-steps: lds %.4g  global %.4g (%.1f %%)  rounds %.4g  (steps per staged round %.1f)"
          % (lds, dr, 100 * dr / max(lds + dr, 1), rounds, lds / max(rounds, 1)))
    print("wall cycles per wave-step: lds %.0f  global %.0f;  per round: plan %.0f stage %.0f" %
          (ph[2] / max(lds, 1), ph[3] / max(dr, 1), ph[0] / max(rounds, 1), ph[1] / max(rounds, 1)))
    print("scheduler words: tickets %d, groups finished %d, error %d" % (raw[60], raw[61], raw[62]))
    if raw[30] > 0:
        print("time-sliced schedule: tickets %d (wasted on finished groups %d), tickets that had to wait %d, polls %d" % (raw[30], raw[31], raw[33], raw[32]))
    simd_cycles = ms * 1e-3 * clock * eng_simds()
    print("SIMD-cycles of the launch per wave-step: %.0f (1024 SIMDs x kernel time x clock / wave-steps)" % (simd_cycles / (lds + dr)))


def eng_simds():
    return torch.cuda.get_device_properties(0).multi_processor_count * 4


if __name__ == "__main__":
    main()
