#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of one tools/profile.sh run from gpurun_out/ into profiles/ (tracked) and regenerate
every number bench.py's roofline object quotes.

    tools/collect_profiles.py <tag under gpurun_out/prof_*> <round name, e.g. r02> <bench config number>

Writes  profiles/<round>_rocprof_config<N>.json      all passes, per kernel (trace + counters per dispatch)
        profiles/<round>_kernel_stats_config<N>.txt  the --kernel-trace --stats table, human readable
        profiles/roofline_inputs.json                per config: the march kernel's counters per launch that bench.py turns
                                                     into roofline fractions with the kernel time and clock it measures live

Roofline inputs and where they come from (all per launch of the march kernel):
  valu_insts          SQ_INSTS_VALU                      wave-instructions issued to the vector ALUs
  lds_insts           SQ_INSTS_LDS                       wave-instructions issued to the LDS (ds_read_b128 = 1 KiB each)
  lds_bank_conflict   SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  hbm_bytes           FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024   (MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE counts half the
                      bytes of a wide coalesced read; re-calibrated in the same run on a kernel of known traffic when the trace
                      holds one: k_interleave_f32 reads = writes = field bytes)
  ray_steps           sum of end_iteration of the profiled launch (from the bench JSON line of the trace pass): bench.py only uses
                      the entry when its own launch has exactly this many ray-steps (same workload)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json_line(path):
    try:
        for line in reversed(open(path).read().strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
    except (OSError, ValueError):
        pass
    return None


def main():
    tag, rnd, cfg = sys.argv[1], sys.argv[2], int(sys.argv[3])
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    trace = json.load(open(os.path.join(src, "trace_summary.json")))
    pmc = {}
    for name in ("fetch", "write", "sq1", "sq2", "tcc", "tcp", "grbm"):
        f = os.path.join(src, "pmc_%s_summary.json" % name)
        if os.path.exists(f):
            for kern, counters in json.load(open(f))["counters"].items():
                pmc.setdefault(kern, {}).update({k: v["per_dispatch"] for k, v in counters.items()})
    prof_name = "%s_rocprof_config%d.json" % (rnd, cfg)
    bench_line = last_json_line(os.path.join(src, "trace.json"))
    out = {"source": "rocprofv3 --kernel-trace --stats (trace) and separate --pmc passes, tools/profile.sh %s --config %d" % (tag, cfg),
           "bench_line_of_trace_pass": bench_line,
           "kernel_trace": trace["kernels"], "kernel_stats_rows": trace["stats_rows"], "pmc_per_dispatch": pmc}
    json.dump(out, open(os.path.join(dst, prof_name), "w"), indent=1)
    with open(os.path.join(dst, "%s_kernel_stats_config%d.txt" % (rnd, cfg)), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --cpu-seconds 0 --steps 4 --warmup 5 --config %d\n" % cfg)
        f.write("%-72s %6s %14s %14s %14s\n" % ("kernel", "calls", "avg_ms", "min_ms", "max_ms"))
        for k, v in sorted(trace["kernels"].items(), key=lambda kv: -kv[1]["total_ns"]):
            f.write("%-72s %6d %14.4f %14.4f %14.4f\n" % (k[:72], v["launches"], v["avg_ns"] / 1e6, v["min_ns"] / 1e6, v["max_ns"] / 1e6))
    # the march kernel of the steady state: the schedule the engine settled on has by far the most launches (the other
    # schedule's kernel appears only in the trial launches of the warm-up)
    march = sorted([k for k in pmc if "k_march" in k], key=lambda k: -trace["kernels"].get(k, {}).get("launches", 0))
    inter = [k for k in pmc if "k_interleave" in k]
    rj_path = os.path.join(dst, "roofline_inputs.json")
    rj = json.load(open(rj_path)) if os.path.exists(rj_path) else {}
    if march and "SQ_INSTS_VALU" in pmc[march[0]]:
        m = pmc[march[0]]
        # the schedule is the kernel's SCHED template argument (0 / 1 / 2 = vrt_options.schedule 1 / 2 / 3; the last argument until
        # the FAST flag followed it): bench.py uses the counters only when the launch it timed ran the same one (vrt_engine_last_schedule)
        import re
        msch = re.search(r",\s*(\d)\s*(?:,\s*(?:true|false)\s*)?>", march[0])
        entry = {"kernel": march[0], "round": rnd, "profile": "profiles/" + prof_name,
                 "schedule": (int(msch.group(1)) + 1) if msch else None,
                 "ray_steps": (bench_line or {}).get("config", {}).get("ray_steps_per_gpu"),
                 "kernel_ms_profiled": trace["kernels"].get(march[0], {}).get("avg_ns", 0) / 1e6,
                 "valu_insts": m.get("SQ_INSTS_VALU"), "salu_insts": m.get("SQ_INSTS_SALU"), "lds_insts": m.get("SQ_INSTS_LDS"),
                 "lds_bank_conflict_share": (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]) if m.get("SQ_LDS_IDX_ACTIVE") else None,
                 "lane_utilisation": (m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"] / 64.0) if m.get("SQ_ACTIVE_INST_VALU") and m.get("SQ_THREAD_CYCLES_VALU") else None,
                 "FETCH_SIZE_KB": m.get("FETCH_SIZE"), "WRITE_SIZE_KB": m.get("WRITE_SIZE"),
                 "hbm_bytes": (m["FETCH_SIZE"] * 1024 * 2 + (m.get("WRITE_SIZE") or 0) * 1024) if m.get("FETCH_SIZE") else None,
                 "hbm_bytes_formula": "FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE counts half the bytes of a wide read)",
                 "GRBM_GUI_ACTIVE": m.get("GRBM_GUI_ACTIVE"),
                 "counters": "SQ_INSTS_VALU, SQ_INSTS_LDS, SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE, FETCH_SIZE, WRITE_SIZE (per dispatch, separate passes)"}
        if inter and "FETCH_SIZE" in pmc[inter[0]]:
            entry["fetch_calibration"] = {"kernel": inter[0], "FETCH_SIZE_KB": pmc[inter[0]]["FETCH_SIZE"],
                                          "WRITE_SIZE_KB": pmc[inter[0]].get("WRITE_SIZE"),
                                          "note": "known traffic: reads = writes = field bytes, so FETCH_SIZE*1024/WRITE_SIZE*1024 should be 0.5"}
        rj["config%d" % cfg] = entry
        json.dump(rj, open(rj_path, "w"), indent=1)
    print("profiles written for", rnd, "config", cfg, "->", dst)


if __name__ == "__main__":
    main()
