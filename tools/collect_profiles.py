#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of one tools/profile.sh run from gpurun_out/ into profiles/ (tracked).

    tools/collect_profiles.py <tag under gpurun_out/prof_*> <round name, e.g. r01> <bench config number>

Also (re)writes profiles/traffic.json, which bench.py reads for roofline.traffic:
HBM bytes per launch = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024.  The factor 2 is the gfx950
FETCH_SIZE correction from MI355X_MICROARCH.md (HBM section); it is re-calibrated in the same run
on k_interleave_f32, whose read volume is known exactly (4 arrays x nvox x 4 B)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, rnd, cfg = sys.argv[1], sys.argv[2], int(sys.argv[3])
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    trace = json.load(open(os.path.join(src, "trace_summary.json")))
    pmc = {}
    for name in ("fetch", "write", "sq1", "sq2", "tcc", "tcp"):
        f = os.path.join(src, "pmc_%s_summary.json" % name)
        if os.path.exists(f):
            for kern, counters in json.load(open(f))["counters"].items():
                pmc.setdefault(kern, {}).update({k: v["per_dispatch"] for k, v in counters.items()})
    out = {"source": "rocprofv3 --kernel-trace --stats (trace) and separate --pmc passes, tools/profile.sh %s --config %d" % (tag, cfg),
           "kernel_trace": trace["kernels"], "kernel_stats_rows": trace["stats_rows"], "pmc_per_dispatch": pmc}
    json.dump(out, open(os.path.join(dst, "%s_rocprof_config%d.json" % (rnd, cfg)), "w"), indent=1)
    # human-readable stats table
    with open(os.path.join(dst, "%s_kernel_stats_config%d.txt" % (rnd, cfg)), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --cpu-seconds 0 --steps 3 --warmup 1 --config %d\n" % cfg)
        f.write("%-72s %6s %14s %14s %14s\n" % ("kernel", "calls", "avg_ms", "min_ms", "max_ms"))
        for k, v in sorted(trace["kernels"].items(), key=lambda kv: -kv[1]["total_ns"]):
            f.write("%-72s %6d %14.4f %14.4f %14.4f\n" % (k[:72], v["launches"], v["avg_ns"] / 1e6, v["min_ns"] / 1e6, v["max_ns"] / 1e6))
    march = [k for k in pmc if "k_march" in k]
    inter = [k for k in pmc if "k_interleave" in k]
    tj_path = os.path.join(dst, "traffic.json")
    tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
    if march and "FETCH_SIZE" in pmc[march[0]]:
        m = pmc[march[0]]
        entry = {"kernel": march[0], "FETCH_SIZE_KB": m["FETCH_SIZE"], "WRITE_SIZE_KB": m.get("WRITE_SIZE"),
                 "hbm_bytes_per_launch": m["FETCH_SIZE"] * 1024 * 2 + (m.get("WRITE_SIZE") or 0) * 1024,
                 "correction": "FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE counts half the bytes)",
                 "round": rnd}
        if inter and "FETCH_SIZE" in pmc[inter[0]]:
            entry["calibration"] = {"kernel": inter[0], "FETCH_SIZE_KB": pmc[inter[0]]["FETCH_SIZE"],
                                    "WRITE_SIZE_KB": pmc[inter[0]].get("WRITE_SIZE"),
                                    "note": "known traffic: reads = writes = field_bytes; FETCH_SIZE*1024/field_bytes should be 0.5"}
        tj["config%d" % cfg] = entry
        json.dump(tj, open(tj_path, "w"), indent=1)
    print("profiles written for", rnd, "config", cfg, "->", dst)


if __name__ == "__main__":
    main()
