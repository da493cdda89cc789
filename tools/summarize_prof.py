#!/usr/bin/env python3
"""Reduce rocprofv3 CSV output to a compact per-kernel summary (engine kernels only).

    summarize_prof.py <prof_dir> <out_json>

Reads every *_kernel_trace.csv / *_kernel_stats.csv / *_counter_collection.csv below <prof_dir>;
keeps the engine's kernels (k_march_*, k_tile_keys, k_interleave_*, rocprim radix sort) and
aggregates: launches, total/avg/min/max duration (ns), and per-counter sums / per-launch averages."""
import csv
import glob
import json
import os
import sys

KEEP = ("k_march", "k_tile", "k_bounds_finish", "k_interleave", "radix", "onesweep", "rocprim", "k_wave_work", "k_ray_bounds", "k_scene_build",
        "k_ray_prologue", "k_field_absmax", "k_absmax_reduce", "k_wave_sort")


def short(name):
    for k in ("k_march_direct", "k_march_slab", "k_tile_keys", "k_interleave_f32", "k_interleave_i16", "k_scene_build", "k_ray_prologue",
              "k_field_absmax"):
        if k in name:
            # keep template arguments readable
            i = name.find(k)
            return name[i:i + 80]
    return name[:80]


def main():
    root, out = sys.argv[1], sys.argv[2]
    res = {"kernels": {}, "counters": {}, "stats_rows": []}
    for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if not any(k in name for k in KEEP):
                    continue
                d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                k = res["kernels"].setdefault(short(name), {"launches": 0, "total_ns": 0, "min_ns": None, "max_ns": 0,
                                                            "vgpr": row.get("VGPR_Count"), "sgpr": row.get("SGPR_Count"),
                                                            "lds": row.get("LDS_Block_Size"), "grid": row.get("Grid_Size"),
                                                            "wg": row.get("Workgroup_Size")})
                k["launches"] += 1
                k["total_ns"] += d
                k["min_ns"] = d if k["min_ns"] is None else min(k["min_ns"], d)
                k["max_ns"] = max(k["max_ns"], d)
    for k in res["kernels"].values():
        k["avg_ns"] = k["total_ns"] / max(k["launches"], 1)
    for f in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if any(k in row.get("Name", "") for k in KEEP):
                    row["Name"] = short(row["Name"])
                    res["stats_rows"].append(row)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if not any(k in name for k in KEEP):
                    continue
                c = res["counters"].setdefault(short(name), {})
                e = c.setdefault(row["Counter_Name"], {"sum": 0.0, "dispatches": 0})
                e["sum"] += float(row["Counter_Value"])
                e["dispatches"] += 1
    for c in res["counters"].values():
        for e in c.values():
            e["per_dispatch"] = e["sum"] / max(e["dispatches"], 1)
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out, {k: (v["launches"], round(v["avg_ns"] / 1e6, 3)) for k, v in res["kernels"].items()})


if __name__ == "__main__":
    main()
