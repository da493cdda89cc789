#!/usr/bin/env python3
"""Compile the engine for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line per kernel
(VGPRs, scratch bytes per lane, occupancy): the evidence that no shipped instantiation spills.

    tools/resource_usage.py [extra hipcc flags ...] > profiles/rNN_kernel_resource_usage.txt
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from volumeraytracer_amd import _build  # noqa: E402


def main():
    flags = [f for f in _build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + sys.argv[1:] + [
        "--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null",
        os.path.join(_build.CSRC, "vrt_engine.hip")]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        sys.exit(r.stderr[-4000:])
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
    print("# " + " ".join(cmd))
    print("%-86s %5s %5s %8s %4s" % ("kernel (demangled)", "VGPR", "SGPR", "scratch", "occ"))
    rows = []
    for b in blocks:
        name = b.split()[0]
        def g(k):
            m = re.search(k + r": (\S+)", b)
            return m.group(1) if m else "?"
        rows.append((name, g("VGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), stdout=subprocess.PIPE, text=True).stdout.split("\n")
    worst = 0
    for (n, v, s, sc, oc), dn in zip(rows, names):
        dn = re.sub(r"\(.*", "", dn).replace("void ", "")
        if dn.startswith("vrt::") or dn.startswith("(anonymous") or "k_" in dn:
            if "rocprim" in dn:
                continue
            print("%-86s %5s %5s %8s %4s" % (dn[:86], v, s, sc, oc))
            worst = max(worst, int(sc) if sc.isdigit() else 0)
    print("# largest scratch of an engine kernel: %d bytes/lane" % worst)


if __name__ == "__main__":
    main()
