"""Builds the in-tree native libraries (hipcc cross-compiles gfx950 without a GPU).

    python -m volumeraytracer_amd._build          # libvrt_hip.so (+ the oracle checker)
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libvrt_hip.so")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    # every fma in the march is written out; the compiler must not add or remove any
    # (bit parity with the reference's CPU build, see csrc/vrt_march.h)
    "-ffp-contract=off",
    # packed fp32 (v_pk_*) issues at half the rate of scalar fp32 on gfx950 and drags hazard s_nops in
    # (tools/ubench/valu_rate.hip): keep the march scalar
    "-fno-slp-vectorize",
    "-Wall", "-Wno-unused-function",
]


def _sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))] + \
           [os.path.join(ROOT, "include", "vrt.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in _sources())


def build_hip(force=False, verbose=False):
    """Compile csrc/vrt_engine.hip -> lib/libvrt_hip.so for gfx950."""
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", LIB + ".tmp", os.path.join(CSRC, "vrt_engine.hip")]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
