"""CPU: the C-ABI library loads, exports every symbol include/vrt.h declares, and fails loudly
(no CPU fallback) when there is no GPU.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

from volumeraytracer_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "vrt.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vrt_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_exported():
    lib = capi.load()
    names = _declared_symbols()
    assert "vrt_create" in names and "vrt_trace" in names and "vrt_destroy" in names
    for n in names:
        assert hasattr(lib, n), "libvrt_hip.so does not export %s" % n
    for n in capi.ABI_SYMBOLS:
        assert n in names


def test_abi_version_and_options_layout():
    lib = capi.load()
    assert lib.vrt_abi_version() == 3
    o = capi.default_options()
    # struct Options defaults, /root/reference/src/types.h:90
    assert (o.loglevel, o.minimum_gpu, o.write_instance, o.max_cpu) == (0, 0x80, 0, 256)
    assert o.device == -1 and o.kernel == capi.KERNEL_AUTO
    assert ctypes.sizeof(capi.Options) == 72
    assert (o.devices, o.host_lanes, o.fast_math) == (0, 0, 0)           # extensions default off: one device, auto chunking


def test_band_cut():
    """The contiguous cut multi-device handles and the chunk pipeline use (pure host arithmetic): covers all rays once, in order,
    whole groups of 64 except at the very end, sizes within one group of each other."""
    for n in (0, 1, 63, 64, 65, 1000, 4096, 100003, 1 << 20):
        for parts in (1, 2, 3, 4, 7, 8):
            b = capi.band_cut(n, parts)
            assert len(b) == parts + 1 and b[0] == 0 and b[-1] == n
            assert all(b[k] <= b[k + 1] for k in range(parts))
            assert all(b[k] % 64 == 0 for k in range(parts) if b[k] < n)
            sizes = [b[k + 1] - b[k] for k in range(parts)]
            full = [x for x in sizes[:-1] if x > 0] or [0]
            assert max(full) - min(full) <= 64


def test_argument_validation_without_gpu():
    # argument errors are reported before any device is touched
    z = np.zeros(8, np.float32)
    tr = np.zeros(8, np.uint32)
    with pytest.raises(capi.VrtError) as e:
        capi.Engine.from_host([2, 4], [z, z], tr)            # 2-D: "Illegal dimension"
    assert e.value.code == -2 and "Illegal dimension" in str(e.value)
    with pytest.raises(capi.VrtError) as e:
        capi.Engine.from_host([2, 2, 3], [z, z, z], tr)      # nvox mismatch
    assert e.value.code == -1


def test_no_silent_cpu_fallback():
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    z = np.zeros(8, np.float32)
    with pytest.raises(capi.VrtError) as e:
        capi.Engine.from_host([2, 2, 2], [z, z, z], np.zeros(8, np.uint32))
    assert e.value.code == -5          # VRT_ERR_NO_DEVICE


def test_product_does_not_import_oracle():
    """The product path must not route through the checker."""
    pkg = os.path.join(ROOT, "volumeraytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, fn), errors="replace").read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), fn
                assert "vrt_oracle" not in src, fn
