"""CPU: the checker itself under AddressSanitizer + UndefinedBehaviorSanitizer (the reference runs its unit
tests under valgrind, Makefile:99-101; sanitizers are this repository's equivalent -- and they belong on the CPU build:
GPU sanitizers are not available on the MI355X pool).

  * oracle/vrt_oracle.c, every entry point, through oracle/sanitize_main.c: no sanitizer report, and the sanitized
    build's results equal the normal build's bits;
  * oracle/ref_driver.cpp + the reference's own CPU sources (compiled where they lie, only where /root/reference exists):
    AddressSanitizer must stay silent; what UndefinedBehaviorSanitizer says about the reference's own code is recorded in
    the test's output, not asserted (it is the reference's behaviour, which the oracle has to reproduce either way)."""
import os
import subprocess

import numpy as np
import pytest

import oracle
from volumeraytracer_amd import workloads as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=23", UBSAN_OPTIONS="print_stacktrace=1",
           OMP_NUM_THREADS="4")


def _case(tmp, N=24, nrays=1500):
    rng = np.random.RandomState(7)
    ior_f = W.ior_volume(N, seed=3, dtype=np.float32).reshape(-1)
    ior_u = W.ior_volume(N, seed=3, dtype=np.uint32).reshape(-1)
    tr = np.full(N ** 3, 0xFFFFFFFF, np.uint32)
    tr[rng.rand(N ** 3) < 0.02] = 12345
    pv = 1.0 + rng.rand(nrays, 3) * (N - 2.001)
    sp = np.round(pv * 65536).astype(np.uint32).reshape(-1)
    d = rng.randn(nrays, 3)
    d /= np.linalg.norm(d, axis=1)[:, None]
    sd_f = d.astype(np.float32).reshape(-1)
    sd_i = np.round(d * 256).astype(np.int16).reshape(-1)
    for name, a in (("ior.f32", ior_f), ("ior.u32", ior_u), ("tr.u32", tr), ("sp.u32", sp), ("sd.f32", sd_f), ("sd.i16", sd_i)):
        a.tofile(os.path.join(tmp, name))
    return N, nrays, ior_f, ior_u, tr, sp, sd_f, sd_i


def test_restatement_under_asan_ubsan(tmp_path):
    tmp = str(tmp_path)
    exe = os.path.join(tmp, "sanitize_main")
    cmd = ["gcc", "-O1", "-fopenmp", "-ffp-contract=off", "-mfma", "-mavx2", "-std=c11", "-Wall"] + SAN + \
          ["-fno-sanitize-recover=undefined", "-I", os.path.join(ROOT, "oracle"), "-o", exe,
           os.path.join(ROOT, "oracle", "sanitize_main.c"), "-lm"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    N, nrays, ior_f, ior_u, tr, sp, sd_f, sd_i = _case(tmp)
    r = subprocess.run([exe, tmp, str(N), str(nrays), "100000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=ENV,
                       timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    assert "sanitize_main ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-3000:]
    # the sanitized build computes what the normal build computes
    for tag, ior, sd, ddt in (("f32", ior_f, sd_f, np.float32), ("i16", ior_u, sd_i, np.int16)):
        want = oracle.port.scene_trace([N] * 3, ior, tr, sp, sd, [1, 1, 1], 100000)
        ep = np.fromfile(os.path.join(tmp, "ep_%s.u32" % tag), np.uint32)
        ed = np.fromfile(os.path.join(tmp, "ed_%s.%s" % (tag, tag)), ddt)
        ei = np.fromfile(os.path.join(tmp, "ei_%s.u32" % tag), np.uint32)
        # sanitize_main stops at the seam: the scene layer's +0x10000 epilogue is applied here (image_util.cpp:771)
        assert np.array_equal(ep + np.uint32(0x10000), want["end_position"]), tag
        assert np.array_equal(ed.view(np.uint8), want["end_direction"].view(np.uint8)), tag
        assert np.array_equal(ei, want["end_iteration"]), tag


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference sources exist in the build container only")
def test_reference_driver_under_asan(tmp_path):
    """oracle/ref_driver.cpp and the reference's CPU path, compiled with the committed recipe's flags plus the
    sanitizers (objects in a temporary directory; nothing of the reference is copied into the repository)."""
    tmp = str(tmp_path)
    ref = "/root/reference/src"
    flags = ["-D_FORCE_INLINES", "-O2", "-fopenmp", "-std=c++17", "-march=native", "-mfma", "-fPIC", "-DNDEBUG", "-DNCUDA", "-msse", "-msse2",
             "-I" + ref, "-w"] + SAN
    objs = []
    jobs = []
    for src in ("util.cpp", "serialize.cpp", "image_util.cpp", "io_util.cpp", "cuda_volume_raytracer.cu"):
        o = os.path.join(tmp, src.rsplit(".", 1)[0] + ".o")
        objs.append(o)
        lang = ["-x", "c++"] if src.endswith(".cu") else []
        jobs.append(subprocess.Popen(["g++"] + flags + lang + ["-c", os.path.join(ref, src), "-o", o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    o = os.path.join(tmp, "ref_driver.o")
    objs.append(o)
    jobs.append(subprocess.Popen(["g++"] + flags + ["-c", os.path.join(ROOT, "oracle", "ref_driver.cpp"), "-o", o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for j in jobs:
        out, _ = j.communicate()
        assert j.returncode == 0, out[-3000:]
    main_cpp = os.path.join(tmp, "ref_sanitize_main.cpp")
    open(main_cpp, "w").write(r'''
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
extern "C" {
int ref_scene_f32(const size_t *, const float *, const uint32_t *, size_t, size_t, const uint32_t *, const float *, uint32_t *, float *,
                  uint32_t *, uint32_t *, uint32_t *, const float *, uint32_t, int);
int ref_scene_i16(const size_t *, const uint32_t *, const uint32_t *, size_t, size_t, const uint32_t *, const int16_t *, uint32_t *, int16_t *,
                  uint32_t *, uint32_t *, uint32_t *, const float *, uint32_t, int);
const char *ref_last_error(void);
}
template <typename T> static std::vector<T> load(const char *dir, const char *name, size_t n)
{
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE *f = fopen(path, "rb");
    std::vector<T> v(n);
    if (!f || fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
    fclose(f);
    return v;
}
template <typename T> static void save(const char *dir, const char *name, const std::vector<T> &v)
{
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE *f = fopen(path, "wb");
    if (!f || fwrite(v.data(), sizeof(T), v.size(), f) != v.size()) exit(2);
    fclose(f);
}
int main(int, char **argv)
{
    const char *dir = argv[1];
    const size_t N = atol(argv[2]), n = atol(argv[3]);
    const size_t b[3] = {N, N, N}, nvox = N * N * N;
    const float inv[3] = {1, 1, 1};
    auto tr = load<uint32_t>(dir, "tr.u32", nvox);
    auto sp = load<uint32_t>(dir, "sp.u32", 3 * n);
    std::vector<uint32_t> ep(3 * n), ei(n), rl(n);
    {
        auto ior = load<float>(dir, "ior.f32", nvox);
        auto sd = load<float>(dir, "sd.f32", 3 * n);
        std::vector<float> ed(3 * n);
        if (ref_scene_f32(b, ior.data(), tr.data(), nvox, n, sp.data(), sd.data(), ep.data(), ed.data(), ei.data(), rl.data(), nullptr, inv, 100000, 4)) { fprintf(stderr, "%s\n", ref_last_error()); return 3; }
        save(dir, "ref_ei_f32.u32", ei);
        save(dir, "ref_ep_f32.u32", ep);
    }
    {
        auto ior = load<uint32_t>(dir, "ior.u32", nvox);
        auto sd = load<int16_t>(dir, "sd.i16", 3 * n);
        std::vector<int16_t> ed(3 * n);
        std::vector<uint32_t> path((size_t)50 * 3 * 32);
        if (ref_scene_i16(b, ior.data(), tr.data(), nvox, n, sp.data(), sd.data(), ep.data(), ed.data(), ei.data(), rl.data(), nullptr, inv, 100000, 4)) { fprintf(stderr, "%s\n", ref_last_error()); return 3; }
        save(dir, "ref_ei_i16.u32", ei);
        save(dir, "ref_ep_i16.u32", ep);
        if (ref_scene_i16(b, ior.data(), tr.data(), nvox, 32, sp.data(), sd.data(), ep.data(), ed.data(), ei.data(), rl.data(), path.data(), inv, 50, 4)) { fprintf(stderr, "%s\n", ref_last_error()); return 3; }
    }
    printf("ref_sanitize_main ok\n");
    return 0;
}
''')
    exe = os.path.join(tmp, "ref_sanitize_main")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-fopenmp"] + SAN + ["-o", exe, main_cpp] + objs, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    # N - 2 = 20: the reference allocates its interleaved field with aligned_alloc(256, bytes) (cuda_volume_raytracer.cu:439,
    # 660) whatever the byte count; C11 wants a multiple of the alignment and AddressSanitizer aborts otherwise
    # ("invalid-aligned-alloc-alignment" -- a finding about the reference, harmless with glibc).  20^3 voxels of 8 or 16 bytes
    # are a multiple of 256, so the run gets past that line and everything after it is checked.
    N, nrays, ior_f, ior_u, tr, sp, sd_f, sd_i = _case(tmp, N=22, nrays=600)
    env = dict(ENV, ASAN_OPTIONS="detect_leaks=0:exitcode=23")      # the reference keeps a few allocations for its lifetime: leaks are not asserted
    r = subprocess.run([exe, tmp, str(N), str(nrays)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900)
    ub = r.stderr.count("runtime error")
    print("UndefinedBehaviorSanitizer reports inside the reference's own code: %d" % ub)
    if ub:
        print("\n".join(l for l in r.stderr.splitlines() if "runtime error" in l)[:2000])
    assert "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0 and "ref_sanitize_main ok" in r.stdout, (r.returncode, r.stderr[-2000:])
    # and it computes what the restatement computes -- up to the contraction pattern: the sanitizers' instrumentation changes
    # which multiply-adds g++ fuses, so this build is not the bit-exact one (tests/test_oracle_vs_reference.py pins the bits
    # against the uninstrumented build); what must hold here is north_star's tolerance
    for tag, ior, sd in (("f32", ior_f, sd_f), ("i16", ior_u, sd_i)):
        want = oracle.port.scene_trace([N] * 3, ior, tr, sp, sd, [1, 1, 1], 100000)
        ei = np.fromfile(os.path.join(tmp, "ref_ei_%s.u32" % tag), np.uint32).astype(np.int64)
        ep = np.fromfile(os.path.join(tmp, "ref_ep_%s.u32" % tag), np.uint32).astype(np.int64)
        assert np.abs(ei - want["end_iteration"].astype(np.int64)).max() <= 1, tag
        assert np.mean(ei == want["end_iteration"]) > 0.99, tag
        same_steps = np.repeat(ei == want["end_iteration"], 3)
        assert np.abs(ep - want["end_position"].astype(np.int64))[same_steps].max() <= 1e-5 * (N << 16) + 2, tag
