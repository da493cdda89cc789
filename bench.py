#!/usr/bin/env python3
"""bench.py -- headline benchmark of the trace_rays hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of rays: (tile-order) sort + march kernel on
every rank and, for N > 1, the RCCL gather of the per-ray outputs to rank 0.  Inputs (field and
rays) are resident in HBM before the timed region.  One process per GPU; the field is replicated,
rays are sharded as contiguous row bands of the ray lattice (weak scaling: every rank marches the
same number of rays).

Prints ONE JSON line on rank 0 (see DESIGN.md §Measurement for the definitions):
  value            = sum(end_iteration over all rays of all ranks) * K / wall time   [ray-steps/s]
  roofline         = the march kernel against the resource that binds it.  The corners of the gather come out of LDS,
                     so HBM does not bind this kernel; vector-ALU issue does.  frac = achieved / peak <= 1 with
                       achieved = SQ_INSTS_VALU per launch (profiles/roofline_inputs.json, rocprofv3 counter pass of the
                                  same workload) / HIP-event duration of the kernel measured live in this run
                       peak     = SIMDs x shader clock / 2 (a wave64 instruction issues over two cycles of a SIMD-32),
                                  the clock measured live inside the kernel (s_memtime / s_memrealtime)
                     next to it: roofline.lds, roofline.hbm_measured (counter bytes) and roofline.hbm_algorithmic, the
                     contract's 128 B (fp32 field; 64 B int16) per ray-step figure, which exceeds the HBM peak because
                     the bytes are served from LDS -- reported, but not a bound
  cpu_baseline     = the oracle (oracle/vrt_oracle.c, "port") on a bounded sample of the same rays, same field, with the
                     GPU's results checked against it bit for bit; cpu_baseline.reference = the REAL reference CPU path
                     (oracle/_ref/libvrt_ref.so, built from /root/reference/src in the build container) timed on a smaller
                     sample in a child process, when that library is present and loads on this host
  N > 1            = rank 0 checks an evenly strided sample of the GATHERED records against the oracle
                     ("gathered_matches_oracle"; a mismatch exits non-zero) and reports every rank's band, kernel time
                     and gather time ("ranks")
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from volumeraytracer_amd import capi, distributed as D, workloads as W  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)

CONFIGS = {
    # BASELINE.json configs[1..4]; config 1 is the CPU-only plumbing case (tests/test_oracle_golden.py)
    2: dict(N=256, R=512, gen="perspective", field="f32", name="256^3 fp32 field, 512x512 perspective rays"),
    3: dict(N=512, R=1024, gen="perspective", field="f32", name="512^3 fp32 field, 1024x1024 perspective rays"),
    4: dict(N=512, R=2048, gen="perspective", field="f32", name="512^3 fp32 field, 2048x2048 perspective rays"),
    5: dict(N=1024, R=1024, gen="ortho", field="i16", lut=True,
            name="1024^3 uint8+LUT volume expanded to the int16 field, 1024x1024 ortho rays, opaque cores"),
    # diagnostic variants (not BASELINE configs)
    6: dict(N=512, R=1024, gen="ortho", field="f32", name="512^3 fp32 field, 1024x1024 orthographic rays (diagnostic)"),
    7: dict(N=512, R=1448, gen="perspective", field="f32", name="512^3 fp32 field, 1448x1448 perspective rays (diagnostic)"),
    8: dict(N=512, R=2048, gen="perspective", field="i16", name="512^3 int16 field, 2048x2048 perspective rays (diagnostic: dense rays on an int16 field)"),
    # small smoke-size config for quick checks
    0: dict(N=96, R=256, gen="perspective", field="f32", name="96^3 fp32 field, 256x256 perspective rays (smoke)"),
}


def synth_field_device(N, field, dev, opaque=False, seed=1234, nblobs=4, amp=0.5):
    """workloads.seam_field evaluated on the GPU with torch (float64), slab by slab.
    Returns (bounds, [dx, dy, dz] device tensors, transl_cropped device tensor (int32 bits))."""
    B = N - 2
    c = torch.tensor(W.blob_centres(N, seed, nblobs), dtype=torch.float64, device=dev)
    a = 16.0 / (N * N)
    tdt = torch.float32 if field == "f32" else torch.int16
    d = [torch.empty(B * B * B, dtype=tdt, device=dev) for _ in range(3)]
    tr = torch.full((B * B * B,), -1, dtype=torch.int32, device=dev)     # 0xFFFFFFFF
    y = torch.arange(1, N - 1, dtype=torch.float64, device=dev)[None, :, None]
    z = torch.arange(1, N - 1, dtype=torch.float64, device=dev)[None, None, :]
    CH = 16
    for x0 in range(0, B, CH):
        x1 = min(B, x0 + CH)
        x = torch.arange(x0 + 1, x1 + 1, dtype=torch.float64, device=dev)[:, None, None]
        n = torch.ones((x1 - x0, B, B), dtype=torch.float64, device=dev)
        g = [torch.zeros_like(n) for _ in range(3)]
        near = torch.zeros_like(n, dtype=torch.bool)
        for k in range(nblobs):
            dd = [x - c[k, 0], y - c[k, 1], z - c[k, 2]]
            r2 = dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]
            e = amp * torch.exp(-r2 * a)
            n = n + e
            for j in range(3):
                g[j] = g[j] - 2 * a * dd[j] * e
            if opaque:
                near |= r2 < (N / 10.0) ** 2
        sl = slice(x0 * B * B, x1 * B * B)
        for j in range(3):
            v = g[j] / n * W.DIFF_PER_DLOGN
            d[j][sl] = (v.to(torch.float32) if field == "f32" else torch.round(v).to(torch.int16)).reshape(-1)
        if opaque:
            tr[sl] = torch.where(near, torch.zeros((), dtype=torch.int32, device=dev),
                                 torch.full((), -1, dtype=torch.int32, device=dev)).reshape(-1)
    return [B, B, B], d, tr


def synth_density_device(N, dev, seed=1234, nblobs=4):
    """workloads.uint8_density_and_lut evaluated on the GPU: uint8 density [N^3] (device tensor)."""
    c = torch.tensor(W.blob_centres(N, seed, nblobs), dtype=torch.float64, device=dev)
    a = 16.0 / (N * N)
    dens = torch.empty(N * N * N, dtype=torch.uint8, device=dev)
    y = torch.arange(N, dtype=torch.float64, device=dev)[None, :, None]
    z = torch.arange(N, dtype=torch.float64, device=dev)[None, None, :]
    CH = 8
    for x0 in range(0, N, CH):
        x1 = min(N, x0 + CH)
        x = torch.arange(x0, x1, dtype=torch.float64, device=dev)[:, None, None]
        n = torch.zeros((x1 - x0, N, N), dtype=torch.float64, device=dev)
        for k in range(nblobs):
            n = n + 0.5 * torch.exp(-((x - c[k, 0]) ** 2 + (y - c[k, 1]) ** 2 + (z - c[k, 2]) ** 2) * a)
        dens[x0 * N * N:x1 * N * N] = torch.clamp(torch.round(n / (0.5 * nblobs) * 255.0 * 2.0), 0, 255).to(torch.uint8).reshape(-1)
    return dens


def lattice_rows(cfg, world, scaling="weak"):
    """Rows of the whole job's ray lattice (R columns each).  weak: every rank marches R rows, so the lattice grows to
    (R*world) x R; strong: the lattice BASELINE names, R x R, is cut into `world` row bands (BASELINE config 4:
    2048 x 2048 rays tiled across the GPUs of the node)."""
    return cfg["R"] * world if scaling == "weak" else cfg["R"]


def make_rays(cfg, rank, world, band=None, scaling="weak"):
    gen = W.perspective_rays if cfg["gen"] == "perspective" else W.ortho_rays
    rows = lattice_rows(cfg, world, scaling)
    # rank r marches one contiguous band of rows (equal counts unless rebalanced)
    return gen(cfg["N"], cfg["R"], rows_total=rows, row_range=band if band is not None else D.row_band(rows, rank, world))


def build_engine(cfg, dev, log=lambda *a: None):
    """Field of a bench config resident on `dev` -> (engine, keep-alive objects, bounds, d, tr, lut_host).
    LUT configs (BASELINE config 5) go through the scene layer (vrt_scene_create_lut): uint8 density + 256-entry transfer
    function -> int16 field on the device; the march is then run on the scene's engine."""
    N = cfg["N"]
    t0 = time.time()
    if cfg.get("lut"):
        dens = synth_density_device(N, dev).cpu().numpy()
        dd = np.arange(256, dtype=np.float64)
        lut_ior = np.round((1.0 + dd / 510.0) * 65536).astype(np.uint32)
        lut_tr = np.where(dd < 160, 0xFFFFFFFF, 0).astype(np.uint32)
        scene = capi.Scene.from_lut([N, N, N], dens, lut_ior, lut_tr)
        eng = scene.engine()
        out = (eng, scene, [N - 2] * 3, None, None, (dens, lut_ior, lut_tr))     # lut_host: for the CPU baseline leg only
    else:
        bounds, d, tr = synth_field_device(N, cfg["field"], dev, opaque=cfg.get("opaque", False))
        torch.cuda.synchronize()
        ftype = capi.VRT_F32 if cfg["field"] == "f32" else capi.VRT_I16
        eng = capi.Engine.from_device(bounds, [x.data_ptr() for x in d], tr.data_ptr(), ftype, d[0].numel())
        out = (eng, None, bounds, d, tr, None)
    log("field %s built on device in %.1f s (%.2f GB resident)" % (cfg["name"], time.time() - t0, eng.field_bytes / 1e9))
    return out


def cpu_baseline(field_host, bounds, sp, sd, inv, iterations, gpu_out, budget_s, log):
    """Time the CPU checker on a bounded sample of this rank's rays (and verify the GPU on it)."""
    import oracle
    n = sp.size // 3
    sp3, sd3 = sp.reshape(-1, 3), sd.reshape(-1, 3)
    threads = oracle.port.num_threads()
    # calibrate on 16384 evenly spread rays (enough work for every thread)
    idx = np.linspace(0, n - 1, min(n, 16384)).astype(np.int64)
    t0 = time.time()
    out = oracle.port.trace(field_host, bounds, sp3[idx].reshape(-1), sd3[idx].reshape(-1), inv, iterations)
    dt = max(time.time() - t0, 1e-4)
    steps = int(out["end_iteration"].astype(np.int64).sum())
    rate = steps / dt
    mean_steps = steps / idx.size
    m = int(min(n, max(idx.size, budget_s * rate / max(mean_steps, 1.0))))
    idx = np.linspace(0, n - 1, m).astype(np.int64)
    t0 = time.time()
    out = oracle.port.trace(field_host, bounds, sp3[idx].reshape(-1), sd3[idx].reshape(-1), inv, iterations)
    dt = time.time() - t0
    steps = int(out["end_iteration"].astype(np.int64).sum())
    ok = (np.array_equal(out["end_position"].reshape(-1, 3), gpu_out["end_position"].reshape(-1, 3)[idx])
          and np.array_equal(out["end_iteration"], gpu_out["end_iteration"][idx])
          and np.array_equal(out["end_direction"].reshape(-1, 3).view(np.uint32),
                             gpu_out["end_direction"].reshape(-1, 3)[idx].view(np.uint32)))
    res = {"value": steps / dt, "unit": "ray-steps/s", "cores": threads, "kind": "port",
           "sample": "%d of %d rays (evenly strided), %d ray-steps, %.2f s, OpenMP %d threads" % (m, n, steps, dt, threads),
           "gpu_matches_bit_exact": bool(ok), "cpu_model": cpu_model()}
    # also the 1-thread rate, on a sample sized for about 2 s
    m1 = int(min(n, max(64, 2.0 * 1.8e7 / max(mean_steps, 1.0))))     # one core does about 1.8e7 ray-steps/s
    idx = np.linspace(0, n - 1, m1).astype(np.int64)
    t0 = time.time()
    out = oracle.port.trace(field_host, bounds, sp3[idx].reshape(-1), sd3[idx].reshape(-1), inv, iterations, nthreads=1)
    dt = max(time.time() - t0, 1e-6)
    res["one_thread"] = {"value": int(out["end_iteration"].astype(np.int64).sum()) / dt, "unit": "ray-steps/s",
                         "sample": "%d rays, %.2f s" % (m1, dt)}
    return res


def reference_leg_child(path):
    """Child process of reference_leg(): time the real reference build on the sample stored at `path` (npz), print JSON."""
    import oracle
    z = np.load(path)
    if not oracle.ref.available():
        print(json.dumps({"error": "oracle/_ref/libvrt_ref.so absent or not loadable"}))
        return
    b = [int(x) for x in z["bounds"]]
    # One call of the reference's seam = TraceRaysCu constructor (w channel + serial interleave of the whole field,
    # cuda_volume_raytracer.cu:654-669) + trace_rays_cu.  The constructor is one-time work per scene and not on the timed path
    # of either side, so it is measured by itself (a call with a single ray) and subtracted.
    t0 = time.time()
    oracle.ref.seam_trace(b, z["dx"], z["dy"], z["dz"], z["tc"], z["sp"][:3], z["sd"][:3], z["inv"], int(z["iterations"]))
    t_ctor = time.time() - t0
    t0 = time.time()
    out = oracle.ref.seam_trace(b, z["dx"], z["dy"], z["dz"], z["tc"], z["sp"], z["sd"], z["inv"], int(z["iterations"]))
    dt = time.time() - t0
    np.savez(path + ".out.npz", end_position=out["end_position"], end_direction=out["end_direction"], end_iteration=out["end_iteration"])
    print(json.dumps({"seconds": max(dt - t_ctor, 1e-3), "seconds_with_constructor": dt, "constructor_seconds": t_ctor,
                      "ray_steps": int(out["end_iteration"].astype(np.int64).sum()),
                      # the reference runs min(omp_get_max_threads(), Options::_max_cpu = 256) threads (cuda_volume_raytracer.cu:809)
                      "threads": min(oracle.port.num_threads(), 256)}))


def reference_leg(host, tc_host, bounds, sp, sd, inv, iterations, port_out_for, budget_s, log):
    """The real reference CPU path (oracle/_ref, compiled from the reference's own sources by oracle/Makefile in the
    build container) timed on this host.  It runs in a child process: the objects were built with the reference
    Makefile's -march=native on another CPU, and an illegal instruction must not take the benchmark down.  Returns the
    "reference" object for cpu_baseline, or None when the library is absent / does not run here."""
    so = os.path.join(ROOT, "oracle", "_ref", "libvrt_ref.so")
    if not os.path.exists(so):
        return None
    import subprocess
    import tempfile
    n = sp.size // 3
    m = int(min(n, 262144))          # 8 of the reference's 0x8000-ray chunks: enough for its per-chunk OpenMP loop to be timed
    idx = np.linspace(0, n - 1, m).astype(np.int64)
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    path = os.path.join(tmpdir, "vrt_ref_leg_%d.npz" % os.getpid())
    try:
        np.savez(path, bounds=np.asarray(bounds), dx=host[0], dy=host[1], dz=host[2], tc=tc_host, sp=sp.reshape(-1, 3)[idx].reshape(-1),
                 sd=sd.reshape(-1, 3)[idx].reshape(-1), inv=np.asarray(inv, np.float32), iterations=iterations)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--reference-leg", path], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=max(60.0, 20 * budget_s))
        if r.returncode != 0:
            log("reference leg: child exited with %d (%s)" % (r.returncode, r.stderr.strip()[-200:]))
            return {"ran": False, "note": "oracle/_ref/libvrt_ref.so does not run on this host (child exit %d)" % r.returncode}
        j = json.loads(r.stdout.strip().splitlines()[-1])
        if "error" in j:
            return {"ran": False, "note": j["error"]}
        got = np.load(path + ".out.npz")
        want = port_out_for(idx)
        same = (np.array_equal(got["end_position"], want["end_position"]) and np.array_equal(got["end_iteration"], want["end_iteration"])
                and np.array_equal(got["end_direction"].view(np.uint32), want["end_direction"].view(np.uint32)))
        return {"ran": True, "value": j["ray_steps"] / j["seconds"], "unit": "ray-steps/s", "cores": j["threads"], "kind": "reference",
                "sample": "%d of %d rays (evenly strided), %d ray-steps, %.2f s (trace_rays_cu only: %.2f s for the call minus %.2f s for the "
                          "constructor's serial field interleave, measured by a one-ray call), the reference's own OpenMP loop over 0x8000-ray chunks"
                          % (m, n, j["ray_steps"], j["seconds"], j["seconds_with_constructor"], j["constructor_seconds"]),
                "matches_port_bit_exact": bool(same)}
    except Exception as ex:                                        # never let the optional leg break the benchmark
        log("reference leg failed:", ex)
        return {"ran": False, "note": str(ex)[:200]}
    finally:
        for f in (path, path + ".out.npz"):
            try:
                os.remove(f)
            except OSError:
                pass


def roofline_object(args, cfg, dev, eng, ray_steps, kernel_ms, clocks, bytes_per_step, sched=None):
    """The march kernel against its binding resource (module docstring).  Counter-derived inputs come from
    profiles/roofline_inputs.json and are used only when that profile was taken on this very workload (same number of
    ray-steps per launch); time and clock are measured in this run."""
    t = kernel_ms * 1e-3
    props = torch.cuda.get_device_properties(dev)
    simds = props.multi_processor_count * 4
    clock = float(np.mean(clocks)) * 1e9 if clocks else None
    inp, src = None, None
    try:
        inp = json.load(open(args.roofline_inputs)).get("config%d" % args.config)
        src = os.path.relpath(args.roofline_inputs, ROOT)
    except Exception:
        inp = None
    # the counters depend on the schedule the launch used (park / resume, re-planning): only a profile of the same one counts
    sched_ok = bool(inp) and (inp.get("schedule") is None or sched is None or inp.get("schedule") == sched)
    same = bool(inp) and inp.get("ray_steps") == ray_steps and args.kernel == "auto" and args.lds_bytes == 0 and args.sort == 0 and sched_ok
    alg = ray_steps * bytes_per_step / t / 1e9
    out = {"bound": "valu-issue", "achieved": None, "peak": None, "unit": "Gwave-instr/s", "frac": None,
           "traffic": None, "traffic_source": None, "kernel_ms": kernel_ms, "clock_ghz": clock / 1e9 if clock else None,
           "clock_source": "s_memtime / s_memrealtime of one wavefront of the march kernel, this run",
           "simds": simds, "schedule_observed": sched,
           "hbm_algorithmic": {"achieved": alg, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / HBM_PEAK_GBS,
                               "bytes_per_ray_step": bytes_per_step,
                               "note": "SURVEY 8(d) figure; exceeds 1 because the gather is served from LDS, not HBM: not a bound"},
           "compulsory_bytes": eng.field_bytes, "measured_copy_GBps": device_copy_gbs(dev)}
    if clock:
        out["peak"] = simds * clock / 2.0 / 1e9
    if same:
        out["inputs"] = {"file": src, "profile": inp.get("profile"), "kernel": inp.get("kernel"), "round": inp.get("round"),
                         "counters": inp.get("counters")}
        if inp.get("valu_insts") and clock:
            out["achieved"] = inp["valu_insts"] / t / 1e9
            out["frac"] = out["achieved"] / out["peak"]
            out["valu"] = {"insts_per_launch": inp["valu_insts"], "insts_per_64_ray_steps": inp["valu_insts"] / (ray_steps / 64.0),
                           "lane_utilisation": inp.get("lane_utilisation"), "counter": "SQ_INSTS_VALU", "frac": out["frac"]}
        if inp.get("lds_insts") and clock:
            lds_bytes = inp["lds_insts"] * 1024.0                       # a wave-wide ds_read_b128 / ds_write_b128 moves 1 KiB
            peak = props.multi_processor_count * 256.0 * clock          # 256 B per clock per CU (MI355X_MICROARCH.md, LDS)
            out["lds"] = {"achieved": lds_bytes / t / 1e9, "peak": peak / 1e9, "unit": "GB/s", "frac": lds_bytes / t / peak,
                          "bank_conflict_share": inp.get("lds_bank_conflict_share"), "counter": "SQ_INSTS_LDS x 1 KiB"}
        if inp.get("hbm_bytes"):
            out["traffic"] = inp["hbm_bytes"]
            out["traffic_source"] = "%s: %s" % (inp.get("profile"), inp.get("hbm_bytes_formula"))
            out["hbm_measured"] = {"achieved": inp["hbm_bytes"] / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": inp["hbm_bytes"] / t / 1e9 / HBM_PEAK_GBS, "counter": "FETCH_SIZE, WRITE_SIZE"}
    elif inp and inp.get("ray_steps") and inp.get("valu_insts") and clock and args.kernel == "auto" and args.lds_bytes == 0 and args.sort == 0 and sched_ok:
        # another cut of the same config (a rank's band at N > 1): the profiled launch's counters scaled by the ray-steps of
        # this launch -- an estimate, marked as one (instructions per ray-step vary by a few per cent between bands)
        k = ray_steps / float(inp["ray_steps"])
        out["inputs"] = {"file": src, "profile": inp.get("profile"), "kernel": inp.get("kernel"), "round": inp.get("round"),
                         "note": "counters of the profiled launch of this config scaled by ray-steps (x %.4f): estimate" % k}
        out["estimated"] = True
        out["achieved"] = inp["valu_insts"] * k / t / 1e9
        out["frac"] = out["achieved"] / out["peak"]
        out["valu"] = {"insts_per_launch": inp["valu_insts"] * k, "insts_per_64_ray_steps": inp["valu_insts"] / (inp["ray_steps"] / 64.0),
                       "counter": "SQ_INSTS_VALU (scaled)", "frac": out["frac"]}
        if inp.get("hbm_bytes"):
            out["traffic"] = inp["hbm_bytes"] * k
            out["traffic_source"] = "%s: %s, scaled by ray-steps (estimate)" % (inp.get("profile"), inp.get("hbm_bytes_formula"))
    else:
        out["inputs"] = {"file": src, "note": "no counter profile of this config: counter-derived fractions omitted"}
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def device_copy_gbs(dev, nbytes=1 << 30, reps=5):
    """Measured HBM copy bandwidth (read + written bytes / time) -- context for the 8 TB/s vendor peak."""
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5,
                    help="untimed steps; the engine settles its schedule choice for the workload during the first four calls")
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument("--kernel", default="auto", choices=["auto", "direct", "slab"])
    ap.add_argument("--schedule", type=int, default=0, help="slab kernel scheduling: 0 auto, 1 static, 2 persistent wavefronts")
    ap.add_argument("--sort", type=int, default=0, help="0 auto, 1 force tile order, -1 caller order")
    ap.add_argument("--lds-bytes", type=int, default=0)
    ap.add_argument("--iterations", type=int, default=100000)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget; 0 disables it")
    ap.add_argument("--roofline-inputs", default=os.path.join(ROOT, "profiles", "roofline_inputs.json"),
                    help="per-launch counters of the march kernel from the rocprofv3 passes (tools/collect_profiles.py)")
    ap.add_argument("--no-rebalance", action="store_true", help="N > 1: keep equal-count row bands (no work rebalancing)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"],
                    help="N > 1: weak = every rank marches R rows of an (R*N) x R lattice (default); strong = the fixed R x R "
                         "lattice cut into N row bands (default for config 4, which BASELINE defines that way)")
    ap.add_argument("--rehearse", action="store_true",
                    help="development: let several ranks share one GPU (local_rank %% device_count) to exercise the "
                         "multi-rank code path on a single-GPU box; timings are meaningless")
    ap.add_argument("--verify-rays", type=int, default=4096,
                    help="N > 1: rays of the gathered records (evenly strided over the whole job) rank 0 checks against the oracle; 0 disables")
    ap.add_argument("--reference-leg", default=None, help=argparse.SUPPRESS)    # internal: child process of reference_leg()
    args = ap.parse_args()
    if args.reference_leg:
        reference_leg_child(args.reference_leg)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs a torch.distributed.run launch with %d ranks" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))

    def log(*a):
        if rank == 0:
            print("[bench]", *a, file=sys.stderr, flush=True)

    if args.rehearse:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            dist.init_process_group(backend="gloo")                # RCCL refuses two ranks on one GPU
        else:
            dist.init_process_group(backend="nccl", device_id=dev)     # "nccl" is RCCL on ROCm
    cdev = torch.device("cpu") if args.rehearse else dev           # where the collectives' tensors live

    cfg = CONFIGS[args.config]
    N = cfg["N"]
    scaling = args.scaling if args.scaling != "auto" else ("strong" if args.config == 4 else "weak")
    eng, scene, bounds, d, tr, lut_host = build_engine(cfg, dev, log)

    inv = [1.0, 1.0, 1.0]
    kern = {"auto": capi.KERNEL_AUTO, "direct": capi.KERNEL_DIRECT, "slab": capi.KERNEL_SLAB}[args.kernel]
    opts = capi.default_options(kernel=kern, sort_rays=args.sort, slab_lds_bytes=args.lds_bytes, schedule=args.schedule)
    stream = torch.cuda.current_stream().cuda_stream

    sp, sd = make_rays(cfg, rank, world, scaling=scaling)
    n = sp.size // 3
    d_sp = torch.from_numpy(sp.view(np.int32)).to(dev)
    d_sd = torch.from_numpy(sd).to(dev)
    # one contiguous output record per rank: [end_pos 3n | end_dir 3n | end_iter n | light n] (32 B/ray)
    out = D.alloc_record(n, dev)
    o_ep, o_ed, o_ei, o_rl = D.record_views(out, n)
    band = None
    if world > 1 and not args.no_rebalance:
        # equal ray counts leave the ranks skewed (rays at the rim of the image leave the volume
        # early: 1.5e9 .. 3.7e9 ray-steps per rank at N = 8), so the bands are re-cut for equal measured work.
        # Untimed set-up: one march of the equal bands, per-row step sums exchanged, contiguous bands of equal
        # work (volumeraytracer_amd.distributed.balanced_bands), rays regenerated.
        eng.trace_device(capi.VRT_F32, n, d_sp.data_ptr(), d_sd.data_ptr(), o_ep.data_ptr(), o_ed.data_ptr(),
                         o_ei.data_ptr(), o_rl.data_ptr(), inv, args.iterations, stream=stream, options=opts)
        torch.cuda.synchronize()
        row_work = o_ei.to(torch.int64).reshape(-1, cfg["R"]).sum(1).to(torch.float64).to(cdev)
        rows_all = [torch.empty_like(row_work) for _ in range(world)]
        dist.all_gather(rows_all, row_work)
        bands = D.balanced_bands(torch.cat(rows_all).cpu().numpy(), world)
        band = bands[rank]
        log("rebalanced row bands:", bands)
        sp, sd = make_rays(cfg, rank, world, band, scaling=scaling)
        n = sp.size // 3
        d_sp = torch.from_numpy(sp.view(np.int32)).to(dev)
        d_sd = torch.from_numpy(sd).to(dev)
        out = D.alloc_record(n, dev)
        o_ep, o_ed, o_ei, o_rl = D.record_views(out, n)
    counts = [n] * world
    if world > 1:
        cnt = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(world)]
        dist.all_gather(cnt, torch.tensor([n], dtype=torch.int64, device=cdev))
        counts = [int(c.item()) for c in cnt]

    # N > 1 over RCCL: double-buffered records, the gather of step k overlaps the march of step k+1
    # (volumeraytracer_amd.distributed.RecordGather); the gloo rehearsal stages through host memory instead
    pipe = D.RecordGather(n, counts, dev, dst=0) if (world > 1 and not args.rehearse) else None

    def step():
        if pipe is not None:
            b_ep, b_ed, b_ei, b_rl = D.record_views(pipe.buffer(), n)
            eng.trace_device(capi.VRT_F32, n, d_sp.data_ptr(), d_sd.data_ptr(), b_ep.data_ptr(), b_ed.data_ptr(),
                             b_ei.data_ptr(), b_rl.data_ptr(), inv, args.iterations, stream=stream, options=opts)
            pipe.launch()
            return
        eng.trace_device(capi.VRT_F32, n, d_sp.data_ptr(), d_sd.data_ptr(), o_ep.data_ptr(), o_ed.data_ptr(),
                         o_ei.data_ptr(), o_rl.data_ptr(), inv, args.iterations, stream=stream, options=opts)
        if world > 1:
            D.gather_records(out.to(cdev), counts, dst=0)

    for _ in range(args.warmup):
        step()
    if pipe is not None:
        pipe.finish()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_begin = time.perf_counter()
    for _ in range(args.steps):
        step()
    if pipe is not None:
        pipe.finish()                                              # every gather of the timed steps has completed
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_begin
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-launch kernel duration from HIP events on the launch stream: re-run the K launches one by
    # one (same inputs, same kernel) outside the wall-clock region so reading the events adds no sync to it
    kernel_ms, clock_ghz = [], []
    for _ in range(args.steps):
        eng.trace_device(capi.VRT_F32, n, d_sp.data_ptr(), d_sd.data_ptr(), o_ep.data_ptr(), o_ed.data_ptr(),
                         o_ei.data_ptr(), o_rl.data_ptr(), inv, args.iterations, stream=stream, options=opts)
        kernel_ms.append(eng.last_march_ms())
        clock_ghz.append(eng.last_march_clock_ghz())
    torch.cuda.synchronize()
    # the schedule these launches used (the optional legs below launch other kernels afterwards)
    sched_observed = eng.last_schedule() if hasattr(capi.load(), "vrt_engine_last_schedule") else None

    # ---- opt-in tolerance mode (vrt_options.fast_math): a SEPARATE figure, never the headline ---------------------------------
    # Same rays, same field, the 10 KiB slab kernel's FAST instantiation; compared with the exact results just computed.
    fast = None
    try:                                                       # (an optional figure must never cost the headline its JSON line)
        if world == 1 and args.kernel == "auto" and args.lds_bytes == 0 and hasattr(capi.Options, "fast_math"):
            exact_rec = out.clone()
            fopts = capi.default_options(kernel=kern, sort_rays=args.sort, slab_lds_bytes=args.lds_bytes, schedule=args.schedule, fast_math=1)
            frec = D.alloc_record(n, dev)
            f_ep, f_ed, f_ei, f_rl = D.record_views(frec, n)

            def fstep():
                eng.trace_device(capi.VRT_F32, n, d_sp.data_ptr(), d_sd.data_ptr(), f_ep.data_ptr(), f_ed.data_ptr(), f_ei.data_ptr(),
                                 f_rl.data_ptr(), inv, args.iterations, stream=stream, options=fopts)
            for _ in range(args.warmup):
                fstep()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fstep()
            torch.cuda.synchronize()
            f_elapsed = time.perf_counter() - t0
            fk = []
            for _ in range(min(args.steps, 5)):
                fstep()
                fk.append(eng.last_march_ms())
            torch.cuda.synchronize()
            x_ep, x_ed, x_ei, _ = D.record_views(exact_rec, n)
            ei_x, ei_f = x_ei.to(torch.int64), f_ei.to(torch.int64)
            same = ei_x == ei_f
            dpos = (x_ep.to(torch.int64) - f_ep.to(torch.int64)).abs().reshape(-1, 3).max(dim=1).values
            ddir = (x_ed.view(torch.float32) - f_ed.view(torch.float32)).abs().reshape(-1, 3).max(dim=1).values
            dnorm = x_ed.view(torch.float32).reshape(-1, 3).norm(dim=1).max()
            fast = {"value": int(ei_f.sum().item()) * args.steps / f_elapsed, "unit": "ray-steps/s", "ms_per_step": f_elapsed / args.steps * 1e3,
                    "kernel_ms": float(np.mean(fk)), "end_iteration_equal_share": float(same.double().mean().item()),
                    "end_iteration_max_abs_diff": int((ei_x - ei_f).abs().max().item()),
                    "position_max_rel_err": float(dpos[same].max().item()) / float(N << 16),
                    "direction_max_rel_err": float((ddir[same].max() / dnorm).item()),
                    "rays_within_1e-5": float(((dpos.double() / float(N << 16) <= 1e-5) & (ddir.double() / float(dnorm) <= 1e-5) & same).double().mean().item()),
                    "note": "vrt_options.fast_math = 1 (tolerance mode); compared with this run's exact results on all rays (errors relative to the "
                            "field extent / the largest direction); a ray that grazes a strongly refracting structure amplifies any rounding "
                            "difference, so the maximum can exceed 1e-5 while rays_within_1e-5 says how many stay inside; NOT the headline"}
            if fast["end_iteration_equal_share"] == 1.0 and fast["position_max_rel_err"] == 0.0 and fast["direction_max_rel_err"] == 0.0:
                fast["ignored"] = True      # the kernel this workload selects (6 KiB segments for dense ray sets, direct) has no tolerance variant
            del exact_rec
    except Exception as ex:                                     # noqa: BLE001
        log("fast_math leg failed:", ex)
        fast = None
    # a scheduler failure (a bounded wait of the time-sliced schedule that ran out) would have left rays unfinished:
    # vrt_trace_status reports it for every device-pointer call issued so far
    if hasattr(capi.load(), "vrt_trace_status"):
        eng.status()
    kavg_local = float(np.mean(kernel_ms))

    # ---- N > 1: what was gathered is what the oracle computes ---------------------------------------------------------
    ranks_info, gathered_ok, verify_note = None, None, None
    if world > 1:
        # one more step, synchronous and timed per rank: march, then the gather on its own
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        if pipe is not None:
            b_ep, b_ed, b_ei, b_rl = D.record_views(pipe.buffer(), n)
            eng.trace_device(capi.VRT_F32, n, d_sp.data_ptr(), d_sd.data_ptr(), b_ep.data_ptr(), b_ed.data_ptr(),
                             b_ei.data_ptr(), b_rl.data_ptr(), inv, args.iterations, stream=stream, options=opts)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            pipe.launch()
            records = pipe.finish()
        else:
            eng.trace_device(capi.VRT_F32, n, d_sp.data_ptr(), d_sd.data_ptr(), o_ep.data_ptr(), o_ed.data_ptr(),
                             o_ei.data_ptr(), o_rl.data_ptr(), inv, args.iterations, stream=stream, options=opts)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            records = D.gather_records(out.to(cdev), counts, dst=0)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        mine = {"rank": rank, "rays": n, "band_rows": list(band) if band is not None else list(D.row_band(lattice_rows(cfg, world, scaling), rank, world)),
                "ray_steps": int(o_ei.to(torch.int64).sum().item()) if pipe is None else int(b_ei.to(torch.int64).sum().item()),
                "kernel_ms": kavg_local, "march_step_ms": (t1 - t0) * 1e3, "gather_ms": (t2 - t1) * 1e3,
                "schedule": sched_observed}
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, mine)
        if rank == 0 and args.verify_rays > 0:
            import oracle
            whole = D.assemble(records, counts)
            pick = D.sample_gathered(counts, args.verify_rays)         # [(rank, local indices)] evenly strided over the whole job
            if d is not None:
                field_host = oracle.port.interleave(d[0].cpu().numpy(), d[1].cpu().numpy(), d[2].cpu().numpy(), tr.cpu().numpy().view(np.uint32))
            else:
                dens, lut_ior, lut_tr = lut_host
                hx, hy, hz, tc_h, _ = oracle.port.scene_build([N, N, N], lut_ior[dens], lut_tr[dens])
                field_host = oracle.port.interleave(hx, hy, hz, tc_h)
            bad, checked, base = 0, 0, 0
            for r, idx in pick:
                if idx.size:
                    rsp, rsd = make_rays(cfg, r, world, tuple(ranks_info[r]["band_rows"]), scaling=scaling)
                    want = oracle.port.trace(field_host, bounds, rsp.reshape(-1, 3)[idx].reshape(-1), rsd.reshape(-1, 3)[idx].reshape(-1),
                                             inv, args.iterations)
                    g = base + idx
                    ok = (np.array_equal(whole["end_position"].reshape(-1, 3)[g], want["end_position"].reshape(-1, 3))
                          and np.array_equal(whole["end_direction_bits"].reshape(-1, 3)[g], want["end_direction"].view(np.uint32).reshape(-1, 3))
                          and np.array_equal(whole["end_iteration"][g], want["end_iteration"])
                          and np.array_equal(whole["remaining_light"][g], want["remaining_light"]))
                    bad += 0 if ok else 1
                    checked += int(idx.size)
                base += counts[r]
            gathered_ok = bad == 0
            verify_note = "%d rays of the gathered records (evenly strided over %d ranks' bands) against the oracle, bit for bit" % (checked, world)
            log("gathered records vs oracle:", "match" if gathered_ok else "MISMATCH in %d rank(s)" % bad, "(%s)" % verify_note)

    if os.environ.get("VRT_DEBUG_STATS") == "1" and rank == 0:
        try:
            log("slab debug stats:", eng.debug_stats())
        except Exception as ex:
            log("no debug stats:", ex)
    steps_local = int(o_ei.to(torch.int64).sum().item())
    steps_total = steps_local
    if world > 1:
        t = torch.tensor([steps_local], dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        steps_total = int(t.item())

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    bytes_per_step = 128 if cfg["field"] == "f32" else 64
    kavg_ms = kavg_local
    roofline = roofline_object(args, cfg, dev, eng, steps_local, kavg_ms, [c for c in clock_ghz if c > 0], bytes_per_step, sched_observed)
    result = {
        "metric": "ray-steps/sec",
        "value": steps_total * args.steps / elapsed,
        "unit": "ray-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": cfg["name"] + ", invscale 1, cap %d" % args.iterations,
                   "rays_per_gpu": n, "ray_steps_per_gpu": steps_local, "field_bytes": eng.field_bytes,
                   "kernel": args.kernel, "sort": args.sort, "parallelism": ("%s x %d ray lattice cut into %d row bands%s, field replicated, %s gather of 32 B/ray to rank 0"
                                   % (lattice_rows(cfg, world, scaling), cfg["R"], world, "" if args.no_rebalance else " of equal measured work",
                                      "gloo (rehearsal on one GPU, host staging)" if args.rehearse else "RCCL")) if world > 1 else "single GPU",
                   "rays_total": int(sum(counts))},
        "roofline": roofline,
    }
    if fast is not None:
        result["fast_math"] = fast
    if world > 1:
        result["ranks"] = ranks_info
        result["gathered_matches_oracle"] = gathered_ok
        result["gathered_check"] = verify_note
    if args.cpu_seconds > 0 and world == 1:                        # CPU baseline: rank 0 at N = 1 only
        import oracle
        if d is not None:
            host = [x.cpu().numpy() for x in d]
            tc_host = tr.cpu().numpy().view(np.uint32)
        else:                                                      # LUT scene: expand and build on the CPU checker too
            t0 = time.time()
            dens, lut_ior, lut_tr = lut_host
            hx, hy, hz, tc_host, _ = oracle.port.scene_build([N, N, N], lut_ior[dens], lut_tr[dens])
            host = [hx, hy, hz]
            log("CPU scene build (oracle, untimed for the baseline): %.1f s" % (time.time() - t0))
        field_host = oracle.port.interleave(host[0], host[1], host[2], tc_host)
        gpu_out = dict(end_position=o_ep.cpu().numpy().view(np.uint32),
                       end_direction=o_ed.cpu().numpy().view(np.float32),
                       end_iteration=o_ei.cpu().numpy().view(np.uint32))
        result["cpu_baseline"] = cpu_baseline(field_host, bounds, sp, sd, inv, args.iterations, gpu_out,
                                              args.cpu_seconds, log)
        if d is not None:                                          # the real reference build, when it travelled and runs here
            sp3, sd3 = sp.reshape(-1, 3), sd.reshape(-1, 3)
            ref = reference_leg(host, tc_host, bounds, sp, sd, inv, args.iterations,
                                lambda idx: oracle.port.trace(field_host, bounds, sp3[idx].reshape(-1), sd3[idx].reshape(-1), inv, args.iterations),
                                args.cpu_seconds, log)
            if ref is not None:
                result["cpu_baseline"]["reference"] = ref
    print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
        if gathered_ok is False:
            sys.exit(3)                                            # a wrong gather must not look like a clean scaling number


if __name__ == "__main__":
    main()
