"""The committed measurement inputs bench.py's roofline object quotes (profiles/roofline_inputs.json, written by
tools/collect_profiles.py from rocprofv3 passes on the GPU box) must stay usable: every entry names an existing profile,
carries the counters and the schedule of the launch they were taken from, and the script still reads the schedule out of
both generations of the march kernel's signature.  CPU-only; nothing here touches a GPU."""
import importlib.util
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_roofline_inputs_are_complete():
    rj = json.load(open(os.path.join(ROOT, "profiles", "roofline_inputs.json")))
    assert {"config3", "config4", "config5"} <= set(rj)
    for name, e in rj.items():
        assert os.path.exists(os.path.join(ROOT, e["profile"])), (name, e["profile"])
        assert "k_march_slab" in e["kernel"], name
        assert e["schedule"] in (1, 2, 3), (name, e["schedule"])          # bench.py quotes the counters only for the same schedule
        assert e["ray_steps"] > 0 and e["valu_insts"] > 0 and e["lds_insts"] > 0 and e["hbm_bytes"] > 0, name
        assert 0.0 < e["lane_utilisation"] <= 1.0 and 0.0 <= e["lds_bank_conflict_share"] < 1.0, name
        # the kernel time of the profiled launch agrees with the --stats table committed next to it
        stats = open(os.path.join(ROOT, "profiles", "%s_kernel_stats_%s.txt" % (e["round"], name))).read()
        assert e["kernel"][:40] in stats, name


def test_schedule_is_read_from_both_kernel_signatures():
    src = open(os.path.join(ROOT, "tools", "collect_profiles.py")).read()
    pat = re.search(r'msch = re\.search\(r"([^"]+)", march\[0\]\)', src)
    assert pat, "collect_profiles.py no longer parses the schedule the way this test knows"
    rx = re.compile(pat.group(1))
    for sig, want in (("k_march_slab<float, float, 10, 4, false, 2>(vrt::MarchParams)", 3),
                      ("k_march_slab<short, float, 10, 4, false, 1, false>(vrt::MarchParams)", 2),
                      ("k_march_slab<float, short, 6, 6, true, 0, true>(vrt::MarchParams)", 1)):
        m = rx.search(sig)
        assert m and int(m.group(1)) + 1 == want, sig


def test_bench_reads_the_inputs_it_documents():
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    src = open(spec.origin).read()
    for key in ("valu_insts", "hbm_bytes", "ray_steps", "schedule"):
        assert key in src, key
