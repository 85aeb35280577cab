#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/prof_bench.sh (gpurun_out/prof) into the summaries kept under profiles/:

    python tools/summarize_prof.py gpurun_out/prof profiles/round1 [steps_profiled]

  bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, jk:: kernels only
  bench_hbm_counters.json  FETCH_SIZE / WRITE_SIZE per jk:: kernel (KB as reported; two separate --pmc passes) and the
                           counting pipeline's HBM bytes per step (bench.py reads roofline.traffic from here)
"""
import csv, glob, json, os, sys

src, dst = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6      # --steps 2 --warmup 4
csv.field_size_limit(1 << 30)


def newest(pattern):
    files = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    if not files:
        sys.exit("missing " + pattern)
    return files[-1]


os.makedirs(dst, exist_ok=True)
def short(n):
    n = n[5:] if n.startswith("void ") else n
    return n.split("(")[0]


rows = [r for r in csv.DictReader(open(newest("trace/*/*kernel_stats.csv"))) if short(r["Name"]).startswith("jk::")]
with open(os.path.join(dst, "bench_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]])

out = {}
for ctr, pat in (("FETCH_SIZE", "pmc_fetch/*/*counter_collection.csv"), ("WRITE_SIZE", "pmc_write/*/*counter_collection.csv")):
    acc = {}
    for r in csv.DictReader(open(newest(pat))):
        name = short(r["Kernel_Name"])
        if r["Counter_Name"] != ctr or not name.startswith("jk::"):
            continue
        a = acc.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    out[ctr] = {k: {"dispatches": v[0], "sum_KB": v[1], "KB_per_dispatch": v[1] / v[0]} for k, v in acc.items()}
pipe = ("jk::part1_kernel", "jk::clamp_counts_kernel", "jk::part2_kernel", "jk::part2f_kernel", "jk::region_insert_kernel", "jk::import3", "jk::count_kernel")
tot_kb = sum(v["sum_KB"] for c in ("FETCH_SIZE", "WRITE_SIZE") for k, v in out[c].items() if k.startswith(pipe))
p1 = [v for k, v in out["FETCH_SIZE"].items() if k.startswith("jk::part1_kernel")]
launches = max(1, (p1[0]["dispatches"] if p1 else steps) // steps)
fetch_kb = sum(v["sum_KB"] for k, v in out["FETCH_SIZE"].items() if k.startswith(pipe))
write_kb = sum(v["sum_KB"] for k, v in out["WRITE_SIZE"].items() if k.startswith(pipe))
out["counting_pipeline"] = {"steps_profiled": steps, "hbm_bytes_per_step": tot_kb * 1024.0 / steps, "launches_per_step": launches,
                            "hbm_bytes_per_launch": tot_kb * 1024.0 / steps / launches,
                            # MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced
                            # streaming read (and these kernels' reads are such streams: bases, record lists, region images) --
                            # doubled before it is compared with a byte count; WRITE_SIZE is exact for streaming stores
                            "hbm_bytes_per_step_corrected": (2.0 * fetch_kb + write_kb) * 1024.0 / steps,
                            "fetch_bytes_per_step_as_reported": fetch_kb * 1024.0 / steps, "write_bytes_per_step": write_kb * 1024.0 / steps,
                            "note": "part1 + part2 + region_insert (+ deferred import): FETCH_SIZE + WRITE_SIZE as reported by rocprofv3, and the "
                                    "corrected figure 2 x FETCH_SIZE + WRITE_SIZE (gfx950 tallies 128-B read requests at 64 B)"}
# the polishing kernels, per polish CALL: a table polishes in several lanes from its second call on, so no kernel's dispatch count is
# the number of calls; the profiled process says how many it made (bench.py: warm-up + timed steps + the untimed host-buffer calls)
pk = ("scan_classify_batch", "find_sync_batch", "find_clean_batch", "seg_init", "seg_walk", "seg_summary", "seg_gather", "seg_stitch", "rescan_batch")
calls = steps + 3
try:
    calls = int(json.loads(open(os.path.join(src, "trace_run.json")).read().strip().splitlines()[-1])["polish_calls_in_process"])
except Exception:
    pass
pf = sum(v["sum_KB"] for k, v in out["FETCH_SIZE"].items() if k.startswith(tuple("jk::%s_kernel" % n for n in pk)))
pw = sum(v["sum_KB"] for k, v in out["WRITE_SIZE"].items() if k.startswith(tuple("jk::%s_kernel" % n for n in pk)))
out["polishing"] = {"polish_calls_profiled": calls, "fetch_bytes_per_call_as_reported": pf * 1024.0 / calls, "write_bytes_per_call": pw * 1024.0 / calls,
                    "hbm_bytes_per_call": (pf + pw) * 1024.0 / calls,
                    "note": "FETCH_SIZE + WRITE_SIZE of the polishing kernels as reported, divided by the polish CALLS of the profiled process (their reads "
                            "are mostly 16-byte slot probes that each bring a 64-byte sector: not the wide streaming reads whose FETCH_SIZE gfx950 halves)"}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jasper_amd._lib import kernel_source_digest
out["kernel_source_sha256"] = kernel_source_digest()      # bench.py cites these counters only while the sources are the same
json.dump(out, open(os.path.join(dst, "bench_hbm_counters.json"), "w"), indent=1)
for r in rows[:14]:
    print("%-34s calls %4s avg %10.1f us" % (short(r["Name"])[:34], r["Calls"], float(r["AverageNs"]) / 1e3))
print("polishing: %.2f GB per call over %d calls" % (out["polishing"]["hbm_bytes_per_call"] / 1e9, calls))
print("counting pipeline: %.2f GB HBM traffic per step as reported, %.2f GB corrected" % (out["counting_pipeline"]["hbm_bytes_per_step"] / 1e9, out["counting_pipeline"]["hbm_bytes_per_step_corrected"] / 1e9))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, v in out[c].items():
        print("  %-11s %-34s %8.3f GB per dispatch x %d" % (c, k[:34], v["KB_per_dispatch"] * 1024 / 1e9, v["dispatches"]))
