#!/usr/bin/env python3
"""Turns gpurun_out/prof_r<N> (scripts/profile_round<N>.sh) into the committed summaries
under profiles/: kernel stats, PMC counters per kernel, and round<N>_pmc.json (HBM bytes per
launch, corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE tallies 128-B read requests at
64 B on gfx950 -> x2; both are in KiB).  Usage: summarize_profiles.py [round]  (default 2)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = int(sys.argv[1]) if len(sys.argv) > 1 else 2
SRC = os.path.join(ROOT, "gpurun_out", "prof_r%d" % ROUND)
TRACES = ("partitioned", "atomic") if ROUND == 1 else (("k31", "k63") if ROUND == 2 else ("k31", "k63", "zipf63", "l33"))
PMCDIR = "pmc_partitioned_*" if ROUND == 1 else "pmc_k31_*"
BUILD = ("tsx::build_segments_kernel" if ROUND == 1 else "tsx::build_segments_stream_kernel<false>" if ROUND == 2
         else "tsx::build_segments_stream_kernel<false, true>")
DST = os.path.join(ROOT, "profiles")
TEXT_BYTES = 2081065118

for path in TRACES:
    f = sorted(glob.glob(os.path.join(SRC, "trace_" + path, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if f:
        shutil.copy(f[-1], os.path.join(DST, "round%d_kernel_stats_%s.csv" % (ROUND, path)))

agg = collections.defaultdict(dict)
per_dispatch = collections.defaultdict(lambda: collections.defaultdict(dict))  # kernel -> counter -> {dispatch id: value}
def newest(pattern):
    """gpurun merges every run's files into gpurun_out/: keep the newest file of each profile directory."""
    best = {}
    for f in glob.glob(pattern):
        d = os.path.dirname(os.path.dirname(f))
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return [best[d] for d in sorted(best)]


for f in newest(os.path.join(SRC, PMCDIR, "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] = agg[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d = per_dispatch[k][r["Counter_Name"]]
        d[int(r["Dispatch_Id"])] = d.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
with open(os.path.join(DST, "round%d_pmc_counters_%s.csv" % (ROUND, "partitioned" if ROUND == 1 else "k31")), "w") as out:
    out.write("kernel,counter,sum_over_dispatches_of_one_bench_run(steps=1,warmup=0)\n")
    for k in sorted(agg):
        for c in sorted(agg[k]):
            out.write("%s,%s,%.3f\n" % (k, c, agg[k][c]))

summary_path = os.path.join(DST, "round%d_pmc.json" % ROUND)
summary = json.load(open(summary_path)) if os.path.exists(summary_path) else {}
if "kernel" in summary:  # first layout of this file: the atomic path only
    summary = {"atomic": summary}
part = {"command": "rocprofv3 --pmc <set> --output-format csv -- python3 bench.py --steps 1 --warmup 0 "
                   "--no-cpu-baseline (one pass per counter set, default = partitioned path, l=30)",
        "kernels": {}}
tot_r = tot_w = 0.0
for k, v in agg.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v and not k.startswith("__amd") and "synth" not in k and "occupied" not in k:
        rd, wr = v["FETCH_SIZE"] * 1024 * 2, v["WRITE_SIZE"] * 1024
        part["kernels"][k] = {"hbm_read_bytes": rd, "hbm_write_bytes": wr,
                              "TCC_EA0_ATOMIC_sum": v.get("TCC_EA0_ATOMIC_sum")}
        tot_r += rd
        tot_w += wr
# the scan kernel: fused with radix level 1 (scan_part_kernel) where the run used it
# (or split in two: strip_desc_kernel + walk_part_kernel -- the scan stage is then both)
TWO = "tsx::walk_part_kernel" in part["kernels"]
WALK = "tsx::walk_part_kernel" if "tsx::walk_part_kernel" in part["kernels"] else "tsx::walk_part_kernel<512>"
TWO = TWO or WALK in part["kernels"]
SCAN = WALK if TWO else ("tsx::scan_part_kernel" if "tsx::scan_part_kernel" in part["kernels"] else "tsx::scan_log_kernel")
FUSED = SCAN != "tsx::scan_log_kernel"
scan = part["kernels"].get(SCAN, {})
part["kernel"] = SCAN
part["hbm_bytes_per_launch"] = scan.get("hbm_read_bytes", 0) + scan.get("hbm_write_bytes", 0)
# the kernel a step spends most time in is the segment build since the scan was split (bench.py: roofline.kernel)
# bench.py stages: the partition kernel runs twice per step (level 1, then level 2), told apart by dispatch order
def nth_dispatch_bytes(kernel, nth):
    v = per_dispatch.get(kernel, {})
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        return None
    rd = [v["FETCH_SIZE"][i] for i in sorted(v["FETCH_SIZE"])]
    wr = [v["WRITE_SIZE"][i] for i in sorted(v["WRITE_SIZE"])]
    if nth >= len(rd) or nth >= len(wr):
        return None
    return rd[nth] * 2048 + wr[nth] * 1024
def plus(a, b):
    return None if a is None or b is None else a + b
part["stages"] = {"scan": plus(nth_dispatch_bytes(SCAN, 0), nth_dispatch_bytes("tsx::strip_desc_kernel", 0)) if TWO
                          else nth_dispatch_bytes(SCAN, 0),
                  "level1": None if FUSED else nth_dispatch_bytes("tsx::partition_ring_kernel" + ("" if ROUND == 1 else "<1>"), 0),
                  "level2": nth_dispatch_bytes("tsx::partition_ring_kernel" + ("" if ROUND == 1 else "<1>" if ROUND == 2 else "<1, 512, false>"), 0 if FUSED else 1),
                  "build": nth_dispatch_bytes(BUILD, 0)}
part["hbm_bytes_whole_path_per_step"] = tot_r + tot_w
lc = agg.get("tsx::line_count_kernel", {})
if "FETCH_SIZE" in lc:
    part["calibration"] = ("line_count_kernel streams the %d-byte text once: FETCH_SIZE*1024*2 = %.0f bytes (%.3fx)"
                           % (TEXT_BYTES, lc["FETCH_SIZE"] * 2048, lc["FETCH_SIZE"] * 2048 / TEXT_BYTES))
summary["partitioned"] = part
json.dump(summary, open(summary_path, "w"), indent=1)
print(json.dumps(part, indent=1))
