#!/usr/bin/env python3
"""Condenses a tools/prof_*.sh output directory (gpurun_out/<dir>) into profiles/<tag>_*.{csv,json}."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
os.makedirs("profiles", exist_ok=True)
ks = glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], "profiles/%s_kernel_stats.csv" % tag)
out = {}
for f in glob.glob(os.path.join(src, "*", "*", "*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "trace_kernel" in r["Kernel_Name"] or "trace3_kernel" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        out[c] = {"per_launch_mean": sum(v) / len(v), "launches": len(v), "kernel": k.replace("(anonymous namespace)::", "")}
derived = {}
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB of fabric (L2 memory-side) requests; on gfx950
    # FETCH_SIZE counts 64 B per request (MI355X_MICROARCH.md §HBM): TCC_MISS x 64 B agrees for this
    # scattered 8-16 B access pattern, so no x2 correction is applied here (that applies to wide streams).
    derived["l2_fabric_read_bytes_per_launch"] = out["FETCH_SIZE"]["per_launch_mean"] * 1024
    derived["l2_fabric_write_bytes_per_launch"] = out["WRITE_SIZE"]["per_launch_mean"] * 1024
if "TCC_HIT_sum" in out:
    derived["l2_hit_rate"] = out["TCC_HIT_sum"]["per_launch_mean"] / (out["TCC_HIT_sum"]["per_launch_mean"] + out["TCC_MISS_sum"]["per_launch_mean"])
if "SQ_WAVE_CYCLES" in out:
    derived["wave_wait_fraction"] = out["SQ_WAIT_ANY"]["per_launch_mean"] / out["SQ_WAVE_CYCLES"]["per_launch_mean"]
    derived["waves_per_launch"] = out["SQ_WAVES"]["per_launch_mean"]
json.dump({"note": note, "source": src, "counters": out, "derived": derived}, open("profiles/%s_pmc.json" % tag, "w"), indent=1)
print(json.dumps(derived, indent=1))
