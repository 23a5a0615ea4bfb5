#!/bin/bash
# Instruction-fetch counters of photon_small_kernel under bench.py --config vpl (gpurun box).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_INSTS --output-format csv -d $R/gpurun_out/r03_vpl_icache -- python3 $R/bench.py --config vpl --no-cpu-baseline --steps 20 --warmup 5 > $R/gpurun_out/r03_vpl_icache.log 2>&1 || exit 1
cd $R && python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/r03_vpl_icache/*/*counter_collection.csv") + glob.glob("gpurun_out/r03_vpl_icache/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "photon_small_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for c, d in sorted(acc.items()):
    v = list(d.values()); print("%-28s %.6g per launch (%d launches; min %.6g max %.6g)" % (c, sum(v) / len(v), len(v), min(v), max(v)))
PY
