#!/bin/bash
# LDS and wait counters of the transmittance kernel (configs[4]) under its bench command (gpurun box).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/r03_transm_lds -- python3 $R/bench.py --config transmittance --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/r03_transm_lds.log 2>&1 || exit 1
cd $R && python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("gpurun_out/r03_transm_lds/*/*counter_collection.csv") + glob.glob("gpurun_out/r03_transm_lds/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "transm_shadow_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for c, d in sorted(acc.items()):
    v = list(d.values()); print("%-28s %.6g per launch (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
