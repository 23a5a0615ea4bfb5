#!/bin/bash
# Round 3: VALU instruction counts of the two kernels that are bound by vector-ALU issue (configs[1]'s shadow-pair
# kernel, configs[4]'s transmittance kernel) under their bench commands (gpurun box).  usage: tools/prof_extras_r03.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in cornell_shadow transmittance; do
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/r03_$cfg/sq -- python3 $R/bench.py --config $cfg --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/r03_${cfg}_sq.log 2>&1 || exit 1
done
cd $R && python3 - <<'PY'
import csv, glob, collections
for cfg, pat in (("cornell_shadow", "trace_kernel<true, true, 2"), ("transmittance", "transm_shadow_kernel")):
    out = open("gpurun_out/r03_%s_counters.txt" % cfg, "w")
    for f in sorted(glob.glob("gpurun_out/r03_%s/*/*/*counter_collection.csv" % cfg)):
        acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if pat in k:
                acc[k.replace("(anonymous namespace)::", "")[:90]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for k, d in acc.items():
            print(k, file=out)
            for c, v in d.items():
                vals = list(v.values())
                print("    %-32s %.6g per launch (%d launches)" % (c, sum(vals) / len(vals), len(vals)), file=out)
    out.close()
    print(open("gpurun_out/r03_%s_counters.txt" % cfg).read())
PY
