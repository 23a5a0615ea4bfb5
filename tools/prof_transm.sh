#!/bin/bash
# Vector-ALU instruction count of transm_shadow_kernel under bench.py --config transmittance (gpurun box).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/transm/sq -- python3 $R/bench.py --config transmittance --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/transm_sq.log 2>&1 || exit 1
cd $R && python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/transm/*/*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'transm' in k:
            acc[k.replace('(anonymous namespace)::','')[:70]][r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    for k, d in acc.items():
        print(k)
        for c, v in d.items():
            vals = list(v.values())
            print('    %-32s %.5g per launch (%d launches)' % (c, sum(vals) / len(vals), len(vals)))
PY
