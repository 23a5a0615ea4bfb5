#!/bin/bash
# Counters of the Cornell shadow-pair kernel (trace_kernel<triples, any-hit, generator 2>) under bench.py --config cornell_shadow (gpurun box).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --config cornell_shadow --no-cpu-baseline --steps 2 --warmup 1"
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/shadow/sq -- $B > $R/gpurun_out/shadow_sq.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_MISS_sum TCC_HIT_sum --output-format csv -d $R/gpurun_out/shadow/tcp -- $B > $R/gpurun_out/shadow_tcp.log 2>&1 || exit 1
cd $R && python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/shadow/*/*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'trace_kernel' in k:
            acc[k.replace('(anonymous namespace)::','')[:70]][r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    for k, d in acc.items():
        print(k)
        for c, v in d.items():
            vals = list(v.values())
            print('    %-32s %.5g per launch (%d launches)' % (c, sum(vals) / len(vals), len(vals)))
PY
