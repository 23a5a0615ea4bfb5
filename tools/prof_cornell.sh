# Counters of the Cornell shadow-batch kernel (trace_kernel<triples, any-hit, generator 2>) under bench.py --extras (gpurun box).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 280 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/cornell/sq -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 --extras > $R/gpurun_out/cornell_sq.log 2>&1 || exit 1
timeout -s KILL 280 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_MISS_sum TCC_HIT_sum --output-format csv -d $R/gpurun_out/cornell/tcp -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 --extras > $R/gpurun_out/cornell_tcp.log 2>&1 || exit 1
timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cornell/kt -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 --extras > $R/gpurun_out/cornell_kt.log 2>&1 || exit 1
cd $R && python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/cornell/*/*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'trace' in k:
            acc[k.replace('(anonymous namespace)::','')[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        print(k)
        for c, v in d.items():
            # counters are per dispatch and per XCC/SE instance: sum instances of one dispatch = len(v)/dispatches unknown; print mean and count
            print('    %-32s sum %.4g over %d rows' % (c, sum(v), len(v)))
for f in glob.glob('gpurun_out/cornell/kt/*/*kernel_stats.csv'):
    print(open(f).read()[:3000])
PY
