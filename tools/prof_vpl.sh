#!/bin/bash
# Kernel timeline of configs[0] (bench.py --config vpl) on the gpurun box.  usage: tools/prof_vpl.sh TAG
TAG=${1:-r03_vpl}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python3 $R/bench.py --config vpl --steps 50 --warmup 5 --no-cpu-baseline > $R/gpurun_out/$TAG.log 2>&1 || exit 1
find $R/gpurun_out/$TAG -name "*kernel_stats.csv" | xargs cat | cut -c1-120,200-320
grep -o 'ms_per_step[^,]*' $R/gpurun_out/$TAG.log
