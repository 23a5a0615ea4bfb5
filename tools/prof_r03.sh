#!/bin/bash
# Round-3 profile of the default bench (S(158)) and of configs[3]'s scene at 2^24 rays (S(500)):
# kernel-trace stats, then one --pmc pass per counter group (gpurun box).  usage: tools/prof_r02.sh TAG
set -e
TAG=${1:-r03_v13}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-other-configs"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}/kt -- $B --steps 5 --warmup 2 > $R/gpurun_out/${TAG}_kt.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}/fetch -- $B --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_fetch.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}/write -- $B --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_write.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/${TAG}/tcc -- $B --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_tcc.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/${TAG}/sq -- $B --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_sq.log 2>&1
echo s158 done
S="$B --tess 500 --steps 2 --warmup 1"
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_s500/kt -- $S > $R/gpurun_out/${TAG}_s500_kt.log 2>&1
timeout -s KILL 280 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/${TAG}_s500/tcc -- $S > $R/gpurun_out/${TAG}_s500_tcc.log 2>&1
timeout -s KILL 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_s500/fetch -- $S > $R/gpurun_out/${TAG}_s500_fetch.log 2>&1
timeout -s KILL 280 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_s500/write -- $S > $R/gpurun_out/${TAG}_s500_write.log 2>&1
timeout -s KILL 280 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/${TAG}_s500/sq -- $S > $R/gpurun_out/${TAG}_s500_sq.log 2>&1
echo s500 done
