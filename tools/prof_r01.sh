set -e
TAG=${1:-prof_r01}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}/kt -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $R/gpurun_out/${TAG}_kt.log 2>&1
echo kt done
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}/fetch -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_fetch.log 2>&1
echo fetch done
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}/write -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_write.log 2>&1
echo write done
timeout -s KILL 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/${TAG}/tcc -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_tcc.log 2>&1
echo tcc done
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/${TAG}/sq -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_sq.log 2>&1
echo sq done
ls -R $R/gpurun_out/$TAG | head -50
