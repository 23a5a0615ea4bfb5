R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 250 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/anyprof/sq -- python3 $R/bench.py --no-cpu-baseline --extras --steps 1 --warmup 0 > $R/gpurun_out/anyprof_sq.log 2>&1 && echo sq done
timeout -s KILL 250 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/anyprof/kt -- python3 $R/bench.py --no-cpu-baseline --extras --steps 1 --warmup 0 > $R/gpurun_out/anyprof_kt.log 2>&1 && echo kt done
