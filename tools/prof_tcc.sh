# L2 hit/miss and fabric bytes of the bench kernel for the library named by GIGL_RT_LIB.
# usage (on the GPU box): GIGL_RT_LIB=build/libX.so bash tools/prof_tcc.sh TAG
R=$GRAFT_REPO_ROOT
TAG=$1
cd /tmp && export TMPDIR=/tmp
[ -n "$GIGL_RT_LIB" ] && export GIGL_RT_LIB=$R/$GIGL_RT_LIB
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/${TAG}/tcc -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_tcc.log 2>&1 && echo tcc done &&
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}/fetch -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_fetch.log 2>&1 && echo fetch done &&
timeout -s KILL 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $R/gpurun_out/${TAG}/tcp -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_tcp.log 2>&1 && echo tcp done
