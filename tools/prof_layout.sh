# TCC hit/miss/request counters of the bench kernel under device layouts 0 and 1 (gpurun box).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for L in ${LAYOUTS:-0 1}; do
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/layout$L/tcc -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --opt layout=$L > $R/gpurun_out/layout${L}_tcc.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $R/gpurun_out/layout$L/tcp -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --opt layout=$L > $R/gpurun_out/layout${L}_tcp.log 2>&1 || exit 1
done
cd $R && python3 tools/pmc_summary.py $(for L in ${LAYOUTS:-0 1}; do echo layout$L; done)
