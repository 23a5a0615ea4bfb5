# Counters of the closest-hit kernel on S(500) (10 M triangles, HBM-resident tree) (gpurun box).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 280 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/s500/tcc -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --tess 500 > $R/gpurun_out/s500_tcc.log 2>&1 || exit 1
timeout -s KILL 280 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $R/gpurun_out/s500/sq -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --tess 500 > $R/gpurun_out/s500_sq.log 2>&1 || exit 1
cd $R && python3 tools/pmc_summary.py s500 && grep -o '"kernel_ms": [0-9.]*' gpurun_out/s500_sq.log
