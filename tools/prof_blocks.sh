# TCC hit/miss and VALU counts of the closest-hit kernel on S(500) with and without leaf triangle blocks
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lb in 0 2; do
  timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/blocks/tcc_lb$lb -- python3 $R/bench.py --tess 500 --rays-log2 24 --no-cpu-baseline --steps 2 --warmup 1 --opt leaf_blocks=$lb > $R/gpurun_out/blocks_tcc_lb$lb.log 2>&1 || exit 1
  timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/blocks/sq_lb$lb -- python3 $R/bench.py --tess 500 --rays-log2 24 --no-cpu-baseline --steps 2 --warmup 1 --opt leaf_blocks=$lb > $R/gpurun_out/blocks_sq_lb$lb.log 2>&1 || exit 1
done
cd $R && python3 tools/pmc_summary.py blocks
