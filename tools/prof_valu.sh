# VALU instruction count and wave cycles of the bench kernel for each library given (gpurun box).
# usage: bash tools/prof_valu.sh build/libA.so build/libB.so ...   (paths relative to the repo root)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  TAG=valu_$(basename $L .so)
  GIGL_RT_LIB=$R/$L timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/$TAG/sq -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/$TAG.log 2>&1 || exit 1
  (cd $R && python3 tools/pmc_summary.py $TAG && grep -o '"kernel_ms": [0-9.]*' gpurun_out/$TAG.log)
done
