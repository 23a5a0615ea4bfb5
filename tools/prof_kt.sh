# kernel-trace stats of one sweep variant: bash tools/prof_kt.sh <outdir> <variant...>
R=$GRAFT_REPO_ROOT
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$OUT -- python3 $R/tools/sweep.py --rounds 3 "$@" > $R/gpurun_out/$OUT.log 2>&1
cat $R/gpurun_out/$OUT/*/*_kernel_stats.csv | cut -c1-200
