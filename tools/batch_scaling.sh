#!/bin/bash
# Launch time against batch size (gpurun box): the slope is the steady-state rate, the intercept what ramp and drain cost.
# usage: tools/batch_scaling.sh TESS OUT [variant...]
T=${1:-158}; OUT=${2:-gpurun_out/batch_scaling_s$T.txt}; shift; shift
V=${@:-stats=0}
: > $OUT
for L in 21 22 23 24 25 26; do
  echo "== 2^$L rays" >> $OUT
  timeout -k 10 280 python tools/sweep.py --tess $T --rays-log2 $L --rounds 3 $V 2>&1 | grep -E "median|changed" >> $OUT || exit 1
done
cat $OUT
