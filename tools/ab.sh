#!/bin/bash
# A/B timing of library builds on one device, interleaved twice: tools/ab.sh OUT libA.so libB.so ... [-- sweep args]
# (run on the GPU box; each run is tools/sweep.py under GIGL_RT_LIB)
set -e
OUT=$1; shift
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" == "--" ] && shift
ARGS=("$@"); [ ${#ARGS[@]} -eq 0 ] && ARGS=(base)
: > "$OUT"
for pass in 1 2; do
  for L in "${LIBS[@]}"; do
    echo "== $L (pass $pass)" >> "$OUT"
    GIGL_RT_LIB=$L timeout -k 10 240 python3 tools/sweep.py "${ARGS[@]}" 2>&1 | grep -v amdgpu.ids >> "$OUT"
  done
done
