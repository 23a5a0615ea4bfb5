#!/bin/bash
# Scheduling knobs on configs[1] (bench.py --config cornell_shadow): kernel ms and Mrays/s per setting (gpurun box).
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 100 python bench.py --config cornell_shadow --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(j['roofline']['kernel_ms'], j['value'])"; }
echo base; run
for c in 768 1024 1536; do for r in 40 48 56 64; do echo chunk $c guided 0 refill $r; run --opt chunk=$c --opt guided=0 --opt refill_thresh=$r; done; done
for t in 8 12 32 40; do echo chunk 1024 guided 0 refill 48 texit $t; run --opt chunk=1024 --opt guided=0 --opt refill_thresh=48 --opt t_exit=$t; done
for g in 1 2 8; do echo chunk 1024 guided $g refill 48; run --opt chunk=1024 --opt guided=$g --opt refill_thresh=48; done
echo any_atomic 0 chunk 1024; run --opt chunk=1024 --opt guided=0 --opt any_atomic=0
echo shadow_order 0 chunk 1024 refill 48; run --opt chunk=1024 --opt guided=0 --opt refill_thresh=48 --opt shadow_order=0
