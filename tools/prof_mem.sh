# Memory-path counters of the bench kernel (one rocprofv3 --pmc pass per counter group).
# usage (on the GPU box): bash tools/prof_mem.sh TAG
R=$GRAFT_REPO_ROOT
TAG=$1
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
    local name=$1; shift
    timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/${TAG}/$name -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_$name.log 2>&1 || { echo "$name failed"; return 1; }
    echo "$name done"
}
run tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum &&
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum &&
run ta TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum &&
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
