#!/bin/bash
# every pass is wrapped in its own `timeout`.
# usage: tools/pmc_mem.sh <tag> <python args...>  -> memory-path counters (TA/TCP/TCC), one rocprofv3 --pmc pass per group
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
PYARGS=("$@")
run() { name=$1; shift; timeout 120 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_${TAG}_$name -o r -- python "${PYARGS[@]}" > $OUT/pmc_${TAG}_$name.log 2>&1; }
run tcp1 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
# (a TA_* pass hung rocprofv3 on this pool for 15 min on 2026-09-28: left out)
run tcc2 TCC_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_avr TCC_EA0_RDREQ_sum
run tcp2 TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_HIT_LRU_READ TCP_GATE_EN1_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
