#!/bin/bash
# usage: tools/pmc_run.sh <tag> <python args...>   -> gpurun_out/pmc_<tag>_<pass>/  (one rocprofv3 --pmc pass per counter group)
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_${TAG}_$name -o r -- python "${PYARGS[@]}" > $OUT/pmc_${TAG}_$name.log 2>&1; }
PYARGS=("$@")
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
