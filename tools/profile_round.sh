#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
# -> gpurun_out/<tag>_bench.json (default bench incl. cpu_baseline), <tag>_kernel_stats.txt (rocprofv3 --kernel-trace --stats
#    of the same command), PMC passes (own runs, --kernel-trace only) -> <tag>_pmc_traffic.json / .txt
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd $R
timeout 900 python bench.py > $OUT/${TAG}_bench.log 2>&1; tail -1 $OUT/${TAG}_bench.log > $OUT/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/${TAG}_trace.log 2>&1
cd $R
python tools/rocprof_summary.py $(ls $OUT/${TAG}_trace/*.db | head -1) $OUT/${TAG}_kernel_stats.txt 13
timeout 900 bash tools/pmc_run.sh $TAG $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline
python tools/pmc_traffic.py $TAG $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_pmc_traffic.txt 2>&1
head -3 $OUT/${TAG}_kernel_stats.txt; head -5 $OUT/${TAG}_pmc_traffic.txt; cut -c1-200 $OUT/${TAG}_bench.json
