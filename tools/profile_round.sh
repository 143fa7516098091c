#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag> [workload]
# -> gpurun_out/<tag>_bench.json            bench.py line of the workload (default workload: the driver's command incl. cpu_baseline + Llama configs[])
#    gpurun_out/<tag>_kernel_stats.txt      rocprofv3 --kernel-trace of the same workload, TIMED steps only (warm-up dispatches dropped)
#    gpurun_out/pmc_traffic_<workload>.json separate --pmc passes (own runs, --kernel-trace only): HBM bytes per launch per kernel
#    gpurun_out/<tag>_pmc_sq_tcc.txt        SQ / TCC counters of the same passes: MFMA utilisation, LDS conflicts, waits, L2 hit rate
# copy what should be judged into profiles/ (pmc_traffic_<workload>.json is what bench.py reads for roofline.traffic).
set -u
TAG=$1
WL=${2:-gpt2s_B32_L1024_C12}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
STEPS=10; WARM=3
[ "$WL" = "gpt2s_B32_L1024_C12" ] || { STEPS=4; WARM=2; }
cd $R
timeout 900 python bench.py --workload $WL > $OUT/${TAG}_bench.log 2>&1; tail -1 $OUT/${TAG}_bench.log > $OUT/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o bench -- python $R/bench.py --workload $WL --steps $STEPS --warmup $WARM --no-cpu-baseline --no-roofline --no-extra-configs > $OUT/${TAG}_trace.log 2>&1
cd $R
python tools/rocprof_summary.py $(ls $OUT/${TAG}_trace/*.db | head -1) $OUT/${TAG}_kernel_stats.txt $STEPS $WARM
timeout 900 bash tools/pmc_run.sh $TAG $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra-configs
python tools/pmc_traffic.py $TAG $OUT/pmc_traffic_${WL}.json $WL > $OUT/${TAG}_pmc_traffic.txt 2>&1
python tools/pmc_sq.py $TAG > $OUT/${TAG}_pmc_sq_tcc.txt 2>&1      # MFMA utilisation (0..1), LDS conflicts, wait share, L2 hit rate per kernel
rm -rf $OUT/${TAG}_trace $OUT/pmc_${TAG}_*      # keep gpurun_out/ small: summaries only
head -3 $OUT/${TAG}_kernel_stats.txt; head -5 $OUT/${TAG}_pmc_traffic.txt; cut -c1-200 $OUT/${TAG}_bench.json
