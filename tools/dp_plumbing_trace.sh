#!/bin/bash
# usage (GPU box, repo root): tools/dp_plumbing_trace.sh -> gpurun_out/r03_dp_plumbing_metric.txt + r03_dp_plumbing_kernel_stats.txt
# metric workload, plain vs --dp-plumbing alternated, then a rocprofv3 kernel trace of the --dp-plumbing run (which kernels the DP machinery adds)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
: > $OUT/r03_dp_plumbing_metric.txt
for rep in 1 2 3; do for mode in plain dp; do
  extra=""; [ $mode = dp ] && extra="--dp-plumbing"
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-live-traffic $extra > /tmp/m_$mode.log 2>/tmp/m_$mode.err
  python - $mode $rep /tmp/m_$mode.log >> $OUT/r03_dp_plumbing_metric.txt <<'PY'
import json, sys
lines = open(sys.argv[3]).read().strip().splitlines()
d = json.loads([l for l in lines if l.startswith("{")][-1])
print(f"{sys.argv[1]:6s} rep {sys.argv[2]}  {d['value']:9.2f} samples/s  {d['ms_per_step']:7.3f} ms/step  trainer loop {d['trainer_loop']['samples_per_s']}  json_last={lines[-1].startswith('{')}")
PY
done; done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/dpp_trace -o bench -- python $R/bench.py --dp-plumbing --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra-configs --no-live-traffic > $OUT/dpp_trace.log 2>&1
cd $R
python tools/rocprof_summary.py $(ls $OUT/dpp_trace/*.db $OUT/dpp_trace/*/*.db 2>/dev/null | head -1) $OUT/r03_dp_plumbing_kernel_stats.txt 10 3
rm -rf $OUT/dpp_trace
cat $OUT/r03_dp_plumbing_metric.txt; head -60 $OUT/r03_dp_plumbing_kernel_stats.txt | cut -c1-160
