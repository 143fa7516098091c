#!/bin/bash
# in-step A/B of two libraries: alternating bench runs inside one gpurun call, per-dispatch kernel times of the profiled replay.
# usage: tools/diag/ab_lib.sh <base_lib.so | VAR=VALUE> <kernel regex> [workload] [reps]  -> prints ms/step and the matching kernels' avg us for both
# (first argument with a '=': the baseline is the SAME library with that environment knob set)
base=$1; case "$base" in *=*) basevar="$base";; *) basevar="MTL_ALLOW_DIAG_LIB=1 MTL_LIB_PATH=$base";; esac; rx=$2; wl=${3:-gpt2s_B32_L1024_C12}; n=${4:-3}
steps=20; warm=5; [ "$wl" = "gpt2s_B32_L1024_C12" ] || { steps=5; warm=2; }
show() { python -c "
import sys, json, re
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
ks = ' '.join(f\"{k['kernel'].split('(')[0][:44]}={k['avg_us']}\" for k in d['kernel_instances'] if re.search(r'''$rx''', k['kernel']))
print(f\"$1 {d['ms_per_step']:9.3f} ms/step   {ks}\")"; }
for i in $(seq $n); do
  env $basevar python bench.py --workload $wl --steps $steps --warmup $warm --no-cpu-baseline --no-extra-configs --no-live-traffic 2>/dev/null | show "base"
  python bench.py --workload $wl --steps $steps --warmup $warm --no-cpu-baseline --no-extra-configs --no-live-traffic 2>/dev/null | show "new "
done
