#!/bin/bash
# in-step A/B of several libraries / environment knobs: alternating bench runs inside ONE gpurun call, per-dispatch kernel times from the detail file.
# usage: tools/diag/ab5.sh "<kernel regex>" <workload> <reps> <label>=<lib.so | VAR=VALUE | -> ...     ("-" = the in-tree library, no knob)
rx=$1; wl=$2; n=$3; shift 3
steps=20; warm=5; [ "$wl" = "gpt2s_B32_L1024_C12" ] || { steps=5; warm=2; }
for i in $(seq $n); do
  for spec in "$@"; do
    label=${spec%%=*}; what=${spec#*=}
    case "$what" in -) envs="";; *.so) envs="MTL_ALLOW_DIAG_LIB=1 MTL_LIB_PATH=$what";; *) envs="$what";; esac
    env $envs python bench.py --workload $wl --steps $steps --warmup $warm --no-cpu-baseline --no-extra-configs --no-live-traffic --detail-file /tmp/_ab5.json >/dev/null 2>/tmp/_ab5.err \
      || { echo "$label FAILED"; tail -3 /tmp/_ab5.err; continue; }
    python - "$label" "$rx" <<'PY'
import json, re, sys
d = json.load(open("/tmp/_ab5.json"))
ks = "  ".join(f"{k['kernel'].split('(')[0][:48]}={k['avg_us']}" for k in (d.get("kernel_instances") or []) if re.search(sys.argv[2], k["kernel"]))
print(f"{sys.argv[1]:10s} {d['ms_per_step']:9.3f} ms/step   {ks}", flush=True)
PY
  done
done
