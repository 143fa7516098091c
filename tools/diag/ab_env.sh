#!/bin/bash
# in-step A/B of an environment knob: alternating bench runs inside one gpurun call. usage: ab_env.sh VAR=VALUE [runs]
kv=$1; n=${2:-3}
for i in $(seq $n); do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('default   ', d['ms_per_step'])"
  env $kv python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$kv', d['ms_per_step'])"
done
