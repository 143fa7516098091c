#!/bin/bash
# usage (GPU box, repo root): tools/dp_plumbing_ab.sh -> gpurun_out/r03_dp_plumbing.txt
# N = 1 step rate with and without the data-parallel machinery live in a one-rank RCCL group (bench.py --dp-plumbing), runs alternated inside one call
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_dp_plumbing.txt
: > $OUT
line() { python - "$1" "$2" <<'PY' >> "$OUT"
import json, sys
label, path = sys.argv[1], sys.argv[2]
try:
    lines = open(path).read().strip().splitlines()
    d = json.loads([l for l in lines if l.startswith("{")][-1])
    if not lines[-1].startswith("{"):
        label += " (JSON NOT LAST on stdout)"
    tl = d.get("trainer_loop") or {}
    print(f"{label:58s} {d['value']:9.2f} samples/s  {d['ms_per_step']:9.3f} ms/step   trainer loop {tl.get('samples_per_s')}   optimizer {d.get('optimizer_ms_per_step')} ms   backend {d.get('dist_backend')}")
except Exception as e:
    print(f"{label:58s} FAILED {e!r}")
PY
}
run() { wl=$1; steps=$2; warm=$3; tag=$4; shift 4
  timeout 600 python bench.py --workload $wl --steps $steps --warmup $warm --no-cpu-baseline --no-extra-configs --no-live-traffic "$@" > /tmp/dpp_$tag.log 2> /tmp/dpp_$tag.err || tail -5 /tmp/dpp_$tag.err >> $OUT
  line "$wl $tag" /tmp/dpp_$tag.log; }
for rep in 1 2; do
  run gpt2s_B32_L1024_C12 20 5 plain_$rep
  run gpt2s_B32_L1024_C12 20 5 dp_plumbing_$rep --dp-plumbing
done
run llama2_7b_semseg_B32_L1024_C12 5 2 plain
run llama2_7b_semseg_B32_L1024_C12 5 2 dp_plumbing --dp-plumbing
run llama3_8b_recon_B32_L1024_C12 5 2 plain
run llama3_8b_recon_B32_L1024_C12 5 2 dp_plumbing --dp-plumbing
run llama2_7b_psm_B32_L2048_C25 4 2 plain
run llama2_7b_psm_B32_L2048_C25 4 2 dp_plumbing --dp-plumbing
cat $OUT
