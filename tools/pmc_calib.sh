#!/bin/bash
# Calibrates the SQ counters' units on ONE GEMM of known MFMA count (tools/gemm_one.py, M = N = K = 8192, 256 x 256 tiles: 2 * 8192^3 FLOP =
# 67 108 864 v_mfma_f32_16x16x32_bf16 per launch), so that tools/pmc_sq.py can print an MFMA utilisation between 0 and 1:
#   util = SQ_VALU_MFMA_BUSY_CYCLES / (n_simd * GRBM_GUI_ACTIVE)   if the counter is summed over the SIMDs in shader cycles.
# usage (GPU box, repo root): tools/pmc_calib.sh > gpurun_out/r04_pmc_calib.txt
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_calib
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/tools/gemm_one.py 8192 8192 8192 1 256 256 2 8"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -d $OUT/a -o r -- python $ARGS > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $OUT/b -o r -- python $ARGS > $OUT/b.log 2>&1
cd $R
python - <<PY
import glob, sqlite3
from collections import defaultdict
for tag in "ab":
    dbs = glob.glob("gpurun_out/pmc_calib/%s/**/*.db" % tag, recursive=True)
    if not dbs:
        print(tag, "no db:", open("gpurun_out/pmc_calib/%s.log" % tag).read()[-600:]); continue
    c = sqlite3.connect(dbs[0])
    agg, n = defaultdict(lambda: defaultdict(float)), defaultdict(set)
    for kn, cn, v, did in c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        if "gemm_nt" in kn:
            agg[kn][cn] += v; n[kn].add(did)
    dur = {}
    try:
        for name, s, e in c.execute("select name, start, end from kernels"):
            if "gemm_nt" in name: dur.setdefault(name, []).append(e - s)
    except Exception as ex:
        print("no kernel table", ex)
    for kn in agg:
        k = len(n[kn])
        print("pass", tag, kn[:70], "launches", k, "avg_us", (sum(dur.get(kn, [0])) / max(len(dur.get(kn, [1])), 1)) / 1e3)
        for cn, v in sorted(agg[kn].items()):
            print("   %-32s per launch %.6g" % (cn, v / k))
print("expected per launch: MFMA instructions (wave level) = 2*8192^3/16384 = %d ; x16 cycles = %.6g" % (2 * 8192**3 // 16384, 2 * 8192**3 / 16384 * 16))
PY
