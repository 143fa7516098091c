#!/bin/bash
# usage (GPU box, repo root): bash tools/probes/gemm_memside_pmc.sh  -> gpurun_out/gemm_memside.txt
# Memory-side counters of gemm_nt_w4_kernel next to the vendor's assembly GEMM on the same shapes (separate --pmc passes, --kernel-trace only).
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_memside_$name -o r -- python $R/tools/probes/gemm_memside_run.py > $OUT/pmc_memside_$name.log 2>&1; }
run ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum
run stall TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_HIT_sum TCC_MISS_sum
run tcp TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum
run grbm GRBM_GUI_ACTIVE TCC_REQ_sum
run sq SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
cd $R
python - <<'PY' > $OUT/gemm_memside.txt
import glob, sqlite3
from collections import defaultdict
SHAPES = ["[4096 x 4096 x 4096]", "[4096 x 4096 x 22016]", "[4096 x 12288 x 4096]"]      # order of tools/probes/gemm_memside_run.py, 6 launches each
val = defaultdict(lambda: defaultdict(dict))       # kernel -> counter -> {dispatch: value}
for db in glob.glob("gpurun_out/pmc_memside_*/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    for kn, cn, v, did in c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        if "gemm_nt_w4" in kn or "Cijk" in kn:
            k = "ours   gemm_nt_w4_kernel<0,1>" if "gemm_nt_w4" in kn else "vendor " + kn[:70]
            val[k][cn][did] = val[k][cn].get(did, 0.0) + v
dur = defaultdict(list)                             # kernel -> [ns] in launch order, from the GRBM pass's own kernel trace (same launches as its GRBM_GUI_ACTIVE)
for db in glob.glob("gpurun_out/pmc_memside_grbm/**/*.db", recursive=True):
    for kn, st, en in sqlite3.connect(db).execute("select name, start, end from kernels order by start"):
        if "gemm_nt_w4" in kn or "Cijk" in kn:
            dur["ours   gemm_nt_w4_kernel<0,1>" if "gemm_nt_w4" in kn else "vendor " + kn[:70]].append(en - st)
print("# per launch, mean of 6 cold launches; separate rocprofv3 --pmc passes. Derived: L2 hit = TCC_HIT / (HIT + MISS); EA read latency = TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ")
print("# (cycles a read spends outstanding at the L2's memory-side port); L1->L2 read latency = TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ; GUI = GRBM_GUI_ACTIVE / 8 XCDs (cycles)")
for si, shape in enumerate(SHAPES):
    print(shape)
    for k in sorted(val):
        m = {}
        for cn, d in val[k].items():
            ids = sorted(d)[6 * si:6 * si + 6]
            m[cn] = sum(d[i] for i in ids) / max(1, len(ids))
        g = lambda n: m.get(n, float("nan"))
        print(f"  {k[:38]:38s} L1->L2 reads {g('TCP_TCC_READ_REQ_sum') / 1e6:7.2f} M  L2 req {g('TCC_REQ_sum') / 1e6:7.2f} M  hit {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):.3f}  "
              f"EA reads {g('TCC_EA0_RDREQ_sum') / 1e6:6.2f} M (to DRAM/MALL {g('TCC_EA0_RDREQ_DRAM_sum') / 1e6:6.2f} M)  EA latency {g('TCC_EA0_RDREQ_LEVEL_sum') / g('TCC_EA0_RDREQ_sum'):6.0f} cyc  "
              f"L1->L2 latency {g('TCP_TCC_READ_REQ_LATENCY_sum') / g('TCP_TCC_READ_REQ_sum'):5.0f} cyc  DRAM credit stalls {g('TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum'):6.0f}  "
              f"TLB misses {g('TCP_UTCL1_TRANSLATION_MISS_sum'):5.0f}  GUI {g('GRBM_GUI_ACTIVE') / 8 / 1e3:7.1f} k cycles", end="")
        print(f"  LDS conflict / active {g('SQ_LDS_BANK_CONFLICT') / max(1.0, g('SQ_LDS_IDX_ACTIVE')):.3f}  MFMA busy {g('SQ_VALU_MFMA_BUSY_CYCLES') / max(1.0, 32.0 * g('SQ_BUSY_CYCLES')):.3f}", end="")
        ds = dur.get(k, [])[6 * si:6 * si + 6]
        if ds:
            us = sum(ds) / len(ds) / 1e3
            print(f"  in {us:7.1f} us of that pass = {g('GRBM_GUI_ACTIVE') / 8 / us:6.0f} MHz")
        else:
            print()
PY
rm -rf $OUT/pmc_memside_*/
cat $OUT/gemm_memside.txt | head -80
for f in $OUT/pmc_memside_*.log; do echo "== $f"; tail -2 $f; done
