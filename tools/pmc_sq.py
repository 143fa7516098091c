#!/usr/bin/env python3
"""Per-kernel SQ / TCC summary of the `sq` and `tcc` rocprofv3 passes of tools/pmc_run.sh: MFMA utilisation (0..1), LDS bank conflict cycles
per LDS-active cycle, the share of wave cycles spent waiting, L2 hit rate. usage: tools/pmc_sq.py <tag> > out.txt

MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles). Units calibrated on one GEMM of known MFMA count
(tools/pmc_calib.sh, profiles/r04_pmc_calibration.txt): SQ_VALU_MFMA_BUSY_CYCLES is the SUM over all SIMDs in shader cycles (exactly
16 x the number of v_mfma_f32_16x16x32_bf16 wave instructions), SQ_BUSY_CYCLES is summed over the 32 shader engines (8 XCDs x 4),
GRBM_GUI_ACTIVE over the 8 XCDs. Hence util = MFMA_BUSY / (32 x SQ_BUSY_CYCLES) [= MFMA_BUSY / (128 x GRBM_GUI_ACTIVE) when that counter
is in the pass]. r03's column printed MFMA_BUSY / SQ_BUSY_CYCLES un-normalised (9.7 - 13.5 = 0.30 - 0.42)."""
import glob
import sqlite3
import sys
from collections import defaultdict


def load(db):
    c = sqlite3.connect(db)
    agg = defaultdict(lambda: defaultdict(float))
    n = defaultdict(set)
    for kn, cn, v, did in c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        agg[kn][cn] += v
        n[kn].add(did)
    return agg, {k: len(v) for k, v in n.items()}


def clean(kn):
    return kn.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:78]


def main(tag):
    sq = glob.glob(f"gpurun_out/pmc_{tag}_sq/**/*.db", recursive=True)
    tcc = glob.glob(f"gpurun_out/pmc_{tag}_tcc/**/*.db", recursive=True)
    a, n = load(sq[0]) if sq else ({}, {})
    t, _ = load(tcc[0]) if tcc else ({}, {})
    print("# launches  mfma_util(0..1)  mfma_util_grbm  lds_conflict/lds_active  wait/wave_cycles  L2 hit   kernel")
    for kn in sorted(a, key=lambda k: -a[k].get("SQ_BUSY_CYCLES", 0)):
        r = a[kn]
        busy, wave = r.get("SQ_BUSY_CYCLES", 0), r.get("SQ_WAVE_CYCLES", 0)
        mf = r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (32.0 * busy) if busy else 0
        grbm = r.get("GRBM_GUI_ACTIVE", 0) or t.get(kn, {}).get("GRBM_GUI_ACTIVE", 0)
        mfg = r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (128.0 * grbm) if grbm else float("nan")
        ldc = r.get("SQ_LDS_BANK_CONFLICT", 0) / r["SQ_LDS_IDX_ACTIVE"] if r.get("SQ_LDS_IDX_ACTIVE") else 0
        wt = r.get("SQ_WAIT_ANY", 0) / wave if wave else 0
        tc = t.get(kn, {})
        hit = tc.get("TCC_HIT_sum", 0) / (tc.get("TCC_HIT_sum", 0) + tc.get("TCC_MISS_sum", 0)) if tc.get("TCC_HIT_sum") else float("nan")
        print(f"{n[kn]:9d}  {mf:15.3f}  {mfg:14.3f}  {ldc:23.3f}  {wt:16.3f}  {hit:6.3f}   {clean(kn)}")


if __name__ == "__main__":
    main(sys.argv[1])
