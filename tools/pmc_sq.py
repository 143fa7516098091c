#!/usr/bin/env python3
"""Per-kernel SQ / TCC summary of the `sq` and `tcc` rocprofv3 passes of tools/pmc_run.sh: MFMA-busy share of the busy cycles, LDS bank
conflict cycles per LDS-active cycle, the share of wave cycles spent waiting, L2 hit rate. usage: tools/pmc_sq.py <tag> > out.txt"""
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
    print("# launches  mfma_busy/busy  lds_conflict/lds_active  wait/wave_cycles  L2 hit   kernel")
    for kn in sorted(a, key=lambda k: -a[k].get("SQ_BUSY_CYCLES", 0)):
        r = a[kn]
        busy, wave = r.get("SQ_BUSY_CYCLES", 0), r.get("SQ_WAVE_CYCLES", 0)
        mf = r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / busy if busy else 0
        ldc = r.get("SQ_LDS_BANK_CONFLICT", 0) / r["SQ_LDS_IDX_ACTIVE"] if r.get("SQ_LDS_IDX_ACTIVE") else 0
        wt = r.get("SQ_WAIT_ANY", 0) / wave if wave else 0
        tc = t.get(kn, {})
        hit = tc.get("TCC_HIT_sum", 0) / (tc.get("TCC_HIT_sum", 0) + tc.get("TCC_MISS_sum", 0)) if tc.get("TCC_HIT_sum") else float("nan")
        print(f"{n[kn]:9d}  {mf:14.3f}  {ldc:23.3f}  {wt:16.3f}  {hit:6.3f}   {clean(kn)}")


if __name__ == "__main__":
    main(sys.argv[1])
