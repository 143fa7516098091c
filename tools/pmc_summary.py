#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc sqlite outputs per kernel name: mean counter value per dispatch."""
import glob
import sqlite3
import sys
from collections import defaultdict


def main(pattern, name_filter=None):
    for db in sorted(glob.glob(pattern, recursive=True)):
        c = sqlite3.connect(db)
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
        if "counters_collection" not in tabs:
            print(db, "no counters_collection view; tables:", [t for t in tabs if "pmc" in t or "counter" in t][:8])
            continue
        cols = [d[0] for d in c.execute("select * from counters_collection limit 1").description]
        agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
        q = "select kernel_name, counter_name, value, dispatch_id from counters_collection" if "kernel_name" in cols else None
        if q is None:
            print(db, cols)
            continue
        per_disp = defaultdict(float)
        meta = {}
        for kn, cn, v, did in c.execute(q):
            per_disp[(did, cn)] += v
            meta[did] = kn
        for (did, cn), v in per_disp.items():
            a = agg[meta[did]][cn]
            a[0] += v
            a[1] += 1
        print("#", db)
        for kn, cs in sorted(agg.items(), key=lambda kv: -sum(x[0] for x in kv[1].values())):
            if name_filter and name_filter not in kn:
                continue
            print(f"  {kn[:120]}")
            for cn, (tot, n) in sorted(cs.items()):
                print(f"      {cn:28s} mean/dispatch {tot / n:16.1f}   dispatches {n}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
