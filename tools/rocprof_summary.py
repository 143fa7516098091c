#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace sqlite DB (ROCm 7.2 default output) as a per-kernel text table.

usage: rocprof_summary.py <db> [out.txt] [timed_steps] [warmup_steps]
With warmup_steps > 0 only the dispatches of the TIMED steps are counted: a step ends with its optimiser kernel
(adam_multi_kernel), so everything up to the end of the warmup_steps-th optimiser dispatch is dropped — the same region
bench.py times and brackets, which is what makes the two comparable."""
import sqlite3
import statistics
import sys
from collections import defaultdict


def main(db, out=None, steps=None, warmup=0):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, start, end from kernels order by start"))
    cut, dropped = None, 0
    if warmup:
        adam_ends = [e for n, s, e in rows if "adam_multi_kernel" in n]
        if len(adam_ends) >= warmup:
            cut = adam_ends[warmup - 1]
            dropped = sum(1 for r in rows if r[1] <= cut)
            rows = [r for r in rows if r[1] > cut]
    per = defaultdict(list)
    for n, s, e in rows:
        per[n].append(e - s)
    tot = sum(sum(v) for v in per.values())
    lines = [f"# rocprofv3 kernel-trace summary of {db}",
             f"# total kernel time {tot / 1e6:.3f} ms over {len(rows)} dispatches"
             + (f" ({tot / 1e6 / steps:.3f} ms per step over {steps} timed steps)" if steps else "")
             + (f"; {dropped} dispatches of the {warmup} warm-up steps dropped" if cut is not None else ""),
             f"{'pct':>7} {'total_ms':>10} {'calls':>7} {'avg_us':>9} {'median_us':>9} {'min_us':>9} {'max_us':>9}  kernel"]
    for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"{100 * sum(v) / tot:6.2f}% {sum(v) / 1e6:10.3f} {len(v):7d} {sum(v) / len(v) / 1e3:9.2f} {statistics.median(v) / 1e3:9.2f} "
                     f"{min(v) / 1e3:9.2f} {max(v) / 1e3:9.2f}  {n[:160]}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    a = sys.argv
    main(a[1], a[2] if len(a) > 2 else None, int(a[3]) if len(a) > 3 else None, int(a[4]) if len(a) > 4 else 0)
