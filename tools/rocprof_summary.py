#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats sqlite DB (ROCm 7.2 default output) as a per-kernel text table."""
import sqlite3
import sys


def main(db, out=None, steps=None):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc"))
    tot = sum(r[2] for r in rows)
    lines = [f"# rocprofv3 kernel-trace summary of {db}", f"# total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches"
             + (f" ({tot / 1e6 / steps:.3f} ms per step over {steps} steps)" if steps else ""),
             f"{'pct':>7} {'total_ms':>10} {'calls':>7} {'avg_us':>9} {'min_us':>9} {'max_us':>9}  kernel"]
    for n, calls, dur, avg, mn, mx in rows:
        lines.append(f"{100 * dur / tot:6.2f}% {dur / 1e6:10.3f} {calls:7d} {avg / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f}  {n[:160]}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, int(sys.argv[3]) if len(sys.argv) > 3 else None)
