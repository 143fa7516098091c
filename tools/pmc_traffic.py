#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE rocprofv3 passes (tools/pmc_run.sh) into HBM bytes per launch per kernel.

GUIDE MI355X_MICROARCH §HBM: the counters are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream
(TCC_EA0_RDREQ tallied at 64 B for 128-B requests) -> x2 on the read side. WRITE_SIZE is taken as reported."""
import glob
import json
import sqlite3
import sys
from collections import defaultdict


def per_kernel(db, counter, warmup_steps=0):
    """{kernel: (mean counter value per dispatch, dispatches)}. warmup_steps > 0: only the dispatches AFTER the warmup_steps-th optimiser
    launch count (a step ends with adam_multi_kernel) — set-up work (weight preparation, casts, the prompt-row cache build) and warm-up steps
    are dropped exactly as tools/rocprof_summary.py drops them from the kernel-trace tables, so both evidence sets share one denominator
    (r03's tables averaged over ALL dispatches of the child run: a direct_copy kernel showed 1.98 GB per launch x 145)."""
    c = sqlite3.connect(db)
    per_disp = defaultdict(float)
    name = {}
    for kn, cn, v, did in c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        if cn == counter:
            per_disp[did] += v
            name[did] = kn
    if warmup_steps:
        adam = sorted(d for d, kn in name.items() if "adam_multi_kernel" in kn)
        if len(adam) >= warmup_steps:
            cut = adam[warmup_steps - 1]
            per_disp = {d: v for d, v in per_disp.items() if d > cut}
    agg = defaultdict(lambda: [0.0, 0])
    for did, v in per_disp.items():
        a = agg[name[did]]
        a[0] += v
        a[1] += 1
    return {k: (t / n, n) for k, (t, n) in agg.items()}


def clean(kn):
    kn = kn.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return kn.split("(")[0]


def csrc_sha16():
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(root, "med-ts-llm_amd", "csrc", "*.h*")) + [os.path.join(root, "include", "medtsllm_hip.h")]):
        with open(fn, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def main(tag, out, workload="", warmup=1):
    f = glob.glob(f"gpurun_out/pmc_{tag}_fetch/**/*.db", recursive=True)
    w = glob.glob(f"gpurun_out/pmc_{tag}_write/**/*.db", recursive=True)
    fetch = per_kernel(f[0], "FETCH_SIZE", warmup) if f else {}
    write = per_kernel(w[0], "WRITE_SIZE", warmup) if w else {}
    res, lines = {}, []
    for kn in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, (0, 0))[0] * fetch.get(k, (0, 0))[1])):
        fk, n = fetch.get(kn, (0.0, 0))
        wk, _ = write.get(kn, (0.0, 0))
        rd, wr = 2.0 * fk * 1024.0, wk * 1024.0
        res[clean(kn)] = {"hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr, "dispatches": n}
        lines.append(f"{rd / 1e6:12.2f} MB read (2x FETCH_SIZE) {wr / 1e6:12.2f} MB written  per launch over {n:5d} launches  {clean(kn)[:110]}")
    # bench.py only trusts a table measured with the kernel sources it runs (same hash as bench.csrc_sha16)
    res["_meta"] = {"csrc_sha16": csrc_sha16(), "workload": workload, "tag": tag,
                    "method": "separate rocprofv3 --pmc passes (FETCH_SIZE x2 x 1024 B, WRITE_SIZE x 1024 B), mean per launch over the dispatches "
                              f"after the first {warmup} optimiser launch(es) (set-up and warm-up dropped)"}
    json.dump(res, open(out, "w"), indent=1)
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "", int(sys.argv[4]) if len(sys.argv) > 4 else 1)
