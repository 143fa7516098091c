#!/usr/bin/env python3
"""Per-SHAPE launch times of one workload's kernels: runs bench.py with MTL_PROF_SHAPES=1 (the GEMM launcher then appends
[MxNxK] to the profiler's kernel names) and prints the `kernel_instances` table. usage: python tools/gemm_shapes.py [workload] [bench flags]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "gpt2s_B32_L1024_C12"
    extra = [a for a in sys.argv[1:] if a != wl]
    env = dict(os.environ, MTL_PROF_SHAPES="1")
    import tempfile
    detail = os.path.join(tempfile.mkdtemp(prefix="mtl_shapes_", dir="/tmp"), "detail.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--steps", "5", "--warmup", "3", "--no-cpu-baseline",
                          "--no-extra-configs", "--no-live-traffic", "--detail-file", detail] + extra, env=env, capture_output=True, text=True)
    if not os.path.exists(detail):      # (the stdout line is the compact one; the kernel instances live in the detail record)
        sys.exit(out.stdout + out.stderr)
    with open(detail) as f:
        d = json.load(f)
    print(f"# {wl}: {d['value']:.1f} {d['unit']}, {d['ms_per_step']:.3f} ms/step; per profiled step ({d.get('profiled_steps', '?')} steps)")
    steps = d.get("profiled_steps") or 5
    tot = 0.0
    for k in d["kernel_instances"]:
        us = k["avg_us"] * k["launches"] / steps
        tot += us
        rate = f"{k['tflops']:7.1f} TF/s" if "tflops" in k else f"{k['gbs']:7.1f} GB/s"
        print(f"{us:9.1f} us/step  {k['launches'] / steps:5.1f} x {k['avg_us']:8.2f} us  {rate}  {k['kernel']}")
    print(f"{tot:9.1f} us/step in profiled launches")


if __name__ == "__main__":
    main()
