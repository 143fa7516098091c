#!/usr/bin/env python3
"""Proxy experiment: the frozen GPT-2-small stack (fwd + pruned bwd) on the metric batch as ONE pass of B = 32 vs TWO half-batches on two
streams (do kernels of independent half-batches overlap each other's epilogue / main-loop phases and tails?). Run on the GPU box."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from med_ts_llm_amd.models.backbone import FrozenBackbone, random_state_dict
    hf = bench.WORKLOADS["gpt2s_B32_L1024_C12"][0]
    dev = torch.device("cuda", 0)
    sd = random_state_dict(hf, seed=0, std=0.02)
    bb = FrozenBackbone(hf, sd, dev)
    B, T, d, n_last = 32, 256, 768, 128
    h0 = torch.randn(B, T, d, device=dev) * 0.1
    dout = (torch.randn(B, n_last, d, device=dev) * 0.01).to(torch.bfloat16)
    drop = (0.1, 0.1, 1234)

    def one(hh, dd):
        out, saved = bb.run_forward(hh, n_last, keep=True, drop=drop, n_save=n_last)
        return bb.run_backward(hh, dd, saved, n_last, n_grad=n_last, drop=drop)

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    halves = [(h0[:B // 2].contiguous(), dout[:B // 2].contiguous()), (h0[B // 2:].contiguous(), dout[B // 2:].contiguous())]

    def single():
        one(h0, dout)

    def dual():
        cur = torch.cuda.current_stream()
        for s, (hh, dd) in zip((s1, s2), halves):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                one(hh, dd)
        cur.wait_stream(s1)
        cur.wait_stream(s2)

    def serial_halves():
        for hh, dd in halves:
            one(hh, dd)

    for name, fn in (("one pass B=32", single), ("two half-batches, two streams", dual), ("two half-batches, one stream", serial_halves),
                     ("one pass B=32", single), ("two half-batches, two streams", dual)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        print(f"{name:34s} {(time.perf_counter() - t0) / n * 1e3:7.3f} ms per fwd+bwd of the stack")


if __name__ == "__main__":
    main()
