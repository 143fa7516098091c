"""YARDSTICK ONLY (tools/, never imported by the product): what the vendor's assembly GEMMs (torch.matmul -> hipBLASLt / rocBLAS) reach on the
plain-store bf16 shapes of the metric step and of the Llama-2-7B step, next to this library's hand-written kernel on the SAME shapes, operands
rotated through pools larger than L2 + MALL (as inside a training step), random bf16 operands, median of 5 rounds of 20 launches. Says how much of
the distance to the 2.5 PF/s peak a per-instruction-scheduled assembly main loop closes on these exact shapes — the tier DESIGN.md calls "not built".
usage (GPU box): python tools/probes/vendor_gemm_yardstick.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch                                          # noqa: E402
from med_ts_llm_amd.hip import ops, _native as N     # noqa: E402

BF16 = torch.bfloat16
SHAPES = [("gpt2 qkv            ", 8192, 2304, 768), ("gpt2 d(fc) -> dx    ", 4096, 768, 3072), ("gpt2 d(qkv) -> dx   ", 4096, 768, 2304),
          ("gpt2 fc (no GELU)   ", 8192, 3072, 768), ("gpt2 proj (no resid)", 8192, 768, 3072),
          ("llama qkv  (cached) ", 4096, 12288, 4096), ("llama o / dx class  ", 4096, 4096, 4096), ("llama d(gate|up)->dx", 4096, 4096, 22016),
          ("llama gate|up plain ", 4096, 22016, 4096), ("llama down plain    ", 4096, 4096, 11008)]
g = torch.Generator().manual_seed(0)
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")


def timed(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(n):
        fn(it)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"{'shape':22s} {'M':>6s} {'N':>6s} {'K':>6s} | {'this library':>22s} | {'torch.matmul (vendor asm)':>26s} | vendor / ours")
for name, M, Nn, K in SHAPES:
    A = torch.randn(M, K, generator=g).to(BF16).cuda()
    B = (torch.randn(Nn, K, generator=g) * 0.05).to(BF16).cuda()
    nA = max(2, min(8, (600 << 20) // (A.numel() * 2)))
    nB = max(2, min(24, (600 << 20) // (B.numel() * 2)))
    poolA, poolB = [A.clone() for _ in range(nA)], [B.clone() for _ in range(nB)]
    out = ops.gemm_nt(A, B, epilogue=N.EPI_STORE)
    poolC = [torch.empty_like(out) for _ in range(4)]
    ref = torch.matmul(A, B.t())
    err = float((out.float() - ref.float()).norm() / ref.float().norm())
    ours, vend = [], []
    for rnd in range(5):
        flush.fill_(rnd)
        ours.append(timed(lambda it: ops.gemm_nt(poolA[it % nA], poolB[it % nB], epilogue=N.EPI_STORE, out=poolC[it % 4])))
        flush.fill_(rnd + 7)
        vend.append(timed(lambda it: torch.matmul(poolA[it % nA], poolB[it % nB].t(), out=poolC[it % 4])))
    to, tv = sorted(ours)[2], sorted(vend)[2]
    fl = 2.0 * M * Nn * K
    print(f"{name:22s} {M:6d} {Nn:6d} {K:6d} | {to:8.1f} us {fl / to / 1e6:7.0f} TF/s | {tv:10.1f} us {fl / tv / 1e6:9.0f} TF/s | {to / tv:5.2f}   (rel diff of the two results {err:.1e})")
