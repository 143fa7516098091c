"""GPU probe (tools/, not product): gemm_nt_w4_kernel (256 x 256 / 4 waves, hand-placed k-loop) against fp32 math, against the 8-wave kernel, and —
timing only — against the vendor's assembly GEMM behind torch.matmul, on cold operands. usage: python tools/probes/gemm_w4_check.py [check|time|all]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch                                          # noqa: E402
from med_ts_llm_amd.hip import ops, _native as N     # noqa: E402

BF16 = torch.bfloat16
what = sys.argv[1] if len(sys.argv) > 1 else "all"
g = torch.Generator().manual_seed(0)


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


def run(A, B, w4, **kw):
    with ops.gemm_tune(bm=256, bn=256, stages=2, waves=4 if w4 else 8):
        return ops.gemm_nt(A, B, **kw)


if what in ("check", "all"):
    bad = 0
    for M, Nn, K in [(256, 256, 64), (256, 256, 128), (256, 256, 192), (512, 768, 320), (1024, 512, 4096), (4096, 4096, 1024)]:
        A = torch.randn(M, K, generator=g).to(BF16).cuda()
        B = (torch.randn(Nn, K, generator=g) * 0.05).to(BF16).cuda()
        ref = A.float() @ B.float().t()
        bias = torch.randn(Nn, generator=g).cuda()
        o4 = run(A, B, True, out_dtype=torch.float32)
        e = rel(o4, ref)
        ob = run(A, B, True, bias=bias)
        eb = rel(ob, ref + bias)
        res = torch.randn(M, Nn, generator=g).cuda()
        orr = run(A, B, True, out_dtype=torch.float32, bias=bias, epilogue=N.EPI_RESID, aux_in=res)
        o8r = run(A, B, False, out_dtype=torch.float32, bias=bias, epilogue=N.EPI_RESID, aux_in=res)
        er = rel(orr, o8r)
        ok = e < 2e-5 and eb < 3e-3 and er < 1e-4
        bad += not ok
        print(f"[{M}x{Nn}x{K}] f32 store vs fp32 math {e:.2e} | bf16+bias {eb:.2e} | resid vs 8-wave kernel {er:.2e}  {'ok' if ok else 'FAIL'}", flush=True)
    # row-mapped A (the pruned backward's operand), SwiGLU / dSwiGLU epilogues against the 8-wave kernel
    M, Nn, K = 1024, 1024, 512
    Abig = torch.randn(8 * 384, K, generator=g).to(BF16).cuda()
    B = (torch.randn(Nn, K, generator=g) * 0.05).to(BF16).cuda()
    a_rows = (128, 384, 256)                    # rows 256..383 of each group of 384
    o4 = run(Abig, B, True, M=M, a_rows=a_rows, out_dtype=torch.float32)
    idx = torch.cat([torch.arange(256, 384) + 384 * i for i in range(8)]).cuda()
    e = rel(o4, Abig[idx].float() @ B.float().t())
    print(f"row-mapped A: {e:.2e} {'ok' if e < 2e-5 else 'FAIL'}"); bad += e >= 2e-5
    A = torch.randn(M, K, generator=g).to(BF16).cuda()
    act4, act8 = torch.empty(M, Nn // 2, dtype=BF16, device="cuda"), torch.empty(M, Nn // 2, dtype=BF16, device="cuda")
    s4 = run(A, B, True, epilogue=N.EPI_SWIGLU, aux_out=act4)
    s8 = run(A, B, False, epilogue=N.EPI_SWIGLU, aux_out=act8)
    e1, e2 = rel(s4, s8), rel(act4, act8)
    print(f"SwiGLU: pre-activations {e1:.2e}, activation {e2:.2e} {'ok' if max(e1, e2) < 3e-3 else 'FAIL'}"); bad += max(e1, e2) >= 3e-3
    gu = torch.randn(M, 2 * Nn, generator=g).to(BF16).cuda()
    d4 = run(A, B, True, epilogue=N.EPI_DSWIGLU, aux_in=gu, out=torch.empty(M, 2 * Nn, dtype=BF16, device="cuda"))
    d8 = run(A, B, False, epilogue=N.EPI_DSWIGLU, aux_in=gu, out=torch.empty(M, 2 * Nn, dtype=BF16, device="cuda"))
    e = rel(d4, d8)
    print(f"dSwiGLU: {e:.2e} {'ok' if e < 3e-3 else 'FAIL'}"); bad += e >= 3e-3
    res = torch.randn(M, Nn, generator=g).cuda()
    r4 = run(A, B, True, out_dtype=torch.float32, epilogue=N.EPI_RESID, aux_in=res, drop=(0.1, 1234))
    r8 = run(A, B, False, out_dtype=torch.float32, epilogue=N.EPI_RESID, aux_in=res, drop=(0.1, 1234))
    e = rel(r4, r8)
    print(f"residual + dropout: {e:.2e} {'ok' if e < 1e-5 else 'FAIL'}"); bad += e >= 1e-5
    # race screen: the same launch 20 times must give the same bits
    A = torch.randn(4096, 4096, generator=g).to(BF16).cuda()
    B = (torch.randn(4096, 4096, generator=g) * 0.05).to(BF16).cuda()
    first = run(A, B, True)
    same = all(torch.equal(first, run(A, B, True)) for _ in range(20))
    print(f"20 repeats bit-identical: {same}"); bad += not same
    print("CHECK", "FAILED" if bad else "PASSED", flush=True)

if what in ("time", "all"):
    SHAPES = [("llama qkv  (cached) ", 4096, 12288, 4096), ("llama o / dx class  ", 4096, 4096, 4096), ("llama d(gate|up)->dx", 4096, 4096, 22016),
              ("llama gate|up plain ", 4096, 22016, 4096), ("llama down plain    ", 4096, 4096, 11008), ("llama3 qkv          ", 4096, 6144, 4096),
              ("gpt2 fc (no GELU)   ", 8192, 3072, 768), ("gpt2 proj (no resid)", 8192, 768, 3072)]
    flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")

    def timed(fn, n=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(n):
            fn(it)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    print(f"{'shape':22s} {'M':>6s} {'N':>6s} {'K':>6s} | {'8-wave 256x256 (r5)':>22s} | {'4-wave hand-placed':>22s} | {'torch.matmul (vendor)':>22s} | vendor/w4")
    for name, M, Nn, K in SHAPES:
        A = torch.randn(M, K, generator=g).to(BF16).cuda()
        B = (torch.randn(Nn, K, generator=g) * 0.05).to(BF16).cuda()
        nA = max(2, min(8, (600 << 20) // (A.numel() * 2)))
        nB = max(2, min(24, (600 << 20) // (B.numel() * 2)))
        poolA, poolB = [A.clone() for _ in range(nA)], [B.clone() for _ in range(nB)]
        poolC = [torch.empty(M, Nn, dtype=BF16, device="cuda") for _ in range(4)]
        t8, t4, tv, ta = [], [], [], []
        for rnd in range(5):
            flush.fill_(rnd)
            t8.append(timed(lambda it: run(poolA[it % nA], poolB[it % nB], False, out=poolC[it % 4])))
            flush.fill_(rnd + 1)
            ta.append(timed(lambda it: ops.gemm_nt(poolA[it % nA], poolB[it % nB], out=poolC[it % 4])))
            flush.fill_(rnd + 3)
            t4.append(timed(lambda it: run(poolA[it % nA], poolB[it % nB], True, out=poolC[it % 4])))
            flush.fill_(rnd + 7)
            tv.append(timed(lambda it: torch.matmul(poolA[it % nA], poolB[it % nB].t(), out=poolC[it % 4])))
        a, b, c, auto = sorted(t8)[2], sorted(t4)[2], sorted(tv)[2], sorted(ta)[2]
        fl = 2.0 * M * Nn * K
        print(f"{name:22s} {M:6d} {Nn:6d} {K:6d} | {a:8.1f} us {fl / a / 1e6:7.0f} TF/s | {b:8.1f} us {fl / b / 1e6:7.0f} TF/s | {c:8.1f} us {fl / c / 1e6:7.0f} TF/s | {b / c:5.2f} | automatic dispatch {auto:8.1f} us = {auto / c:4.2f} x vendor", flush=True)
