"""GPU probe (tools/, not product; needs tools/diag/_lib_w4diag.so = tools/build_variant.sh w4diag only=mtl_gemm -DMTL_DIAG -DMTL_DIAG_W4VAR):
what the parts of gemm_nt_w4_kernel's k-loop cost — the full loop against the generator's ablations (no in-loop LDS-DMA / no barriers / no fragment
reads / neither DMA nor barriers / MFMAs only; the ablations compute garbage, only their time means something), cold operands, one-round shapes.
usage: MTL_ALLOW_DIAG_LIB=1 MTL_LIB_PATH=tools/diag/_lib_w4diag.so [MTL_GEMM_G=256,4] python tools/probes/gemm_w4_ablate.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch                                          # noqa: E402
from med_ts_llm_amd.hip import ops, _native as N     # noqa: E402

BF16 = torch.bfloat16
g = torch.Generator().manual_seed(0)
SHAPES = [("4096 x 4096 x 22016", 4096, 4096, 22016), ("4096 x 4096 x 4096", 4096, 4096, 4096), ("4096 x 12288 x 4096", 4096, 12288, 4096)]
VARS = [("shipped", 2), ("column-major MFMA order", 3), ("row-block-major MFMA order", 4), ("MFMA only", 5), ("slot instructions behind the MFMA pair", 6), ("local DMA", 7)]      # = the generator's VARIANTS list (tune_stages - 2)
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")


def timed(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(n):
        fn(it)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("MTL_GEMM_G =", os.environ.get("MTL_GEMM_G"))
for name, M, Nn, K in SHAPES:
    A = torch.randn(M, K, generator=g).to(BF16).cuda()
    B = (torch.randn(Nn, K, generator=g) * 0.05).to(BF16).cuda()
    nA = max(2, min(8, (600 << 20) // (A.numel() * 2)))
    nB = max(2, min(24, (600 << 20) // (B.numel() * 2)))
    poolA, poolB = [A.clone() for _ in range(nA)], [B.clone() for _ in range(nB)]
    poolC = [torch.empty(M, Nn, dtype=BF16, device="cuda") for _ in range(4)]
    res = {}
    for rnd in range(3):
        for label, st in VARS:
            flush.fill_(rnd)

            def go(it, st=st):
                with ops.gemm_tune(bm=256, bn=256, stages=st, waves=4):
                    ops.gemm_nt(poolA[it % nA], poolB[it % nB], out=poolC[it % 4])
            res.setdefault(label, []).append(timed(go))
    nkt, rounds = K // 64, (M // 256) * (Nn // 256) / 256
    line = f"{name:22s}"
    for label, _ in VARS:
        t = sorted(res[label])[1]
        line += f" | {label} {t:7.1f} us ({t / nkt / rounds * 1e3:5.0f} ns/k-tile)"
    print(line, flush=True)


# phase stamps of every workgroup's first tile (diagnostic build): entry -> k-loop begin -> k-loop end -> epilogue stores complete, 10 ns ticks;
# and the SHADER clock counter (s_memtime) around the k-loop: cycles per k-tile against the 2048 its 64 MFMAs occupy, and the clock the loop ran at
import ctypes as C                                   # noqa: E402
for name, M, Nn, K in SHAPES[:2]:
    A = torch.randn(M, K, generator=g).to(BF16).cuda()
    B = (torch.randn(Nn, K, generator=g) * 0.05).to(BF16).cuda()
    out = torch.empty(M, Nn, dtype=BF16, device="cuda")
    for vlabel, vst in (("full loop", 2), ("MFMA only", 5)):
        ws = torch.zeros(6 * 256, dtype=torch.int64, device="cuda")
        ga = N.GemmArgs()
        ga.A, ga.lda, ga.B, ga.ldb, ga.C, ga.ldc, ga.c_dtype = A.data_ptr(), K, B.data_ptr(), K, out.data_ptr(), Nn, N.MTL_BF16
        ga.M, ga.N, ga.K, ga.alpha, ga.split_k, ga.epilogue = M, Nn, K, 1.0, 1, N.EPI_STORE
        ga.workspace, ga.workspace_bytes = ws.data_ptr(), ws.numel() * 8
        ga.tune_mode, ga.tune_bm, ga.tune_bn, ga.tune_stages, ga.tune_waves = 2, 256, 256, vst, 4
        for rep in range(3):
            flush.fill_(rep)
            ops.check(N.lib().mtl_gemm_nt(C.byref(ga), ops.stream()), "mtl_gemm_nt")
            torch.cuda.synchronize()
        t = ws.view(256, 6).cpu().double()
        t0 = t[:, 0].min()
        d = lambda a: f"{float(a.mean()) / 100:6.2f} us (min {float(a.min()) / 100:6.2f}, max {float(a.max()) / 100:6.2f})"
        print(f"{name} [{vlabel}]: start skew {d(t[:, 0] - t0)} | setup {d(t[:, 1] - t[:, 0])} | k-loop {d(t[:, 2] - t[:, 1])} = "
              f"{float((t[:, 2] - t[:, 1]).mean()) * 10 / (K // 64):6.0f} ns/k-tile | epilogue {d(t[:, 3] - t[:, 2])} | end of last workgroup {float(t[:, 3].max() - t0) / 100:6.2f} us")
        cyc = t[:, 5] - t[:, 4]
        mhz = cyc / ((t[:, 2] - t[:, 1]) / 100)
        print(f"    k-loop {float(cyc.mean()) / (K // 64):7.0f} s_memtime ticks per k-tile (128 MFMAs x 16 cycles = 2048) -> MFMA issue share {2048 * (K // 64) / float(cyc.mean()):.3f} "
              f"if a tick is a shader cycle; ticks per us during the loop {float(mhz.mean()):6.0f} (min {float(mhz.min()):6.0f}, max {float(mhz.max()):6.0f})", flush=True)
