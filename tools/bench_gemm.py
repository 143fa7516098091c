"""A/B microbenchmark of the GEMM variants on the GPT-2-small / Llama layer shapes (GPU box only)."""
import sys, os, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from med_ts_llm_amd.hip import ops, _native as N

lib = N.lib()
BF16 = torch.bfloat16
shapes = [("qkv   ", 8192, 2304, 768, N.EPI_STORE), ("aproj ", 8192, 768, 768, N.EPI_RESID), ("fc    ", 8192, 3072, 768, N.EPI_GELU),
          ("mproj ", 8192, 768, 3072, N.EPI_RESID), ("b_dact", 4096, 3072, 768, N.EPI_DGELU), ("b_dxfc", 4096, 768, 3072, N.EPI_STORE), ("b_dO", 4096, 768, 768, N.EPI_STORE), ("dact  ", 8192, 3072, 768, N.EPI_DGELU), ("dx_fc ", 8192, 768, 3072, N.EPI_STORE),
          ("dx_qkv", 8192, 768, 2304, N.EPI_STORE), ("b_dxqkv", 4096, 768, 2304, N.EPI_STORE), ("llama_qkv", 8192, 12288, 4096, N.EPI_STORE), ("llama_down", 8192, 4096, 11008, N.EPI_RESID), ("llama_gateup", 8192, 22016, 4096, N.EPI_STORE), ("llama_dx", 4096, 4096, 12288, N.EPI_STORE), ("llama_oproj", 8192, 4096, 4096, N.EPI_RESID), ("llama_dgu", 4096, 11008, 4096, N.EPI_STORE),
          # the prompt-row-cached Llama step's M = 4096 shapes (round 5 sweep)
          ("l4k_qkv", 4096, 12288, 4096, N.EPI_STORE), ("l4k_sq", 4096, 4096, 4096, N.EPI_STORE), ("l4k_dqkv", 4096, 4096, 12288, N.EPI_STORE),
          ("l4k_dgu", 4096, 4096, 22016, N.EPI_STORE), ("l4k_oproj", 4096, 4096, 4096, N.EPI_RESID), ("l4k_down", 4096, 4096, 11008, N.EPI_RESID)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if s[0].strip() in sys.argv[1:]]
variants = [(1, 256, 128, 3, 16), (1, 256, 192, 2, 8), (1, 256, 256, 2, 8)]
if os.environ.get("GEMM_VARIANTS"):      # e.g. GEMM_VARIANTS="128,96,2,4;128,96,4,4"
    variants = [(1,) + tuple(int(x) for x in v.split(",")) for v in os.environ["GEMM_VARIANTS"].split(";")]
g = torch.Generator().manual_seed(0)
for name, M, Nn, K, epi in shapes:
    A = torch.randn(M, K, generator=g).to(BF16).cuda()
    B = (torch.randn(Nn, K, generator=g) * 0.05).to(BF16).cuda()
    if os.environ.get("GEMM_ZERO") == "1":      # zero-filled operands: same instruction stream at a lower power draw (the chip clocks to its power budget)
        A.zero_(); B.zero_()
    bias = torch.randn(Nn, generator=g).cuda()
    kw = {}
    if epi == N.EPI_RESID:
        kw = dict(out_dtype=torch.float32, aux_in=torch.randn(M, Nn, generator=g).cuda())
    elif epi == N.EPI_GELU:
        kw = dict(aux_out=torch.empty(M, Nn, dtype=BF16, device="cuda"))
    elif epi == N.EPI_DGELU:
        kw = dict(aux_in=torch.randn(M, Nn, generator=g).to(BF16).cuda())
    outs, times = {}, {v: [] for v in variants}
    CM = os.environ.get("GEMM_COLD", "")
    COLD = bool(CM)      # GEMM_COLD=1 (all) or any of the letters A, B, C: which operands rotate      # rotate operands/outputs through pools larger than L2 + MALL (as inside a training step)
    if COLD:
        poolA = [A.clone() for _ in range(8)]
        poolB = [B.clone() for _ in range(24)]
        flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    for rnd in range(5):
        for v in variants:
            ops._TUNE["gemm"] = (1 if v[0] == 0 else 2,) + tuple(v[1:])      # per-call fields of mtl_gemm_args (the library keeps no switch)
            try:
                out = ops.gemm_nt(A, B, bias=bias, epilogue=epi, **kw)
            except RuntimeError:
                times[v].append(float("inf")); outs[v] = None
                continue
            poolC = [torch.empty_like(out) for _ in range(8)] if COLD else None
            torch.cuda.synchronize()
            if COLD:
                flush.fill_(rnd)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            PF = os.environ.get("GEMM_PREFETCH") == "1"
            sink = torch.zeros((), dtype=torch.float32, device="cuda")
            for it in range(20):
                if COLD and PF:      # touch the weights the next GEMM will use (stand-in for a side-stream prefetch into the memory-side cache)
                    torch.sum(poolB[it % 24], dim=(0, 1), dtype=torch.float32, out=sink)
                if COLD:
                    ops.gemm_nt(poolA[it % 8] if CM in "1" or "A" in CM else A, poolB[it % 24] if CM in "1" or "B" in CM else B, bias=bias, epilogue=epi, out=poolC[it % 8] if CM in "1" or "C" in CM else out, **{k: v_ for k, v_ in kw.items() if k != "out_dtype"})
                else:
                    ops.gemm_nt(A, B, bias=bias, epilogue=epi, out=out, **{k: v_ for k, v_ in kw.items() if k != "out_dtype"})
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 20 * 1e3)
            outs[v] = out.float().clone()
    ref = outs[variants[0]]
    fl = 2.0 * M * Nn * K
    line = f"{name} M={M} N={Nn} K={K} epi={epi}: "
    for v in variants:
        t = sorted(times[v])[len(times[v]) // 2]
        if outs[v] is None:
            line += f" mode{v}: unsupported |"
            continue
        same = torch.equal(outs[v], ref)
        line += f" mode{v}: {t:7.1f}us {fl / t / 1e6:7.1f}TF {'==' if same else '!='} |"
    print(line)
ops._TUNE["gemm"] = (0, 0, 0, 0, 0)
