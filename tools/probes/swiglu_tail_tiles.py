"""GPU probe (tools/, not product): the tail launch of the whole-rounds column split of the fused SwiGLU GEMM (Llama-2 gate|up with the prompt-row
cache: [4096 x 1536 x 4096] after 5 whole rounds of 256 x 256 tiles) under different tile configurations, cold operands.
usage: python tools/probes/swiglu_tail_tiles.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch                                          # noqa: E402
from med_ts_llm_amd.hip import ops, _native as N     # noqa: E402

BF16 = torch.bfloat16
g = torch.Generator().manual_seed(0)
M, Nn, K = 4096, 1536, 4096
A = torch.randn(M, K, generator=g).to(BF16).cuda()
B = (torch.randn(Nn, K, generator=g) * 0.02).to(BF16).cuda()
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")


def run(cfg, epi):
    out = torch.empty((M, Nn // 2 if epi == N.EPI_SWIGLU else Nn), dtype=BF16, device="cuda")
    aux = torch.empty((M, Nn), dtype=BF16, device="cuda") if epi == N.EPI_SWIGLU else None
    def go():
        with ops.gemm_tune(bm=cfg[0], bn=cfg[1], stages=cfg[2], waves=cfg[3]):
            if epi == N.EPI_SWIGLU:
                ops.gemm_nt(A, B, out=aux, epilogue=epi, aux_out=out)
            else:
                ops.gemm_nt(A, B, out=out, epilogue=epi)
    try:
        go()
    except Exception as e:                      # noqa: BLE001
        return None, str(e)[:60]
    torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); go(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], (aux if aux is not None else out).float().abs().mean().item()


for name, epi in (("SWIGLU", N.EPI_SWIGLU), ("STORE", N.EPI_STORE)):
    for cfg in ((256, 128, 3, 16), (256, 96, 3, 8), (256, 96, 2, 8), (128, 192, 2, 8), (256, 192, 2, 8), (128, 96, 3, 4), (128, 128, 2, 8)):
        t, chk = run(cfg, epi)
        print(f"{name:7s} tile {cfg}: {t if t is None else round(t, 1)} us   (check {chk})", flush=True)
