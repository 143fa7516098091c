"""run with MTL_LIB_PATH=tools/diag/libdiag_attn_timing.so: phase cycle counts of attn_fwd_res_kernel's wave 0 per workgroup"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from med_ts_llm_amd.hip import ops
B, T, H, D = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (32, 256, 12, 64)))
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B, T, 3 * H * D, generator=g).to(torch.bfloat16).cuda()
q, k, v = qkv[..., :H * D], qkv[..., H * D:2 * H * D], qkv[..., 2 * H * D:]
for drop in ((0.0, 0), (0.1, 7)):
    for _ in range(3):
        o, lse = ops.attention_fwd(q, k, v, H, H, D, 1 / math.sqrt(D), True, dropout=drop)
    torch.cuda.synchronize()
    t = lse.flatten()[: B * H * 8].view(B * H, 8).cpu()
    names = ["load K,V + barrier", "first tile (short)", "second tile (long) + stores", "total"]
    print(f"B={B} T={T} H={H} D={D} drop={drop[0]}: cycles of wave 0, median / p10 / p90 over {B * H} workgroups")
    for i, n in enumerate(names):
        c = t[:, i].sort().values
        print(f"   {n:30s} {c[len(c) // 2]:8.0f} {c[len(c) // 10]:8.0f} {c[len(c) * 9 // 10]:8.0f}")
    t0 = t[:, 4].double() * 4096 + t[:, 5].double()
    t0 = t0 - t0.min()
    end = t0 + t[:, 3].double()
    srt = t0.sort().values
    print("   workgroup start times (ticks after the first): p10 %.0f  p50 %.0f  p75 %.0f  p90 %.0f  max %.0f ; last end %.0f" % (
        srt[len(srt) // 10], srt[len(srt) // 2], srt[len(srt) * 3 // 4], srt[len(srt) * 9 // 10], srt[-1], end.max()))
    hw = t[:, 6].long()
    cu = ((hw >> 8) & 0xf) + 16 * ((hw >> 12) & 1) + 32 * ((hw >> 13) & 7) + 1000 * t[:, 7].long()
    uniq, cnt = cu.unique(return_counts=True)
    print("   distinct (XCC, SE, SH, CU) slots used: %d ; workgroups per slot: %s" % (len(uniq), dict(zip(*[x.tolist() for x in cnt.unique(return_counts=True)]))))
