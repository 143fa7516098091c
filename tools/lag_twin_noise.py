#!/usr/bin/env python3
"""Evidence for tests/test_gpu_golden.py's lag comparison: the reference's `calcute_lags` (R:models/medtsllm.py:530-538) ranks a
circular autocorrelation, which is symmetric — corr[k] == corr[L-k] in exact arithmetic — so every lag but 0 and L/2 has a twin of
equal value. Which twin comes first in torch.topk is decided by FFT round-off (twins come out bit-identical or 1 ulp apart) and by
topk's handling of exact ties. Run on the CPU (the build container): prints, per golden sample, twin pairs that are bitwise equal /
1 ulp apart, the reference's recorded order and a stable sort's order."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

G = Path(__file__).resolve().parent.parent / "tests" / "golden"
for name in sys.argv[1:] or ["gpt2_concat_fc", "llamagqa_concat_fc"]:
    z = np.load(G / f"case_{name}.npz")
    meta = json.loads((G / f"case_{name}.json").read_text())
    x = torch.from_numpy(z["x_enc"])[:, :, 0].unsqueeze(1)
    f = torch.fft.rfft(x, dim=-1)
    corr = torch.fft.irfft(f * torch.conj(f), dim=-1).mean(1)
    L = corr.shape[-1]
    for b in range(corr.shape[0]):
        c = corr[b]
        eq = sum(bool(c[k] == c[L - k]) for k in range(1, L // 2))
        stats = next(p for p in meta["prompts"][b] if p.startswith("Input statistics"))
        print(f"{name}[{b}] L={L}: {eq} of {L // 2 - 1} twin pairs bitwise equal, the others differ by "
              f"{max(abs(float(c[k] - c[L - k])) for k in range(1, L // 2)):.2e}; reference recorded {stats[stats.index('lags are'):]!r}; "
              f"stable descending sort gives {torch.sort(c, descending=True, stable=True).indices[:5].tolist()}")
