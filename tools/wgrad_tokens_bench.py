#!/usr/bin/env python
"""Token-major wgrad (cs_gemm_wgrad_tn + split-K reduction) at token counts around a multiple of 64: the exact-tile kernel vs the ragged one
(the recipe's 2 x 4097 = 8194 tokens are 128 K tiles + 2 tokens).  usage (GPU box): python tools/wgrad_tokens_bench.py [tokens ...]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clipself_amd.hip import HipOps  # noqa: E402
ops = HipOps()
shapes = {"W1|W2": (4096, 768), "W3": (768, 2048), "q|k|v": (2304, 768), "proj": (768, 768)}
toks = [int(a) for a in sys.argv[1:]] or [8192, 8194, 12608]
for T in toks:
    tot = 0.0
    line = f"{T:6d} tokens:"
    for name, (N, K) in shapes.items():
        dY = torch.randn(T, N, device="cuda").to(torch.bfloat16)
        X = torch.randn(T, K, device="cuda").to(torch.bfloat16)
        dW = torch.zeros(N, K, device="cuda")
        ws = torch.empty(max(ops.gemm_wgrad_tn_workspace(N, K, T), 16), dtype=torch.uint8, device="cuda")
        for _ in range(3):
            ops.gemm_wgrad_tn(dY, X, dW, ws)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm_wgrad_tn(dY, X, dW, ws)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        tot += us
        dW.zero_()
        ops.gemm_wgrad_tn(dY, X, dW, ws)                  # (the kernel accumulates into dW)
        ref = dY[:, :64].float().T @ X[:, :64].float()
        err = float((dW[:64, :64] - ref).norm() / ref.norm())
        line += f"  {name} {us:6.1f} us (err {err:.1e})"
    print(line + f"  | block total {tot:6.1f} us", flush=True)
