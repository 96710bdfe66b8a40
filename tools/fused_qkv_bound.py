#!/usr/bin/env python
"""Measured bound for VERDICT r4 item 2 (fused q|k|v GEMM -> attention, one workgroup per (crop, head) unit).  A unit is a 197 x 192 x 768 product.  Built on
the streaming kernel's 256 x 256 tile it is that tile with 23 % padding rows and 25 % padding columns -- i.e. exactly the q|k|v GEMM of a problem with 256 rows
per crop and 256 columns per head, which this script times next to the real one (2048 crops): the GEMM half of such a fused kernel before RoPE, the K / V images,
the softmax and the P.V products are paid.  usage (GPU box): python tools/fused_qkv_bound.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clipself_amd.hip import HipOps  # noqa: E402

ops = HipOps()
BF = torch.bfloat16


def timed(M, N, K=768, n=10):
    A = torch.randn(M, K, device="cuda").to(BF)
    B = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    bias = torch.randn(N, device="cuda")
    C = torch.empty(M, N, dtype=BF, device="cuda")
    for _ in range(3):
        ops.gemm_nt(A, B, C, bias, None, epi=0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.gemm_nt(A, B, C, bias, None, epi=0)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


real = timed(2048 * 197, 2304)
padded = timed(2048 * 256, 12 * 256)
print(f"q|k|v GEMM, 2048 crops: as launched today (403456 x 2304 x 768) {real:7.1f} us | one 256 x 256 tile per (crop, head) unit (524288 x 3072 x 768) {padded:7.1f} us "
      f"= {padded / real:.2f} x; gate of the fused kernel: 1900 us per block INCLUDING the attention (today: GEMM + ~745 us of attention)", flush=True)
