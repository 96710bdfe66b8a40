#!/usr/bin/env python
"""cs_swiglu_bwd_colsum against cs_swiglu_bwd + cs_colsum_bf16 at the student's shape (12 608 rows, hidden 2048).  usage (GPU box): python tools/swiglu_bwd_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clipself_amd.hip import HipOps  # noqa: E402

ops = HipOps()
M, Hd = 12608, 2048
dh = torch.randn(M, Hd, device="cuda").bfloat16()
x12 = torch.randn(M, 2 * Hd, device="cuda").bfloat16()
dx = torch.empty(M, 2 * Hd, dtype=torch.bfloat16, device="cuda")
cs = torch.zeros(2 * Hd, device="cuda")
ws = torch.empty(ops.colsum_workspace(M, 2 * Hd), dtype=torch.uint8, device="cuda")


def t(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


fused = t(lambda: ops.swiglu_bwd_colsum(dh, x12, dx, cs, ws))
sep = t(lambda: (ops.swiglu_bwd(dh, x12, dx), ops.colsum_bf16(dx, cs, ws)))
print(f"swiglu_bwd_colsum: {fused:.1f} us ({258.2 / fused:.2f} TB/s of its 258 MB) | swiglu_bwd + colsum_bf16: {sep:.1f} us", flush=True)
