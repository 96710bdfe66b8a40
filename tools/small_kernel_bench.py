#!/usr/bin/env python
"""Timings of the teacher's small stream kernels at the benchmark's size (2048 crops): cs_ln_stats_finalize with 12 and 64 slices, cs_attn_cls_fwd.
usage (GPU box): python tools/small_kernel_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clipself_amd.hip import HipOps  # noqa: E402

ops = HipOps()
B, N, C, H = 2048, 197, 768, 12
M = B * N


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
for P, npp, cols in ((12, 64, 768), (64, 32, 2048)):
    part = torch.rand(P, M, 2, device="cuda")
    us = timeit(lambda: ops.ln_stats_finalize(part, npp, cols, mean, rstd))
    print(f"ln_stats_finalize P={P}: {us:6.1f} us ({part.numel() * 4 / us / 1e6:5.2f} TB/s)")
    # even / odd row counts agree bit for bit on their common rows (the two-rows-per-thread form needs an even M)
    m2, r2 = torch.empty(M - 1, device="cuda"), torch.empty(M - 1, device="cuda")
    ops.ln_stats_finalize(part[:, :M - 1].contiguous(), npp, cols, m2, r2)
    ops.ln_stats_finalize(part, npp, cols, mean, rstd)
    assert torch.equal(m2, mean[:M - 1]) and torch.equal(r2, rstd[:M - 1]), "two-row and one-row forms differ"
q = torch.randn(B, C, device="cuda").bfloat16()
kv = torch.randn(M, 2 * C, device="cuda").bfloat16()
cos, sin = torch.rand(N - 1, 64, device="cuda"), torch.rand(N - 1, 64, device="cuda")
out = torch.empty(B, C, dtype=torch.bfloat16, device="cuda")
us = timeit(lambda: ops.attn_cls_fwd(q, kv, cos, sin, out, B, N, H, 0.125))
print(f"attn_cls_fwd: {us:6.1f} us ({kv.numel() * 2 / us / 1e6:5.2f} TB/s)")
