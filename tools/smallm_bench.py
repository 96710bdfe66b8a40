#!/usr/bin/env python
"""cs_gemm_nt schedules on the GEMM shapes of a SMALL per-GPU batch -- the reference recipe's 2 images x 4097 tokens = 8194 rows (33 row tiles
of 256), where the launch is tile-count-bound rather than MFMA-bound: the heuristic's choice (cfg 0) against every forced schedule.
usage (GPU box): python tools/smallm_bench.py [rows]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clipself_amd.hip import HipOps  # noqa: E402

BF = torch.bfloat16
ops = HipOps()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8194
shapes = [("qkv fwd   N=2304 K=768  bf16", 2304, 768, 0), ("W1|W2 fwd N=4096 K=768  bf16", 4096, 768, 0), ("proj fwd  N=768  K=768  resid", 768, 768, 2),
          ("W3 fwd    N=768  K=2048 resid", 768, 2048, 2), ("dgrad W12 N=768  K=4096 bf16", 768, 4096, 0), ("dgrad qkv N=768  K=2304 bf16", 768, 2304, 0),
          ("dgrad W3  N=2048 K=768  bf16", 2048, 768, 0), ("dgrad prj N=768  K=768  bf16", 768, 768, 0)]
for name, N, K, epi in shapes:
    A = torch.randn(M, K, device="cuda").to(BF)
    B = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    bias = torch.randn(N, device="cuda")
    C = torch.randn(M, N, device="cuda") if epi == 2 else torch.empty(M, N, dtype=BF, device="cuda")
    extra = C if epi == 2 else None
    line = f"{name} M={M}:"
    for cfg in (0, 1, 2, 3, 9, 11, 0):
        flags = cfg << 4
        try:
            for _ in range(3):
                ops.gemm_nt(A, B, C, bias, extra, epi=epi, flags=flags)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                ops.gemm_nt(A, B, C, bias, extra, epi=epi, flags=flags)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 30
            line += f"  cfg{cfg}: {us:6.1f} us ({2.0 * M * N * K / us / 1e6:4.0f} TF/s)"
        except RuntimeError as e:
            line += f"  cfg{cfg}: n/a"
    print(line, flush=True)
