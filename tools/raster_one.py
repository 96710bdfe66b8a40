#!/usr/bin/env python
"""One W1|W2-shaped launch (the step's dominant kernel: norm2 + SiLU*mul + ffn_ln partials, M = crops x 197) under a given raster, N times.
For tools/raster_pmc.sh: run it under `rocprofv3 --pmc FETCH_SIZE` for the traffic and bare for the time.
usage: python tools/raster_one.py <flags hex> [crops=2048] [launches=6]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clipself_amd.hip import HipOps  # noqa: E402

flags = int(sys.argv[1], 16)
crops = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
ops = HipOps()
M, C, Hd = crops * 197, 768, 2048
BF = torch.bfloat16
xb = torch.randn(M, C, device="cuda").to(BF)
mean, rstd = torch.randn(M, device="cuda") * 0.1, torch.rand(M, device="cuda") + 0.5
W12, c12, d12 = (torch.randn(2 * Hd, C, device="cuda") * 0.05).to(BF), torch.randn(2 * Hd, device="cuda"), torch.randn(2 * Hd, device="cuda")
hid, part_h = torch.empty(M, Hd, dtype=BF, device="cuda"), torch.empty(4 * (Hd // 128), M, 2, device="cuda")
run = lambda: ops.gemm_nt_ln(xb, W12, hid, bias=d12, ln_mean=mean, ln_rstd=rstd, ln_colsum=c12, stats_part=part_h, epi=3, group=Hd, flags=flags)
for _ in range(2):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
print(f"flags {flags:#x}: {us:8.1f} us per launch ({2.0 * M * 2 * Hd * C / us / 1e6:5.0f} TF/s), algorithmic bytes "
      f"{(M * C * 2 + 2 * Hd * C * 2 + M * Hd * 2) / 1e9:.3f} GB", flush=True)
