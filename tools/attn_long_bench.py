#!/usr/bin/env python
"""cs_attn_fwd (with lse) and cs_attn_bwd on a g x g (+CLS) token grid: microseconds per launch and the share of the MFMA peak (forward 4 N^2 d,
backward 10 N^2 d FLOPs per head), for the round-1 kernels (CS_ATTN_FWD_V1 / CS_ATTN_BWD_V1, read per launch) and the current ones, interleaved.
usage (GPU box): python tools/attn_long_bench.py [images [grid [heads [reps]]]]     e.g. 2 64 12 (the recipe's 4097 tokens), 64 14 12, 16 24 16"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clipself_amd.hip import HipOps  # noqa: E402

ops = HipOps()
argv = sys.argv[1:]
B = int(argv[0]) if argv else 2
g = int(argv[1]) if len(argv) > 1 else 64
H = int(argv[2]) if len(argv) > 2 else 12
REP = int(argv[3]) if len(argv) > 3 else 10
N, C = g * g + 1, H * 64
qkv = torch.randn(B * N, 3 * C, device="cuda").to(torch.bfloat16)
freqs = 10000.0 ** (-torch.arange(0, 32, 2).float() / 32)
ang = (torch.arange(g).float() / g * 16)[:, None] * freqs[None, :]
ang = ang.repeat_interleave(2, dim=-1)
full = torch.cat([ang[:, None, :].expand(g, g, 32), ang[None, :, :].expand(g, g, 32)], dim=-1).reshape(g * g, 64)
cos, sin = full.cos().contiguous().cuda(), full.sin().contiguous().cuda()
out = torch.empty(B * N, C, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B * H, N, device="cuda")
dout = torch.randn_like(out)
dqkv = torch.empty_like(qkv)
ws = torch.empty(ops.attn_bwd_workspace(B, N, H), dtype=torch.uint8, device="cuda")
PEAK = 2500.0


def timed(fn):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REP


ffl, bfl = 4.0 * N * N * 64 * B * H, 10.0 * N * N * 64 * B * H
for rnd in range(2):
    for ver in ("v1", "v2"):
        for k in ("CS_ATTN_FWD_V1", "CS_ATTN_BWD_V1"):
            if ver == "v1":
                os.environ[k] = "1"
            else:
                os.environ.pop(k, None)
        f = timed(lambda: ops.attn_fwd(qkv, cos, sin, out, lse, B, N, H, 0.125))
        b = timed(lambda: ops.attn_bwd(qkv, out, dout, lse, cos, sin, dqkv, ws, B, N, H, 0.125))
        print(f"{B} images x {H} heads x {N} tokens, {ver} pass {rnd}: fwd {f:8.1f} us = {ffl / f / 1e6 / PEAK:.3f} of peak | "
              f"bwd (prep + dq + dkv) {b:8.1f} us = {bfl / b / 1e6 / PEAK:.3f} of peak", flush=True)
