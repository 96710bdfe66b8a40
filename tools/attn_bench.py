#!/usr/bin/env python
"""GPU micro-benchmark of cs_attn_fwd on the teacher's shape (512 crops x 12 heads x 197 tokens); see profiles/r01_n_*.
env: CS_ATTN_DBG (ablation bits), CS_ATTN_LDSPAD, CS_ATTN_DEBUG.   usage: python tools/attn_bench.py [crops]"""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clipself_amd.hip import HipOps
ops = HipOps()
B, N, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 512), 197, 12
C = H * 64
qkv = torch.randn(B * N, 3 * C, device="cuda").to(torch.bfloat16)
g = 14                                                   # separable tables as rope.py:118-142 builds them
freqs = 10000.0 ** (-torch.arange(0, 32, 2).float() / 32)
ang = (torch.arange(g).float() / g * 16)[:, None] * freqs[None, :]
ang = ang.repeat_interleave(2, dim=-1)
full = torch.cat([ang[:, None, :].expand(g, g, 32), ang[None, :, :].expand(g, g, 32)], dim=-1).reshape(g * g, 64)
cos, sin = full.cos().contiguous().cuda(), full.sin().contiguous().cuda()
out = torch.empty(B * N, C, dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    ops.attn_fwd(qkv, cos, sin, out, None, B, N, H, 0.125)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.attn_fwd(qkv, cos, sin, out, None, B, N, H, 0.125)
e1.record(); torch.cuda.synchronize()
print("attn_fwd us:", e0.elapsed_time(e1) * 50)
if len(sys.argv) > 2 and sys.argv[2] == "bwd":          # python tools/attn_bench.py 64 bwd : the student's backward (dsum prep + dQ + dK/dV kernels)
    lse = torch.empty(B * H, N, device="cuda")
    ops.attn_fwd(qkv, cos, sin, out, lse, B, N, H, 0.125)
    dout = torch.randn_like(out)
    dqkv = torch.empty_like(qkv)
    ws = torch.empty(ops.attn_bwd_workspace(B, N, H), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        ops.attn_bwd(qkv, out, dout, lse, cos, sin, dqkv, ws, B, N, H, 0.125)
    e0.record()
    for _ in range(20):
        ops.attn_bwd(qkv, out, dout, lse, cos, sin, dqkv, ws, B, N, H, 0.125)
    e1.record(); torch.cuda.synchronize()
    print("attn_bwd us:", e0.elapsed_time(e1) * 50)
