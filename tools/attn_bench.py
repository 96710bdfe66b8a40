import sys, torch
sys.path.insert(0, "/root/repo")
from clipself_amd.hip import HipOps
from oracle.eva_ref import rope_tables
ops = HipOps()
B, N, H = 512, 197, 12
C = H * 64
qkv = torch.randn(B * N, 3 * C, device="cuda").to(torch.bfloat16)
cos, sin = [t.cuda() for t in rope_tables(14, 64)]
out = torch.empty(B * N, C, dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    ops.attn_fwd(qkv, cos, sin, out, None, B, N, H, 0.125)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.attn_fwd(qkv, cos, sin, out, None, B, N, H, 0.125)
e1.record(); torch.cuda.synchronize()
print("attn_fwd us:", e0.elapsed_time(e1) * 50)
