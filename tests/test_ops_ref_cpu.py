"""CPU: the per-kernel references (oracle/ops_ref.py) are themselves checked against torch autograd / the
monolithic oracle pieces, so that the `-m gpu` kernel tests have a trustworthy yardstick."""
import torch
import torch.nn.functional as F

from oracle.eva_ref import apply_rope, rope_tables
from oracle.ops_ref import RefOps

BF, F32 = torch.bfloat16, torch.float32


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_layernorm_bwd_is_autograd():
    ops = RefOps()
    M, C = 37, 96
    x = torch.randn(M, C, requires_grad=True)
    gamma, beta = (1 + 0.1 * torch.randn(C)).requires_grad_(True), torch.randn(C).requires_grad_(True)
    dy = torch.randn(M, C).to(BF)
    y = F.layer_norm(x, (C,), gamma, beta, 1e-6)
    y.backward(dy.float())
    yr, mean, rstd = torch.empty(M, C, dtype=BF), torch.empty(M), torch.empty(M)
    ops.layernorm_fwd(x.detach(), gamma.detach(), beta.detach(), yr, mean, rstd)
    assert rel(yr.float(), y.detach()) < 4e-3
    dx, dg, db = torch.empty(M, C), torch.zeros(C), torch.zeros(C)
    ops.layernorm_bwd(dy, x.detach(), gamma.detach(), mean, rstd, dx, 1, dg, db)
    assert rel(dx, x.grad) < 1e-5 and rel(dg, gamma.grad) < 1e-5 and rel(db, beta.grad) < 1e-5


def test_attention_ref_is_autograd():
    ops = RefOps()
    B, g, H = 2, 4, 2
    Ntok, C = g * g + 1, H * 64
    cos, sin = rope_tables(g, 64)
    qkv = torch.randn(B * Ntok, 3 * C, dtype=torch.float64).requires_grad_(True)
    t = qkv.reshape(B, Ntok, 3, H, 64).permute(2, 0, 3, 1, 4)
    q, k, v = apply_rope(t[0], cos.double(), sin.double()), apply_rope(t[1], cos.double(), sin.double()), t[2]
    att = ((q * 0.125) @ k.transpose(-1, -2)).softmax(-1)
    o = (att @ v).permute(0, 2, 1, 3).reshape(B * Ntok, C)
    dout = torch.randn(B * Ntok, C, dtype=torch.float64)
    o.backward(dout)
    o_r, lse = torch.empty(B * Ntok, C, dtype=BF), torch.empty(B * H, Ntok)
    ops.attn_fwd(qkv.detach().to(BF), cos, sin, o_r, lse, B, Ntok, H, 0.125)
    assert rel(o_r.float(), o.detach()) < 1e-2
    dqkv = torch.zeros(B * Ntok, 3 * C, dtype=BF)
    ops.attn_bwd(qkv.detach().to(BF), o_r, dout.to(BF), lse, cos, sin, dqkv, None, B, Ntok, H, 0.125)
    assert rel(dqkv.float(), qkv.grad) < 2e-2


def test_swiglu_l2norm_cosine_refs_are_autograd():
    ops = RefOps()
    M, Hd = 11, 16
    x12 = torch.randn(M, 2 * Hd).to(BF)
    xf = x12.float().requires_grad_(True)
    h = F.silu(xf[:, :Hd]) * xf[:, Hd:]
    dh = torch.randn(M, Hd).to(BF)
    h.backward(dh.float())
    dx = torch.empty(M, 2 * Hd, dtype=BF)
    ops.swiglu_bwd(dh, x12, dx)
    assert rel(dx.float(), xf.grad) < 4e-3
    x = torch.randn(9, 32, requires_grad=True)
    y = F.normalize(x, dim=-1)
    dy = torch.randn(9, 32)
    y.backward(dy)
    yr, inv, dxr = torch.empty(9, 32), torch.empty(9), torch.empty(9, 32, dtype=BF)
    ops.l2norm_fwd(x.detach(), yr, inv)
    ops.l2norm_bwd(dy, yr, inv, dxr)
    assert rel(yr, y.detach()) < 1e-6 and rel(dxr.float(), x.grad) < 4e-3


def test_gemm_patch_and_transpose_refs():
    ops = RefOps()
    nimg, G, N, K = 3, 4, 8, 16
    A, W = torch.randn(nimg * G, K).to(BF), torch.randn(N, K).to(BF)
    bias, pos = torch.randn(N), torch.randn(G + 1, N)
    C = torch.zeros(nimg * (G + 1), N)
    ops.gemm_nt(A, W, C, bias, pos, epi=5, group=G)
    want = (A.float() @ W.float().T + bias).reshape(nimg, G, N) + pos[1:]
    assert torch.allclose(C.reshape(nimg, G + 1, N)[:, 1:], want, atol=1e-5)
    inp = torch.randn(5, 3).to(BF)
    out = torch.empty(3, 64, dtype=BF)
    ops.transpose_bf16(inp, out)
    assert torch.equal(out[:, :5], inp.T) and float(out[:, 5:].abs().max()) == 0
