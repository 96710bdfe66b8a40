"""TEST INFRASTRUCTURE -- CPU oracle, never imported by the product path.

Per-kernel references: one plain-torch (fp32 math) restatement for every entry point of the C ABI
(include/clipself_hip.h), with the *same tensor-level signature* as clipself_amd.hip.HipOps, writing into the
caller's output tensors.  Used (a) by the `-m gpu` tests to check each HIP kernel in isolation on identical inputs
and (b) by the CPU tests to run the step engine's op decomposition (clipself_amd/engine.py) against the monolithic
oracle (oracle/eva_ref.py) -- i.e. to prove on CPU that the hand-written backward chain is the true gradient.

Reference file:line for each op is the same as listed in include/clipself_hip.h.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .roi_align_ref import roi_align_1x1

EPI_BF16, EPI_F32, EPI_RESID_F32, EPI_SWIGLU_BF16, EPI_ATOMIC_F32, EPI_PATCH_F32, EPI_RESID_LN_F32, EPI_GELU_BF16, EPI_QGELU_BF16 = range(9)
DX_BF16, DX_F32_ASSIGN, DX_F32_ACCUM = range(3)


def _rope_rows(t, cos, sin, inverse=False):
    """t [..., N, d]; rows 1.. rotated by tables [(N-1), d] (rope.py:25-29,163-164); inverse = transpose rotation."""
    body = t[..., 1:, :]
    pairs = body.reshape(*body.shape[:-1], -1, 2)
    c = cos.reshape(cos.shape[0], -1, 2)
    s = sin.reshape(sin.shape[0], -1, 2)
    x0, x1 = pairs[..., 0], pairs[..., 1]
    if not inverse:
        y0 = x0 * c[..., 0] - x1 * s[..., 0]
        y1 = x1 * c[..., 1] + x0 * s[..., 1]
    else:
        y0 = x0 * c[..., 0] + x1 * s[..., 1]
        y1 = x1 * c[..., 1] - x0 * s[..., 0]
    out = torch.stack((y0, y1), dim=-1).flatten(-2)
    return torch.cat((t[..., :1, :], out), dim=-2)


def _act(x, quick):
    """MLP activation of the OpenAI-CLIP ViT: nn.GELU (erf) or QuickGELU (open_clip/transformer.py:31-34,211)."""
    return x * torch.sigmoid(1.702 * x) if quick else F.gelu(x)


class RefOps:
    name = "ref"

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype)

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype)

    # ------------------------------------------------------------------------------------------
    def gemm_nt(self, A, B, C, bias=None, extra=None, epi=EPI_BF16, splits=1, group=0, flags=0):
        acc = A.float() @ B.float().T
        if epi == EPI_SWIGLU_BF16:
            if bias is not None:
                acc = acc + bias
            x1, x2 = acc[:, :group], acc[:, group:]
            C.copy_((F.silu(x1) * x2).to(torch.bfloat16))
            return
        if bias is not None:
            acc = acc + bias
        if epi == EPI_BF16:
            C.copy_(acc.to(torch.bfloat16))
        elif epi in (EPI_GELU_BF16, EPI_QGELU_BF16):
            C.copy_(_act(acc, epi == EPI_QGELU_BF16).to(torch.bfloat16))
        elif epi == EPI_F32:
            C.copy_(acc)
        elif epi == EPI_RESID_F32:
            C.copy_(extra + acc)
        elif epi == EPI_ATOMIC_F32:
            C.add_(acc)
        elif epi == EPI_PATCH_F32:
            M, N = acc.shape
            nimg = M // group
            out = C.reshape(nimg, group + 1, -1)
            out[:, 1:, :N] = acc.reshape(nimg, group, N) + extra[1:group + 1, :N]
        else:
            raise ValueError(epi)

    def quant_rows_fp8(self, x, q, scale):
        """cs_quant_rows_fp8: per-row amax/448 scaling, round-to-nearest-even to OCP e4m3 (torch.float8_e4m3fn), K padded to 128 with zeros.
        q is a uint8 / float8 byte buffer [M, Kp]."""
        M, K = x.shape
        xf = x.float()
        amax = xf.abs().amax(dim=1)
        inv = torch.where(amax > 0, 448.0 / amax, torch.zeros_like(amax))
        scale.copy_(torch.where(amax > 0, amax / 448.0, torch.ones_like(amax)))
        q8 = (xf * inv[:, None]).to(torch.float8_e4m3fn)
        out = q.view(torch.uint8)
        out.zero_()
        out[:, :K] = q8.view(torch.uint8)

    def gemm_nt_f8(self, A8, B8, C, row_scale, col_scale, bias=None, extra=None, epi=EPI_BF16, flags=0):
        """cs_gemm_nt_f8: exact products of the e4m3 values, fp32 accumulation, scales and bias in the epilogue."""
        a = A8.view(torch.float8_e4m3fn).float()
        b = B8.view(torch.float8_e4m3fn).float()
        acc = (a @ b.T) * row_scale[:, None] * col_scale[None, :]
        if bias is not None:
            acc = acc + bias
        if epi == EPI_BF16:
            C.copy_(acc.to(torch.bfloat16))
        elif epi == EPI_RESID_F32:
            C.copy_(extra + acc)
        else:
            raise ValueError(epi)

    def gemm_nt_ln(self, A, B, C, bias=None, extra=None, ln_mean=None, ln_rstd=None, ln_colsum=None, stats_part=None, xb_out=None,
                   epi=EPI_RESID_LN_F32, group=0, flags=0):
        acc = A.float() @ B.float().T
        if ln_mean is not None:                                    # folded LayerNorm: rstd * (acc - mean * colsum)
            acc = ln_rstd[:, None] * (acc - ln_mean[:, None] * ln_colsum[None, :])
        if bias is not None:
            acc = acc + bias
        if epi in (EPI_RESID_F32, EPI_RESID_LN_F32):
            o = extra + acc
            C.copy_(o)
            if xb_out is not None:
                xb_out[:, :o.shape[1]] = o.to(torch.bfloat16)
            if stats_part is not None:                             # per 64-column slice (sum, sum of squares) of the fp32 outputs
                for s in range((o.shape[1] + 63) // 64):
                    blk = o[:, 64 * s:64 * s + 64]
                    stats_part[s, :, 0] = blk.sum(-1)
                    stats_part[s, :, 1] = (blk * blk).sum(-1)
            return
        if epi == EPI_BF16:
            C.copy_(acc.to(torch.bfloat16))
            return
        if epi in (EPI_GELU_BF16, EPI_QGELU_BF16):
            C.copy_(_act(acc, epi == EPI_QGELU_BF16).to(torch.bfloat16))
            return
        assert epi == EPI_SWIGLU_BF16
        x1, x2 = acc[:, :group], acc[:, group:]
        C.copy_((F.silu(x1) * x2).to(torch.bfloat16))
        if stats_part is not None:                                 # per 32-column slice (sum, sum of squares) of the rounded outputs
            h = C.float()
            for s in range(stats_part.shape[0]):
                blk = h[:, 32 * s:32 * s + 32]
                stats_part[s, :, 0] = blk.sum(-1)
                stats_part[s, :, 1] = (blk * blk).sum(-1)

    @staticmethod
    def split_planes(x):
        """fp32 -> (hi bf16, lo int16): the two halves of y = bits(x) + 0x8000 (include/clipself_hip.h, cs_gemm_nt_ln_split)."""
        y = x.contiguous().view(torch.int32) + 0x8000
        hi = (y >> 16).to(torch.int16).view(torch.bfloat16)
        lo = (y & 0xFFFF).to(torch.int16)                          # wraps to the signed view of the low 16 bits
        return hi, lo

    @staticmethod
    def join_planes(hi, lo):
        y = (hi.contiguous().view(torch.int16).to(torch.int32) << 16) | (lo.to(torch.int32) & 0xFFFF)
        return (y - 0x8000).view(torch.float32)

    def gemm_nt_ln_split(self, A, B, hi, lo, bias, ln_mean, ln_rstd, ln_colsum, x_in=None, x_out=None, stats_part=None, flags=0):
        assert x_in is None or x_out is None
        acc = A.float() @ B.float().T
        acc = ln_rstd[:, None] * (acc - ln_mean[:, None] * ln_colsum[None, :]) + bias
        o = (x_in if x_in is not None else self.join_planes(hi, lo)) + acc
        if x_out is not None:
            x_out.copy_(o)
        else:
            h, l = self.split_planes(o)
            hi.copy_(h)
            lo.copy_(l)
        if stats_part is not None:
            for s in range((o.shape[1] + 63) // 64):
                blk = o[:, 64 * s:64 * s + 64]
                stats_part[s, :, 0] = blk.sum(-1)
                stats_part[s, :, 1] = (blk * blk).sum(-1)

    def crop_resize(self, image_u8, boxes, size, pad_center=True, mean=None, std=None, out=None):
        """cs_crop_resize_u8 through Pillow itself (oracle/pil_crops_ref.py)."""
        from .pil_crops_ref import OPENAI_MEAN, OPENAI_STD, pil_crops
        res = torch.from_numpy(pil_crops(image_u8.cpu().numpy(), boxes.cpu().numpy(), size, pad_center, mean or OPENAI_MEAN, std or OPENAI_STD))
        if out is not None:
            out.copy_(res)
            return out
        return res

    def resize_bilinear(self, x, size):
        return F.interpolate(x.float(), size=(size, size), mode="bilinear")

    def gemm_wgrad_workspace(self, M, N, K):
        return 16

    def gemm_wgrad(self, A, B, dW, workspace):
        dW.add_(A.float() @ B.float().T)

    def gemm_wgrad_tn_workspace(self, N, K, tokens):
        return 16 if (tokens > 0 and N >= 8 and K >= 8 and N % 8 == 0 and K % 8 == 0) else 0

    def gemm_wgrad_tn(self, dY, X, dW, workspace):
        dW.add_(dY.float().T @ X.float())

    def ln_stats_finalize(self, part, npp, C, mean, rstd, eps=1e-6):
        P = part.shape[0]
        keep = [p for p in range(P) if C - p * npp > 0]
        s = part[keep, :, 0].double().sum(0)
        q = part[keep, :, 1].double().sum(0)
        mu = s / C
        mean.copy_(mu.float())
        rstd.copy_(torch.rsqrt((q / C - mu * mu).clamp_min(0) + eps).float())

    def attn_fwd_stats(self, qkv, cos, sin, out, lse, stats_part, B, Ntok, H, scale):
        self.attn_fwd(qkv, cos, sin, out, lse, B, Ntok, H, scale)
        o = out[:, :H * 64].float().reshape(B * Ntok, H, 64)
        stats_part[:, :, 0] = o.sum(-1).T
        stats_part[:, :, 1] = (o * o).sum(-1).T

    def gelu_fwd(self, x, y, quick=False):
        y.copy_(_act(x.float(), quick).to(y.dtype))

    def gelu_bwd(self, dy, x, dx, quick=False):
        with torch.enable_grad():                                  # autograd is the derivative oracle (also when called inside a backward)
            xf = x.float().detach().requires_grad_(True)
            (g,) = torch.autograd.grad(_act(xf, quick), xf, dy.float())
        dx.copy_(g.to(dx.dtype))

    def layernorm_fwd_f32(self, x, gamma, beta, y, mean=None, rstd=None, eps=1e-5):
        mu = x.mean(-1, keepdim=True)
        var = ((x - mu) ** 2).mean(-1, keepdim=True)
        r = torch.rsqrt(var + eps)
        y.copy_((x - mu) * r * gamma + beta)
        if mean is not None:
            mean.copy_(mu[:, 0])
            rstd.copy_(r[:, 0])

    def layernorm_fwd(self, x, gamma, beta, y, mean=None, rstd=None, eps=1e-6, q8=None, q_scale=None):
        xf = x.float()
        mu = xf.mean(-1, keepdim=True)
        var = ((xf - mu) ** 2).mean(-1, keepdim=True)
        r = torch.rsqrt(var + eps)
        if y is not None:
            y.copy_(((xf - mu) * r * gamma + beta).to(torch.bfloat16))
        if q8 is not None:                                         # the e4m3 copy of y: the row quantiser applied to what y holds
            qv = q8.view(torch.uint8)
            Kp = (y.shape[1] + 127) // 128 * 128
            self.quant_rows_fp8(y, qv[:, :Kp], q_scale)
        if mean is not None:
            mean.copy_(mu[:, 0])
            rstd.copy_(r[:, 0])

    def layernorm_bwd_workspace(self, M, C):
        return 4

    def layernorm_bwd_q8(self, dy, x, gamma, mean, rstd, dx, dx_mode, dgamma, dbeta, accumulate, workspace, dx_copy, copy_colsum, q8, q_scale):
        self.layernorm_bwd(dy, x, gamma, mean, rstd, dx, dx_mode, dgamma, dbeta, accumulate, workspace, dx_copy, copy_colsum)
        C = x.shape[1]
        self.quant_rows_fp8(dx_copy, q8[:, :(C + 127) // 128 * 128], q_scale)

    def layernorm_bwd(self, dy, x, gamma, mean, rstd, dx, dx_mode, dgamma=None, dbeta=None, accumulate=False, workspace=None,
                      dx_copy=None, copy_colsum=None, q8=None, q_scale=None):
        if q8 is not None:
            return self.layernorm_bwd_q8(dy, x, gamma, mean, rstd, dx, dx_mode, dgamma, dbeta, accumulate, workspace, dx_copy, copy_colsum,
                                         q8, q_scale)
        xh = (x.float() - mean[:, None]) * rstd[:, None]
        g = dy.float()
        gy = g * gamma
        m1 = gy.mean(-1, keepdim=True)
        m2 = (gy * xh).mean(-1, keepdim=True)
        d = rstd[:, None] * (gy - m1 - xh * m2)
        if dx_mode == DX_BF16:
            dx.copy_(d.to(torch.bfloat16))
        elif dx_mode == DX_F32_ASSIGN:
            dx.copy_(d)
        else:
            dx.add_(d)
        if dgamma is not None:
            if accumulate:
                dgamma.add_((g * xh).sum(0))
                dbeta.add_(g.sum(0))
            else:
                dgamma.copy_((g * xh).sum(0))
                dbeta.copy_(g.sum(0))
        if dx_copy is not None:                                    # bf16 copy of the updated rows + its column sums (a bias gradient)
            assert dx_mode != DX_BF16
            dx_copy.copy_(dx.to(torch.bfloat16))
            if copy_colsum is not None:
                cs = dx_copy.float().sum(0)
                if accumulate:
                    copy_colsum.add_(cs)
                else:
                    copy_colsum.copy_(cs)

    def layernorm_fwd_q8(self, x, gamma, beta, y, q8, q_scale, mean=None, rstd=None, eps=1e-6):
        self.layernorm_fwd(x, gamma, beta, y, mean, rstd, eps, q8=q8, q_scale=q_scale)

    def l2norm_fwd(self, x, y, inv_norm, eps=1e-12):
        inv = 1.0 / x.norm(dim=-1).clamp_min(eps)
        y.copy_(x * inv[:, None])
        if inv_norm is not None:
            inv_norm.copy_(inv)

    def l2norm_bwd(self, dy, y, inv_norm, dx):
        dot = (dy * y).sum(-1, keepdim=True)
        dx.copy_(((dy - y * dot) * inv_norm[:, None]).to(torch.bfloat16))

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _split_heads(qkv, B, Ntok, H):
        C = H * 64
        t = qkv[:, :3 * C].float().reshape(B, Ntok, 3, H, 64).permute(2, 0, 3, 1, 4)
        return t[0], t[1], t[2]

    @staticmethod
    def _r(t):
        return t.to(torch.bfloat16).float()

    def _attn_core(self, qkv, cos, sin, B, Ntok, H, scale):
        q, k, v = self._split_heads(qkv, B, Ntok, H)
        q = self._r(_rope_rows(q, cos, sin))
        k = self._r(_rope_rows(k, cos, sin))
        s = (q @ k.transpose(-1, -2)) * scale
        lse = torch.logsumexp(s, dim=-1)
        p = torch.exp(s - lse[..., None])
        return q, k, v, p, lse

    def attn_fwd(self, qkv, cos, sin, out, lse, B, Ntok, H, scale):
        q, k, v, p, l = self._attn_core(qkv, cos, sin, B, Ntok, H, scale)
        # the kernel feeds un-normalised bf16 probabilities exp(s - max) to the PV MFMA and divides by the fp32 row sum
        s = (q @ k.transpose(-1, -2)) * scale
        mx = s.max(-1, keepdim=True).values
        e = torch.exp(s - mx)
        o = (self._r(e) @ v) / e.sum(-1, keepdim=True)
        out.copy_(o.permute(0, 2, 1, 3).reshape(B * Ntok, H * 64).to(torch.bfloat16))
        if lse is not None:
            lse.copy_(l.reshape(B * H, Ntok))

    def attn_cls_fwd(self, q, kv, cos, sin, out, B, Ntok, H, scale):
        C = H * 64
        qh = q[:, :C].float().reshape(B, 1, H, 64).permute(0, 2, 1, 3)                       # CLS query: never rotated
        k = kv[:, :C].float().reshape(B, Ntok, H, 64).permute(0, 2, 1, 3)
        v = kv[:, C:2 * C].float().reshape(B, Ntok, H, 64).permute(0, 2, 1, 3)
        k = self._r(_rope_rows(k, cos, sin))
        s = (qh @ k.transpose(-1, -2)) * scale
        e = torch.exp(s - s.max(-1, keepdim=True).values)
        o = (self._r(e) @ v) / e.sum(-1, keepdim=True)
        out.copy_(o.permute(0, 2, 1, 3).reshape(B, C).to(torch.bfloat16))

    def attn_query_fwd(self, q, kv, allow, out, B, Q, Ntok, H, scale):
        """cs_attn_query_fwd: Q query rows per image against the image's keys / values, allow [B*Q, Ntok] uint8; same rounding points as
        attn_cls_fwd (fp32 scores, exp(s - max) rounded to bf16 for P.V, fp32 row sum of the unrounded ones), no rotary embedding."""
        C = H * 64
        qh = q[:, :C].float().reshape(B, Q, H, 64).permute(0, 2, 1, 3)
        k = kv[:, :C].float().reshape(B, Ntok, H, 64).permute(0, 2, 1, 3)
        v = kv[:, C:2 * C].float().reshape(B, Ntok, H, 64).permute(0, 2, 1, 3)
        s = (qh @ k.transpose(-1, -2)) * scale
        s = s.masked_fill(~allow.view(B, 1, Q, Ntok).bool(), float("-inf"))
        e = torch.exp(s - s.max(-1, keepdim=True).values)
        o = (self._r(e) @ v) / e.sum(-1, keepdim=True)
        out.copy_(o.permute(0, 2, 1, 3).reshape(B * Q, C).to(torch.bfloat16))

    def attn_bwd_workspace(self, B, Ntok, H):
        return 4

    def attn_bwd(self, qkv, o, dout, lse, cos, sin, dqkv, workspace, B, Ntok, H, scale):
        q, k, v, p, _ = self._attn_core(qkv, cos, sin, B, Ntok, H, scale)
        do = dout.float().reshape(B, Ntok, H, 64).permute(0, 2, 1, 3)
        of = o.float().reshape(B, Ntok, H, 64).permute(0, 2, 1, 3)
        dsum = (do * of).sum(-1, keepdim=True)
        dv = self._r(p).transpose(-1, -2) @ do
        dp = do @ v.transpose(-1, -2)
        ds = self._r(p * (dp - dsum) * scale)
        dq = _rope_rows(ds @ k, cos, sin, inverse=True)
        dk = _rope_rows(ds.transpose(-1, -2) @ q, cos, sin, inverse=True)
        full = torch.stack((dq, dk, dv), dim=0).permute(1, 3, 0, 2, 4).reshape(B * Ntok, 3 * H * 64)
        dqkv[:, :3 * H * 64] = full.to(torch.bfloat16)

    # ------------------------------------------------------------------------------------------
    def swiglu_fwd(self, x12, h):
        Hd = h.shape[1]
        x1, x2 = x12[:, :Hd].float(), x12[:, Hd:2 * Hd].float()
        h.copy_((F.silu(x1) * x2).to(torch.bfloat16))

    def swiglu_bwd(self, dh, x12, dx12, q8=None, q_scale=None):
        Hd = dh.shape[1]
        x1, x2, d = x12[:, :Hd].float(), x12[:, Hd:2 * Hd].float(), dh.float()
        sig = torch.sigmoid(x1)
        dx12[:, :Hd] = (d * x2 * (sig + x1 * sig * (1 - sig))).to(torch.bfloat16)
        dx12[:, Hd:2 * Hd] = (d * x1 * sig).to(torch.bfloat16)
        if q8 is not None:                       # cs_swiglu_bwd_q8: = quant_rows_fp8 of the rounded output
            self.quant_rows_fp8(dx12[:, :2 * Hd], q8[:, :(2 * Hd + 127) // 128 * 128], q_scale)

    def swiglu_bwd_colsum(self, dh, x12, dx12, colsum, workspace=None):
        self.swiglu_bwd(dh, x12, dx12)
        self.colsum_bf16(dx12[:, :2 * dh.shape[1]], colsum[:2 * dh.shape[1]])

    def swiglu_bwd_q8(self, dh, x12, dx12, q8, q_scale):
        self.swiglu_bwd(dh, x12, dx12, q8, q_scale)

    def cast_f32_bf16(self, x, y):
        y.copy_(x.to(torch.bfloat16))

    def transpose_bf16(self, inp, out):
        R = inp.shape[0]
        out.zero_()
        out[:, :R] = inp.T

    def transpose_bf16_batched(self, pairs):
        for inp, out in pairs:
            self.transpose_bf16(inp, out)

    def colsum_workspace(self, M, N):
        return 4

    def colsum_bf16(self, x, out, workspace=None):
        out.add_(x.float().sum(0))

    def im2row(self, img, out, p):
        B, _, S, _ = img.shape
        g = S // p
        patches = img.float().reshape(B, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(B * g * g, 3 * p * p)
        out[:, :3 * p * p] = patches.to(torch.bfloat16)
        out[:, 3 * p * p:] = 0                      # zero the K padding (p=14: 588 -> 640)

    def cls_row(self, x, cls, pos):
        x[:, 0, :] = cls + pos[0]

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _rois_pixels(rois, grid_h, grid_w):
        r = rois.detach().float().clone()
        r[:, [1, 3]] *= grid_w
        r[:, [2, 4]] *= grid_h
        return r

    def roialign_fwd(self, feat, rois, pooled, grid_h, grid_w, tok_off):
        B, Ntok, E = feat.shape
        fmap = feat[:, tok_off:tok_off + grid_h * grid_w].reshape(B, grid_h, grid_w, E)
        pooled.copy_(roi_align_1x1(fmap, self._rois_pixels(rois, grid_h, grid_w)))

    def roialign_bwd(self, dpooled, rois, dfeat, grid_h, grid_w, tok_off):
        B, Ntok, E = dfeat.shape
        with torch.enable_grad():          # may be called from inside an autograd backward (grad mode off)
            probe = torch.zeros(B, grid_h, grid_w, E, requires_grad=True)
            out = roi_align_1x1(probe, self._rois_pixels(rois, grid_h, grid_w))
            (g,) = torch.autograd.grad(out, probe, dpooled.detach())
        dfeat[:, tok_off:tok_off + grid_h * grid_w] += g.reshape(B, grid_h * grid_w, E)

    def cosine_loss_fwd(self, student, teacher, stats, loss, weight):
        ns, nt = student.norm(dim=-1).clamp_min(1e-12), teacher.norm(dim=-1).clamp_min(1e-12)
        cos = (student * teacher).sum(-1) / (ns * nt)
        stats[:, 0] = cos
        stats[:, 1] = 1.0 / ns
        stats[:, 2] = 1.0 / nt
        loss[0] = weight * (1.0 - cos.mean())

    def cosine_loss_bwd(self, student, teacher, stats, dstudent, weight, grad_scale=1.0, upstream=None):
        K = student.shape[0]
        if upstream is not None:
            grad_scale = grad_scale * float(upstream.reshape(-1)[0])
        cos, i_s, i_t = stats[:, 0:1], stats[:, 1:2], stats[:, 2:3]
        dstudent.copy_((-weight * grad_scale / K) * (teacher * i_t - cos * student * i_s) * i_s)

    def fed_bce_fwd(self, logits, tgt, rowloss, loss, ns, temp, weight):
        z = logits[:, :ns] * temp
        t = torch.zeros_like(z)
        rows = torch.nonzero(tgt >= 0)[:, 0]
        t[rows, tgt[rows].long()] = 1.0
        rl = F.binary_cross_entropy_with_logits(z, t, reduction="none").sum(-1)
        rowloss.copy_(rl)
        loss[0] = weight * rl.mean()

    def fed_bce_bwd(self, logits, tgt, dz, ns, temp, weight, upstream=None):
        K = logits.shape[0]
        z = logits[:, :ns] * temp
        t = torch.zeros_like(z)
        rows = torch.nonzero(tgt >= 0)[:, 0]
        t[rows, tgt[rows].long()] = 1.0
        coef = weight / K * (float(upstream.reshape(-1)[0]) if upstream is not None else 1.0)
        dz.zero_()
        dz[:, :ns] = (coef * temp * (torch.sigmoid(z) - t)).to(torch.bfloat16)

    def adamw_step(self, p, g, m, v, shadow, flags, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
        act = (flags & 1).bool().repeat_interleave(64)
        dec = (flags & 2).bool().repeat_interleave(64)
        gg = g * grad_scale
        bc1 = 1.0 - beta1 ** step
        bc2s = math.sqrt(1.0 - beta2 ** step)
        pn = torch.where(dec, p * (1.0 - lr * wd), p)
        mn = beta1 * m + (1 - beta1) * gg
        vn = beta2 * v + (1 - beta2) * gg * gg
        pn = pn - (lr / bc1) * (mn / (vn.sqrt() / bc2s + eps))
        p.copy_(torch.where(act, pn, p))
        m.copy_(torch.where(act, mn, m))
        v.copy_(torch.where(act, vn, v))
        if shadow is not None:
            shadow.copy_(torch.where(act, p.to(torch.bfloat16), shadow))
