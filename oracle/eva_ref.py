"""TEST INFRASTRUCTURE -- CPU oracle, never imported by the product path.

fp32 CPU restatement (plain torch) of the CLIPSelf hot path of wusize/CLIPSelf:

  CLIPSelf.__call__                    src/training/clipself.py:7-49
  CustomCLIP.encode_image / encode_pseudo_boxes / encode_dense
                                       src/open_clip/eva_clip/model.py:313-340
  EVAVisionTransformer.forward_features / encode_dense / extract_roi_features
                                       src/open_clip/eva_clip/eva_vit_model.py:533-629
  Block.forward / forward_without_attn eva_vit_model.py:300-332
  Attention.forward (math branch) / proj_without_attn
                                       eva_vit_model.py:174-256
  SwiGLU.forward                       eva_vit_model.py:98-105
  PatchEmbed.forward                   eva_vit_model.py:350-356
  rescale_positional_embedding / _denormalize_boxes
                                       eva_vit_model.py:631-664
  VisionRotaryEmbeddingFast            src/open_clip/eva_clip/rope.py:96-214
  LayerNorm                            src/open_clip/eva_clip/transformer.py:52-58
  AdamW grouping / lock                src/training/main.py:161-166,198-213;
                                       eva_vit_model.py:500-516
  cosine_lr                            src/training/scheduler.py:9-10,43-53
  step ordering                        src/training/train.py:80-122

It is functional (a dict of tensors keyed by the reference's state-dict names)
so that it shares no code structure with the reference's nn.Module classes.
Pinned against the reference itself: oracle/gen_golden.py imports the real
reference in the survey container and tests/test_oracle_vs_golden.py checks
this restatement against the captured outputs (tests/golden/*.npz).

``emulate_bf16=True`` rounds tensors to bf16 at the points where the HIP path
stores bf16 (GEMM operands, LN outputs, q/k/v, attention output, hidden), with
a straight-through gradient, to give the "bf16 reference" the north star's
1e-3 tolerance is quoted against.  ``emulate_bf16="kernel"`` and
``encode_image_frozen_schedule`` / ``frozen_block`` move those points to exactly
where the kernels of the training / frozen schedule round (un-normalised
attention probabilities, folded LayerNorms on the split stream, fused
SiLU*mul): where bf16 rounds accounts for ~5e-3 of the distance between two
bf16 evaluations of the tower (tests/test_gpu_parity.py, profiles/r03_parity.md).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from .roi_align_ref import roi_align_1x1


# ----------------------------------------------------------------------------
# rounding hook
# ----------------------------------------------------------------------------
class _Round:
    """emulate_bf16 = False | True | "kernel".  True rounds at the generic points listed in the module docstring; "kernel" additionally
    moves the attention's probability rounding to where the HIP attention kernels round: the UN-normalised exp(s - running max) of the
    online softmax is what goes to the P.V MFMA in bf16, chunk by chunk, and the fp32 row sum divides afterwards (_online_softmax_pv,
    csrc/attention.hip), instead of rounding softmax(s)."""

    def __init__(self, on):
        self.on = bool(on)
        self.kernel_points = on == "kernel"

    def __call__(self, t: torch.Tensor) -> torch.Tensor:
        if not self.on:
            return t
        return t + (t.detach().to(torch.bfloat16).to(t.dtype) - t.detach())


# ----------------------------------------------------------------------------
# RoPE tables (rope.py:118-142, :179-214)
# ----------------------------------------------------------------------------
def rope_tables(grid: int, head_dim: int, pt_seq_len: int = 16, theta: float = 10000.0):
    """cos, sin tables [grid*grid, head_dim] for a grid x grid token map.

    half = head_dim // 2 rotary dims per axis; freqs = theta^(-2i/half), i < half/2;
    positions t = arange(grid)/grid*pt_seq_len; each frequency repeated twice
    (interleaved pairs); row-axis block then column-axis block."""
    half = head_dim // 2
    freqs = 1.0 / (theta ** (torch.arange(0, half, 2)[: half // 2].float() / half))
    t = torch.arange(grid).float() / grid * pt_seq_len
    ang = t[:, None] * freqs[None, :]                 # [grid, half/2]
    ang = ang.repeat_interleave(2, dim=-1)            # [grid, half]
    rows = ang[:, None, :].expand(grid, grid, half)
    cols = ang[None, :, :].expand(grid, grid, half)
    full = torch.cat([rows, cols], dim=-1).reshape(grid * grid, head_dim)
    return full.cos(), full.sin()


def rotate_pairs(x: torch.Tensor) -> torch.Tensor:
    # rope.py:25-29: (x0,x1,x2,x3,...) -> (-x1,x0,-x3,x2,...)
    x = x.reshape(*x.shape[:-1], -1, 2)
    return torch.stack((-x[..., 1], x[..., 0]), dim=-1).flatten(-2)


def apply_rope(t: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """t [B,H,N,d] with token 0 = CLS passed through (eva_vit_model.py:198-204)."""
    body = t[:, :, 1:, :]
    body = body * cos + rotate_pairs(body) * sin
    return torch.cat((t[:, :, :1, :], body), dim=2)


# ----------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------
def layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def pos_embed_for(sd, cfg, grid: int, prefix="visual."):
    pe = sd[prefix + "pos_embed"]
    if grid == cfg.grid:
        return pe
    # eva_vit_model.py:631-643  bicubic, align_corners=False, CLS slot copied
    C = pe.shape[2]
    pe2 = pe[0, 1:].T.contiguous().view(1, C, cfg.grid, cfg.grid)
    pe2 = F.interpolate(pe2, (grid, grid), mode="bicubic", align_corners=False).view(C, grid * grid)
    out = pe.new_zeros(1, 1 + grid * grid, C)
    out[0, 0] = pe[0, 0]
    out[0, 1:] = pe2.T
    return out


def stem(sd, cfg, images, rq, prefix="visual."):
    """patch-embed conv as unfold-GEMM + cls + pos (eva_vit_model.py:537-544)."""
    B, _, Hh, Ww = images.shape
    p = cfg.patch_size
    g = Hh // p
    w = sd[prefix + "patch_embed.proj.weight"].reshape(cfg.width, -1)     # [C, 3*p*p], (c,py,px) order
    patches = images.reshape(B, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * p * p)
    x = rq(patches) @ rq(w).T + sd[prefix + "patch_embed.proj.bias"]
    cls = sd[prefix + "cls_token"].expand(B, -1, -1)
    x = torch.cat((cls, x), dim=1)
    return x + pos_embed_for(sd, cfg, g, prefix), g


def _online_softmax_pv(att, v, rq):
    """softmax(att) @ v as attention.hip computes it: keys in chunks of 96 (<= 224 keys: attn_fwd8_kernel) or 224 (longer sequences),
    un-normalised probabilities relative to the RUNNING maximum rounded to bf16 for the P.V product, fp32 row sums of the unrounded ones,
    output accumulators rescaled when the maximum moves.  att [..., Nq, Nk], v [..., Nk, d] -> [..., Nq, d]."""
    N = att.shape[-1]
    step = 96 if N <= 224 else 224
    m = att.new_full(att.shape[:-1] + (1,), float("-inf"))
    lsum = torch.zeros_like(m)
    acc = att.new_zeros(att.shape[:-1] + (v.shape[-1],))
    for lo in range(0, N, step):
        sc = att[..., lo:lo + step]
        m_new = torch.maximum(m, sc.amax(dim=-1, keepdim=True))
        alpha = torch.exp(m - m_new)
        e = torch.exp(sc - m_new)
        lsum = lsum * alpha + e.sum(dim=-1, keepdim=True)
        acc = acc * alpha + rq(e) @ v[..., lo:lo + step, :]
        m = m_new
    return acc / lsum


def attention(sd, cfg, x, blk, cos, sin, rq):
    """x = norm1 output [B,N,C] (eva_vit_model.py:174-247, math branch)."""
    B, N, C = x.shape
    H, d = cfg.heads, cfg.head_width
    xq = rq(x)
    q = xq @ rq(sd[blk + "attn.q_proj.weight"]).T + sd[blk + "attn.q_bias"]
    k = xq @ rq(sd[blk + "attn.k_proj.weight"]).T
    v = xq @ rq(sd[blk + "attn.v_proj.weight"]).T + sd[blk + "attn.v_bias"]
    q, k, v = (rq(t).reshape(B, N, H, d).permute(0, 2, 1, 3) for t in (q, k, v))
    q = rq(apply_rope(q, cos, sin))
    k = rq(apply_rope(k, cos, sin))
    att = (q * (d ** -0.5)) @ k.transpose(-2, -1)
    if rq.kernel_points:
        o = _online_softmax_pv(att, v, rq).transpose(1, 2).reshape(B, N, C)
    else:
        att = att.softmax(dim=-1)
        o = (rq(att) @ v).transpose(1, 2).reshape(B, N, C)
    o = rq(o)
    o = rq(layer_norm(o, sd[blk + "attn.inner_attn_ln.weight"], sd[blk + "attn.inner_attn_ln.bias"], cfg.ln_eps))
    return o @ rq(sd[blk + "attn.proj.weight"]).T + sd[blk + "attn.proj.bias"]


def proj_without_attn(sd, cfg, x, blk, rq):
    # eva_vit_model.py:249-256
    v = rq(rq(x) @ rq(sd[blk + "attn.v_proj.weight"]).T + sd[blk + "attn.v_bias"])
    v = rq(layer_norm(v, sd[blk + "attn.inner_attn_ln.weight"], sd[blk + "attn.inner_attn_ln.bias"], cfg.ln_eps))
    return v @ rq(sd[blk + "attn.proj.weight"]).T + sd[blk + "attn.proj.bias"]


def swiglu(sd, cfg, x, blk, rq):
    # eva_vit_model.py:98-105
    xq = rq(x)
    x1 = xq @ rq(sd[blk + "mlp.w1.weight"]).T + sd[blk + "mlp.w1.bias"]
    x2 = xq @ rq(sd[blk + "mlp.w2.weight"]).T + sd[blk + "mlp.w2.bias"]
    x1, x2 = rq(x1), rq(x2)
    h = rq(F.silu(x1) * x2)
    h = rq(layer_norm(h, sd[blk + "mlp.ffn_ln.weight"], sd[blk + "mlp.ffn_ln.bias"], cfg.ln_eps))
    return h @ rq(sd[blk + "mlp.w3.weight"]).T + sd[blk + "mlp.w3.bias"]


def block(sd, cfg, x, i, cos, sin, rq, with_attn=True, prefix="visual."):
    blk = f"{prefix}blocks.{i}."
    n1 = rq(layer_norm(x, sd[blk + "norm1.weight"], sd[blk + "norm1.bias"], cfg.ln_eps))
    if with_attn:
        x = x + attention(sd, cfg, n1, blk, cos, sin, rq)
    else:
        x = x + proj_without_attn(sd, cfg, n1, blk, rq)
    n2 = rq(layer_norm(x, sd[blk + "norm2.weight"], sd[blk + "norm2.bias"], cfg.ln_eps))
    return x + swiglu(sd, cfg, n2, blk, rq)


# ----------------------------------------------------------------------------
# tower entry points
# ----------------------------------------------------------------------------
def encode_image(sd, cfg, images, emulate_bf16=False, prefix="visual."):
    """Teacher path: full ViT -> final LN -> CLS -> head (eva_vit_model.py:581-586)."""
    rq = _Round(emulate_bf16)
    x, g = stem(sd, cfg, images, rq, prefix)
    cos, sin = rope_tables(g, cfg.head_width, cfg.pt_hw_seq_len)
    for i in range(cfg.layers):
        x = block(sd, cfg, x, i, cos, sin, rq, True, prefix)
    x = rq(layer_norm(x, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], cfg.ln_eps))[:, 0]
    return x @ rq(sd[prefix + "head.weight"]).T + sd[prefix + "head.bias"]


# ----------------------------------------------------------------------------
# bf16 emulation at the FROZEN tower's own rounding points
# ----------------------------------------------------------------------------
def _plane_round(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> the bf16 value held by the `hi` plane of the split residual stream: upper half of bits(x) + 0x8000, i.e. round to
    nearest with halves away from zero (include/clipself_hip.h, cs_gemm_nt_ln_split)."""
    y = (x.float().contiguous().view(torch.int32) + 0x8000) & ~0xFFFF
    return y.view(torch.float32)


def _folded_linear(a_bf16, stats_of, ln_w, ln_b, W, b, eps, rq):
    """LayerNorm(z) . W^T + b the way the frozen tower evaluates it (engine._build_folds / cs_gemm_nt_ln): the GEMM contracts the
    UN-normalised bf16 rows `a_bf16` with bf16(gamma (.) W); the row statistics (of `stats_of`, fp32) and beta enter in the fp32 epilogue:
    rstd * (acc - mean * colsum) + (W . beta + b).  Same function as eva_vit_model.py:102-103,218-219,306-307 up to where bf16 rounds."""
    Wf = rq(W * ln_w[None, :])
    colsum = Wf.sum(dim=1)
    d = W @ ln_b + b
    mean = stats_of.mean(-1, keepdim=True)
    var = (stats_of * stats_of).mean(-1, keepdim=True) - mean * mean
    rstd = torch.rsqrt(var + eps)
    return rstd * (a_bf16 @ Wf.T - mean * colsum) + d


def frozen_block(sd, cfg, x, i, cos, sin, folded=True, fold_norm1=True, prefix="visual.", fold_kv=False, fold_block=True):
    """One block of the frozen (teacher) tower with bf16 rounding exactly where the HIP schedule rounds (see
    encode_image_frozen_schedule).  x fp32 [B, N, C] -> fp32 [B, N, C].  folded=False: the plain schedule of the CLS-only last block
    (fold_kv: its keys and values nevertheless come from the folded norm1 GEMM on the split stream, its query from a plain LayerNorm);
    fold_norm1=False: block 0, whose norm1 is a LayerNorm kernel; fold_block=False: norm1 and norm2 are LayerNorm kernels in every block
    (the schedule the engine's row-statistics guard falls back to, engine.block_folds_active) and only the sub-LayerNorms are folded."""
    rq = _Round("kernel")
    eps = cfg.ln_eps
    B, N, C = x.shape
    H, d = cfg.heads, cfg.head_width
    blk = f"{prefix}blocks.{i}."
    wqkv = torch.cat([sd[blk + "attn.q_proj.weight"], sd[blk + "attn.k_proj.weight"], sd[blk + "attn.v_proj.weight"]])
    bqkv = torch.cat([sd[blk + "attn.q_bias"], torch.zeros_like(sd[blk + "attn.q_bias"]), sd[blk + "attn.v_bias"]])
    w12 = torch.cat([sd[blk + "mlp.w1.weight"], sd[blk + "mlp.w2.weight"]])
    b12 = torch.cat([sd[blk + "mlp.w1.bias"], sd[blk + "mlp.w2.bias"]])
    Hd = sd[blk + "mlp.w1.weight"].shape[0]
    if not (folded and fold_norm1 and fold_block):
        n1 = rq(layer_norm(x, sd[blk + "norm1.weight"], sd[blk + "norm1.bias"], eps))
        qkv = n1 @ rq(wqkv).T + bqkv
        if fold_kv:
            kv = _folded_linear(_plane_round(x), x, sd[blk + "norm1.weight"], sd[blk + "norm1.bias"], wqkv, bqkv, eps, rq)
            qkv = torch.cat([qkv[..., :C], kv[..., C:]], dim=-1)
    else:
        qkv = _folded_linear(_plane_round(x), x, sd[blk + "norm1.weight"], sd[blk + "norm1.bias"], wqkv, bqkv, eps, rq)
    q, k, v = (rq(t).reshape(B, N, H, d).permute(0, 2, 1, 3) for t in (qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]))
    q, k = rq(apply_rope(q, cos, sin)), rq(apply_rope(k, cos, sin))
    att = (q * (d ** -0.5)) @ k.transpose(-2, -1)
    o = rq(_online_softmax_pv(att, v, rq).transpose(1, 2).reshape(B, N, C))     # un-normalised probabilities go to the P.V MFMA in bf16
    if folded:
        x = x + _folded_linear(o, o, sd[blk + "attn.inner_attn_ln.weight"], sd[blk + "attn.inner_attn_ln.bias"],
                               sd[blk + "attn.proj.weight"], sd[blk + "attn.proj.bias"], eps, rq)
        if fold_block:
            x12 = _folded_linear(_plane_round(x), x, sd[blk + "norm2.weight"], sd[blk + "norm2.bias"], w12, b12, eps, rq)
        else:
            x12 = rq(layer_norm(x, sd[blk + "norm2.weight"], sd[blk + "norm2.bias"], eps)) @ rq(w12).T + b12
    else:
        o = rq(layer_norm(o, sd[blk + "attn.inner_attn_ln.weight"], sd[blk + "attn.inner_attn_ln.bias"], eps))
        x = x + (o @ rq(sd[blk + "attn.proj.weight"]).T + sd[blk + "attn.proj.bias"])
        n2 = rq(layer_norm(x, sd[blk + "norm2.weight"], sd[blk + "norm2.bias"], eps))
        x12 = n2 @ rq(w12).T + b12
    h = rq(F.silu(x12[..., :Hd]) * x12[..., Hd:])                 # x1 / x2 stay in the fp32 accumulators: never rounded on their own
    if folded:
        return x + _folded_linear(h, h, sd[blk + "mlp.ffn_ln.weight"], sd[blk + "mlp.ffn_ln.bias"], sd[blk + "mlp.w3.weight"],
                                  sd[blk + "mlp.w3.bias"], eps, rq)
    h = rq(layer_norm(h, sd[blk + "mlp.ffn_ln.weight"], sd[blk + "mlp.ffn_ln.bias"], eps))
    return x + (h @ rq(sd[blk + "mlp.w3.weight"]).T + sd[blk + "mlp.w3.bias"])


def encode_image_frozen_schedule(sd, cfg, images, prefix="visual.", return_stream=False, fold_block=True):
    """encode_image() with bf16 rounding exactly where the frozen (teacher) schedule of the HIP path rounds -- an independent restatement of
    that schedule's ARITHMETIC, not of its code (clipself_amd/engine.py: _teacher_block_folded, _block_fwd_cls):
      * operands of every GEMM bf16, accumulation and epilogues fp32, q|k|v / attention output / SwiGLU hidden stored bf16;
      * the four LayerNorms of a block are folded into the following GEMM (_folded_linear): norm1 / norm2 see the residual stream rounded
        half-away-from-zero (the split stream's hi plane) with the statistics of its fp32 values; inner_attn_ln / ffn_ln see the stored
        bf16 activations with the statistics of those rounded values; block 0 keeps a plain norm1; the last (CLS-only) block is unfolded except for
        its keys and values, which come from the folded norm1 GEMM on the split stream like everywhere else;
      * SiLU(x1) * x2 is formed from the fp32 accumulators (x1 / x2 are never stored), unlike emulate_bf16=True above, which rounds them
        as the training schedule does; the attention's P.V product takes the un-normalised exp(s - max) in bf16 (_Round("kernel")).
    Against this oracle the kernels' own error is what is left (summation order + the rounding flips it triggers); against
    encode_image(emulate_bf16=True) the different rounding points alone move the features by ~5e-3 (profiles/r03_parity.md).
    return_stream: also the fp32 residual stream in front of every block ([L + 1] tensors, the last one is the tower's output stream).
    fold_block=False: the schedule with norm1 / norm2 as LayerNorm kernels (sub-LayerNorms still folded), see frozen_block."""
    rq = _Round("kernel")
    x, g = stem(sd, cfg, images, rq, prefix)
    cos, sin = rope_tables(g, cfg.head_width, cfg.pt_hw_seq_len)
    L = cfg.layers
    stream = [x]
    for i in range(L):
        x = frozen_block(sd, cfg, x, i, cos, sin, folded=i < L - 1, fold_norm1=i > 0, prefix=prefix,
                         fold_kv=(i == L - 1 and L > 1 and fold_block), fold_block=fold_block)
        stream.append(x)
    out = rq(layer_norm(x, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], cfg.ln_eps))[:, 0]
    out = out @ rq(sd[prefix + "head.weight"]).T + sd[prefix + "head.bias"]
    return (out, stream) if return_stream else out


def encode_dense(sd, cfg, images, emulate_bf16=False, prefix="visual."):
    """Student dense path -> L2-normalised token map [B, g*g, E] (eva_vit_model.py:588-623)."""
    rq = _Round(emulate_bf16)
    x, g = stem(sd, cfg, images, rq, prefix)
    cos, sin = rope_tables(g, cfg.head_width, cfg.pt_hw_seq_len)
    for i in range(cfg.layers - 1):
        x = block(sd, cfg, x, i, cos, sin, rq, True, prefix)
    x = block(sd, cfg, x, cfg.layers - 1, cos, sin, rq, False, prefix)[:, 1:]
    x = rq(layer_norm(x, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], cfg.ln_eps))
    x = x @ rq(sd[prefix + "head.weight"]).T + sd[prefix + "head.bias"]
    return F.normalize(x, dim=-1), g


def rois_from_list(normed_boxes_list, g: int) -> torch.Tensor:
    """list[Tensor[k_i,4]] in [0,1] -> [K,5] (batch, x0,y0,x1,y1) in token-grid units,
    float32 arithmetic like _denormalize_boxes (eva_vit_model.py:655-664)."""
    rows = []
    for i, b in enumerate(normed_boxes_list):
        bb = b.detach().float().clone()
        bb[:, [0, 2]] *= g
        bb[:, [1, 3]] *= g
        rows.append(torch.cat([torch.full((len(bb), 1), float(i)), bb], dim=1))
    return torch.cat(rows) if rows else torch.zeros(0, 5)


def encode_pseudo_boxes(sd, cfg, images, normed_boxes_list, emulate_bf16=False, prefix="visual."):
    dense, g = encode_dense(sd, cfg, images, emulate_bf16, prefix)
    B = images.shape[0]
    feat = dense.reshape(B, g, g, -1)                      # NHWC view of the token map
    return roi_align_1x1(feat, rois_from_list(normed_boxes_list, g))


def split_valid(normed_boxes, image_crops):
    """clipself.py:29-36"""
    rois_list, crops_list = [], []
    for bb, cc in zip(normed_boxes, image_crops):
        valid = bb[:, -1] > 0.5
        rois_list.append(bb[valid, :4])
        crops_list.append(cc[valid])
    return rois_list, torch.cat(crops_list)


def _tower(cfg):
    """The module restating cfg's tower family: this one (EVA02) or oracle/clip_vit_ref.py (OpenAI-CLIP ViT)."""
    if getattr(cfg, "arch", "eva02") == "openai":
        from . import clip_vit_ref
        return clip_vit_ref
    import sys
    return sys.modules[__name__]


def clipself_loss(student_sd, teacher_sd, cfg, batch, cosine_weight=1.0, emulate_bf16=False):
    """CLIPSelf.__call__ (clipself.py:7-49) -> (loss, student_roi, teacher_feats)."""
    images, normed_boxes, image_crops = batch
    rois_list, crops = split_valid(normed_boxes, image_crops)
    tower = _tower(cfg)
    with torch.no_grad():
        teacher = tower.encode_image(teacher_sd, cfg, crops, emulate_bf16)
    student = tower.encode_pseudo_boxes(student_sd, cfg, images, rois_list, emulate_bf16)
    ns = F.normalize(student, dim=-1)
    nt = F.normalize(teacher, dim=-1)
    loss = (1.0 - (ns * nt).sum(-1).mean()) * cosine_weight
    return loss, student, teacher


# ----------------------------------------------------------------------------
# training-step restatement
# ----------------------------------------------------------------------------
def trainable_names(sd, cfg, unlocked_groups: int, prefix="visual."):
    """visual.lock (eva_vit_model.py:500-516): everything frozen except the last
    ``unlocked_groups`` blocks; text tower frozen (model.py:284-288); logit_scale trainable."""
    keep = []
    first = cfg.layers - unlocked_groups if unlocked_groups > 0 else 0   # blocks[-0:] == all blocks
    for name in sd:
        if name.startswith(prefix + "blocks."):
            if int(name[len(prefix + "blocks."):].split(".")[0]) >= first:
                keep.append(name)
        elif name == "logit_scale":
            keep.append(name)
    return keep


def is_no_decay(name: str, ndim: int) -> bool:
    # main.py:199  p.ndim < 2 or "bn" in n or "ln" in n or "bias" in n or 'logit_scale' in n
    return ndim < 2 or "bn" in name or "ln" in name or "bias" in name or "logit_scale" in name


def make_optimizer(params: dict, lr, wd, betas=(0.9, 0.999), eps=1e-8):
    """main.py:198-213 with torch.optim.AdamW itself as the arithmetic oracle."""
    no_decay = [p for n, p in params.items() if is_no_decay(n, p.ndim)]
    decay = [p for n, p in params.items() if not is_no_decay(n, p.ndim)]
    return torch.optim.AdamW(
        [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": wd}],
        lr=lr, betas=betas, eps=eps)


def cosine_lr_value(step: int, base_lr: float, warmup: int, total: int) -> float:
    # scheduler.py:9-10,43-53
    if step < warmup:
        return base_lr * (step + 1) / warmup
    e, es = step - warmup, total - warmup
    return 0.5 * (1 + np.cos(np.pi * e / es)) * base_lr


def train_steps(student_sd, teacher_sd, cfg, batches, lr=1e-5, wd=0.1, warmup=1000, total_steps=10000,
                unlocked_groups=None, cosine_weight=1.0, emulate_bf16=False):
    """train_one_epoch body (train.py:80-122) for len(batches) steps, in place on student_sd.
    Returns per-step dict(loss, lr) and the grads of the last step."""
    unlocked_groups = cfg.layers if unlocked_groups is None else unlocked_groups
    if "logit_scale" not in student_sd:
        student_sd["logit_scale"] = torch.ones([]) * math.log(1 / 0.07)
    names = _tower(cfg).trainable_names(student_sd, cfg, unlocked_groups)
    for n in student_sd:
        student_sd[n].requires_grad_(n in names)
    opt = make_optimizer({n: student_sd[n] for n in names}, lr, wd)
    log, grads = [], {}
    for step, batch in enumerate(batches):
        cur = cosine_lr_value(step, lr, warmup, total_steps)
        for gparam in opt.param_groups:
            gparam["lr"] = cur
        opt.zero_grad()
        loss, _, _ = clipself_loss(student_sd, teacher_sd, cfg, batch, cosine_weight, emulate_bf16)
        loss.backward()
        grads = {n: (student_sd[n].grad.detach().clone() if student_sd[n].grad is not None else None) for n in names}
        opt.step()
        with torch.no_grad():
            student_sd["logit_scale"].clamp_(0, math.log(100))
        log.append({"loss": float(loss.detach()), "lr": float(cur)})
    return log, grads


# ------------------------------------------------------------------------------------------------ RegionCLIP (BASELINE configs[4])
def regionclip_loss(sd, cfg, images, boxes, noun_embeddings, appeared=None, contrast_weight=1.0, emulate_bf16=False):
    """Restatement of RegionCLIP.__call__ (/root/reference/src/training/region_clip.py:28-67): L2-normalised RoI features of the valid
    boxes against the L2-normalised noun bank, * exp(logit_scale), binary cross-entropy over the federated column subset `appeared`
    (get_fed_loss_inds, :7-16; drawn here with the reference's own calls when not given), summed over columns, mean over boxes.
    boxes [B, max_boxes, 6] = (x0, y0, x1, y1 in [0,1], label, valid).  Pinned on tests/golden/tiny_regionclip.npz."""
    import torch.nn.functional as F
    rois, labels = [], []
    for per_image in boxes:
        keep = per_image[per_image[:, -1] > 0.5]
        labels.append(keep[:, 4].long())
        rois.append(keep[:, :4])
    labels = torch.cat(labels)
    feats = F.normalize(encode_pseudo_boxes(sd, cfg, images, rois, emulate_bf16=emulate_bf16), dim=-1)
    nouns = F.normalize(noun_embeddings.float(), dim=-1)
    temp = (sd["logit_scale"] if "logit_scale" in sd else torch.ones([]) * math.log(1 / 0.07)).exp().detach()
    logits = feats @ nouns.T * temp
    target = torch.zeros_like(logits)
    target[range(len(labels)), labels] = 1.0
    if appeared is None:
        appeared = torch.unique(labels)
        if len(appeared) < 100:
            prob = appeared.new_ones(nouns.shape[0]).float()
            prob[appeared] = 0
            appeared = torch.cat([appeared, torch.multinomial(prob, 100 - len(appeared), replacement=False)])
    loss = F.binary_cross_entropy_with_logits(logits[:, appeared], target[:, appeared], reduction="none").sum(-1).mean()
    return loss * contrast_weight
