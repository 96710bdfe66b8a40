"""Step engine of the OpenAI-CLIP vision transformer on the CLIPSelf hot path: the same flat-buffer
machinery and kernels as the EVA02 engine (clipself_amd/engine.py), with this tower family's schedule.  Where the reference
does each stage (paths under /root/reference/src/open_clip/):

  stem            transformer.py:551-569     conv1 (no bias; im2row + GEMM) + class_embedding + positional_embedding, ln_pre
  block           transformer.py:232-244     x += out_proj(MHA(ln_1 x)); x += c_proj(act(c_fc(ln_2 x)))
  attention       transformer.py:203,217-230 nn.MultiheadAttention: fused in_proj [3C,C] (+bias on q, k and v), softmax(qk^T/8)v
  last dense blk  transformer.py:247-260     out_proj(v-slice of in_proj(ln_1 x)), no attention (maskclip style)
  MLP             transformer.py:209-213,31-34   GELU (erf) or QuickGELU
  image head      transformer.py:486-494     ln_post(x[:,0]) @ proj
  dense head      transformer.py:576-587     normalize(ln_post(x[:,1:]) @ proj)
  mask attention  transformer.py:660-671,736-834   extract_type='v1' / encode_masks(mask_attn=True): extra query tokens (mask_attn_pool, inference)
  lock            transformer.py:391-422     groups = [stem, positional_embedding, blocks..., last block]; the last n train
                                             (n > L: positional_embedding, then conv1 / class_embedding / ln_pre: _stem_bwd)

The frozen teacher uses the EVA02 engine's schedule tricks unchanged: CLS-query-only last block, ln_1 / ln_2 folded into the in_proj / c_fc
GEMMs with the residual GEMMs emitting the bf16 copy and the row statistics of the stream (`_teacher_block_folded`).

Differences to the EVA02 schedule: no RoPE (the attention kernels get identity tables), no sub-LayerNorms, `proj` is a bias-free
[C,E] matrix kept transposed for the forward GEMM, and the last dense block owns no never-reached *tensor* (q/k are rows of the one
in_proj_weight parameter, whose gradient rows stay zero -- exactly what autograd hands torch's AdamW).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .config import TowerCfg
from .engine import (BF16, DX_F32_ACCUM, DX_F32_ASSIGN, EPI_BF16, EPI_F32, EPI_GELU_BF16, EPI_PATCH_F32, EPI_QGELU_BF16,
                     EPI_RESID_F32, F32, EvaEngine, _round_up)


def clip_vit_layout(cfg: TowerCfg, prefix: str = "visual."):
    """Allocation groups in layer order with the reference's state-dict names (transformer.py:355-389,190-215)."""
    C, Hd, E, p = cfg.width, cfg.hidden, cfg.embed_dim, cfg.patch_size
    Kp = _round_up(3 * p * p, 64)
    t = lambda name, shape, storage=None: (name, shape, storage if storage is not None else shape)
    groups = [[t(prefix + "class_embedding", (C,))], [t(prefix + "positional_embedding", (cfg.tokens, C))],
              [t(prefix + "conv1.weight", (C, 3, p, p), (C, Kp))], [t(prefix + "ln_pre.weight", (C,))], [t(prefix + "ln_pre.bias", (C,))]]
    for i in range(cfg.layers):
        b = f"{prefix}transformer.resblocks.{i}."
        groups += [[t(b + "ln_1.weight", (C,))], [t(b + "ln_1.bias", (C,))],
                   [t(b + "attn.in_proj_weight", (3 * C, C))], [t(b + "attn.in_proj_bias", (3 * C,))],
                   [t(b + "attn.out_proj.weight", (C, C))], [t(b + "attn.out_proj.bias", (C,))],
                   [t(b + "ln_2.weight", (C,))], [t(b + "ln_2.bias", (C,))],
                   [t(b + "mlp.c_fc.weight", (Hd, C))], [t(b + "mlp.c_fc.bias", (Hd,))],
                   [t(b + "mlp.c_proj.weight", (C, Hd))], [t(b + "mlp.c_proj.bias", (C,))]]
    groups += [[t(prefix + "ln_post.weight", (C,))], [t(prefix + "ln_post.bias", (C,))], [t(prefix + "proj", (C, E))]]
    return groups


class ClipVitEngine(EvaEngine):
    BLOCK_TAG = "transformer.resblocks."

    def __init__(self, cfg: TowerCfg, ops, trainable: bool = False, prefix: str = "visual."):
        if cfg.hidden % 64 or cfg.width % 64 or cfg.embed_dim % 64:
            raise NotImplementedError(f"{cfg.name}: width, MLP width and embed_dim must be multiples of 64")
        super().__init__(cfg, ops, trainable=trainable, prefix=prefix)
        self.fold_sub_ln = False                                # there are no sub-LayerNorms in this family
        # Frozen towers, encode_image(): ln_1 / ln_2 are applied inside the in_proj / c_fc GEMM epilogues (gamma folded into a bf16 copy
        # of the weight, the residual GEMMs emit the bf16 copy of the stream and its row statistics), and the last block runs for the
        # CLS query only (forward() consumes x[:, 0] alone, transformer.py:486-494).  Same switches as the EVA02 engine.
        self.fold_block_ln = not trainable
        self.cls_only_last_block = True
        self.fold_cls_block = True                              # ln_1 of the CLS-only block folded into its K|V GEMM (A/B switch)
        # lock() with more groups than blocks (transformer.py:391-422): 1 = positional_embedding trains, 2 = conv1 / class_embedding / ln_pre too
        self.stem_level = 0

    def _layout(self):
        return clip_vit_layout(self.cfg, self.prefix)

    def _never_reached(self, i, name):
        return False

    # ------------------------------------------------------------------------------------------ parameters
    def sync_transposed(self, blocks=None):
        """proj^T [E,C] for the forward head GEMM (every tower), and W^T shadows of the trainable blocks for the dgrad GEMMs."""
        cfg, C, Hd = self.cfg, self.cfg.width, self.cfg.hidden
        pairs = [(self.w[self.prefix + "proj"], self._wt_alloc("head_fwd", C, cfg.embed_dim))]
        if self.trainable:
            for i in (range(self.first_trainable, cfg.layers) if blocks is None else blocks):
                b = f"{self.prefix}{self.BLOCK_TAG}{i}."
                pairs.append((self.w[b + "attn.in_proj_weight"], self._wt_alloc((i, "qkv"), 3 * C, C)))
                pairs.append((self.w[b + "attn.out_proj.weight"], self._wt_alloc((i, "proj"), C, C)))
                pairs.append((self.w[b + "mlp.c_fc.weight"], self._wt_alloc((i, "fc"), Hd, C)))
                pairs.append((self.w[b + "mlp.c_proj.weight"], self._wt_alloc((i, "cproj"), C, Hd)))
        self.ops.transpose_bf16_batched(pairs)

    def sync_shadow(self):
        self.ops.cast_f32_bf16(self.master, self.shadow)
        self._pos_cache.clear()
        self.block_fold_ratio = None
        self.sync_transposed()
        if self.fold_block_ln:
            self._build_folds()

    def _build_folds(self):
        """gamma (.) W in bf16, its row sums and W.beta + b for ln_1 -> in_proj and ln_2 -> c_fc of every block (one-time, after a load)."""
        self.fold = {}
        with torch.no_grad():
            for i in range(self.cfg.layers):
                b = f"{self.prefix}{self.BLOCK_TAG}{i}."
                out = {}
                for key, wname, bname, ln in (("qkv", "attn.in_proj_weight", "attn.in_proj_bias", "ln_1"), ("fc", "mlp.c_fc.weight", "mlp.c_fc.bias", "ln_2")):
                    W, bias = self.p[b + wname], self.p[b + bname]
                    Wf = (W * self.p[b + ln + ".weight"][None, :]).to(BF16).contiguous()
                    out[key] = (Wf, Wf.float().sum(dim=1).contiguous(), (W @ self.p[b + ln + ".bias"] + bias).contiguous())
                self.fold[i] = out

    def set_trainable_blocks(self, unlocked_groups: int):
        """VisionTransformer.lock (transformer.py:391-422): of the groups [[conv1, class_embedding, ln_pre], positional_embedding, block 0 ..
        L-1] the last n train; n = 0 freezes everything, n = L + 1 adds the positional embedding, n >= L + 2 the stem (a slice longer than
        the list is the whole list).  ln_post and proj never train in this family (:405-408)."""
        L = self.cfg.layers
        self.stem_level = min(max(unlocked_groups - L, 0), 2)
        super().set_trainable_blocks(min(unlocked_groups, L))               # (clears train_all)
        if unlocked_groups <= 0:
            self.first_trainable = L
            if self.trainable:
                self.flags.zero_()

    def set_trainable_all(self):
        """No lock at all (training.main without --lock-image, src/training/main.py:161-166): besides the stem and the blocks, ln_post and proj
        train (transformer.py:576-584 on the dense path)."""
        self.stem_level = 2
        super().set_trainable_all()

    def _nonblock_trains(self, name):
        if self.train_all:
            return True
        tail = name[len(self.prefix):]
        if tail == "positional_embedding":
            return self.stem_level >= 1
        return self.stem_level >= 2 and tail in ("conv1.weight", "class_embedding", "ln_pre.weight", "ln_pre.bias")

    def _pos_trains(self):
        return self.stem_level >= 1

    # ------------------------------------------------------------------------------------------ tables
    def rope_tables(self, grid: int):
        """No rotary embedding in this family: cos = 1, sin = 0 turn the attention kernels' rotation into the identity."""
        key = ("rope", grid)
        if key not in self._tables:
            shape = (grid * grid, self.cfg.head_width)
            self._tables[key] = (torch.ones(shape, dtype=F32, device=self.device), torch.zeros(shape, dtype=F32, device=self.device))
        return self._tables[key]

    def pos_for(self, grid: int):
        """positional_embedding [N, C] fp32, bicubic-rescaled for a non-native grid (transformer.py:724-734); cached per grid."""
        if grid not in self._pos_cache:
            pe = self.p[self.prefix + "positional_embedding"]
            if grid != self.cfg.grid:
                C = pe.shape[1]
                pe2 = pe[1:].T.contiguous().view(1, C, self.cfg.grid, self.cfg.grid)
                pe2 = F.interpolate(pe2, (grid, grid), mode="bicubic", align_corners=False).view(C, grid * grid)
                pe = torch.cat([pe[:1], pe2.T], dim=0)
            self._pos_cache[grid] = pe.contiguous()
        return self._pos_cache[grid]

    # ------------------------------------------------------------------------------------------ forward pieces
    def _stem(self, images, keep=None):
        """keep (dict, stem_level > 0): the im2row matrix, ln_pre's input and row statistics for _stem_bwd."""
        ops, cfg, P = self.ops, self.cfg, self.prefix
        B, _, S, _ = images.shape
        p, C = cfg.patch_size, cfg.width
        g = S // p
        N = g * g + 1
        A = ops.empty((B * g * g, self.Kpe), BF16)
        ops.im2row(images.contiguous(), A, p)
        x = ops.empty((B, N, C), F32)
        pos = self.pos_for(g)
        ops.gemm_nt(A, self.storage_of(self.shadow, P + "conv1.weight"), x.view(B * N, C), extra=pos, epi=EPI_PATCH_F32, group=g * g)
        ops.cls_row(x, self.p[P + "class_embedding"], pos)
        y = ops.empty((B, N, C), F32)                       # ln_pre's output is the residual stream
        mean = rstd = None
        if keep is not None:
            mean, rstd = ops.empty((B * N,), F32), ops.empty((B * N,), F32)
            keep.update(patches=A, x_pre=x.view(B * N, C), st=(mean, rstd))
        ops.layernorm_fwd_f32(x.view(B * N, C), self.p[P + "ln_pre.weight"], self.p[P + "ln_pre.bias"], y.view(B * N, C), mean, rstd, cfg.ln_eps)
        return y, g

    def _block_fwd(self, i, x, B, N, cos, sin, with_attn=True, save=None, inplace=True):
        ops, cfg = self.ops, self.cfg
        C, Hd, H, eps = cfg.width, cfg.hidden, cfg.heads, cfg.ln_eps
        b = f"{self.prefix}{self.BLOCK_TAG}{i}."
        M = B * N
        keep = save is not None
        st = (lambda: (ops.empty((M,), F32), ops.empty((M,), F32))) if keep else (lambda: (None, None))

        ln1 = ops.empty((M, C), BF16)
        m1, r1 = st()
        ops.layernorm_fwd(x, self.p[b + "ln_1.weight"], self.p[b + "ln_1.bias"], ln1, m1, r1, eps)
        wqkv, bqkv = self.w[b + "attn.in_proj_weight"], self.p[b + "attn.in_proj_bias"]
        qkv = lse = None
        att = ops.empty((M, C), BF16)
        if with_attn:
            qkv = ops.empty((M, 3 * C), BF16)
            ops.gemm_nt(ln1, wqkv, qkv, bias=bqkv, epi=EPI_BF16)
            lse = ops.empty((B * H, N), F32) if keep else None
            ops.attn_fwd(qkv, cos, sin, att, lse, B, N, H, cfg.head_width ** -0.5)
        else:
            ops.gemm_nt(ln1, wqkv[2 * C:], att, bias=bqkv[2 * C:], epi=EPI_BF16)        # proj_without_attn: the value rows only
        x1 = x if inplace else ops.empty((M, C), F32)
        ops.gemm_nt(att, self.w[b + "attn.out_proj.weight"], x1, bias=self.p[b + "attn.out_proj.bias"], extra=x, epi=EPI_RESID_F32)

        ln2 = ops.empty((M, C), BF16)
        m2, r2 = st()
        ops.layernorm_fwd(x1, self.p[b + "ln_2.weight"], self.p[b + "ln_2.bias"], ln2, m2, r2, eps)
        hid = ops.empty((M, Hd), BF16)
        fc = None
        if keep:
            fc = ops.empty((M, Hd), BF16)
            ops.gemm_nt(ln2, self.w[b + "mlp.c_fc.weight"], fc, bias=self.p[b + "mlp.c_fc.bias"], epi=EPI_BF16)
            ops.gelu_fwd(fc, hid, cfg.quick_gelu)
        else:
            ops.gemm_nt(ln2, self.w[b + "mlp.c_fc.weight"], hid, bias=self.p[b + "mlp.c_fc.bias"],
                        epi=EPI_QGELU_BF16 if cfg.quick_gelu else EPI_GELU_BF16)
        x2 = x1 if inplace else ops.empty((M, C), F32)
        ops.gemm_nt(hid, self.w[b + "mlp.c_proj.weight"], x2, bias=self.p[b + "mlp.c_proj.bias"], extra=x1, epi=EPI_RESID_F32)
        if keep:
            save.update(x0=x, ln1=ln1, st1=(m1, r1), qkv=qkv, lse=lse, att=att, with_attn=with_attn, x1=x1, ln2=ln2, st2=(m2, r2),
                        fc=fc, hid=hid)
        return x2

    def _teacher_block_folded(self, i, x, xb, st, B, N, cos, sin, emit_next, lo=None):
        """One frozen-tower block with both LayerNorms folded into the GEMMs (in place on x).  xb / st = bf16 copy and (mean, rstd) of x as
        left by the previous block's c_proj GEMM, or None (first block: plain ln_1 kernel).  Returns (xb, st) for the next block when
        emit_next.  With lo (int16 [M, C]; round 4) the stream lives in the two 16-bit planes (xb, lo) between the first residual GEMM of
        the tower, which reads fp32 x, and the last one (emit_next False), which writes fp32 x again -- cs_gemm_nt_ln_split as in the EVA02
        engine: 8 instead of 10 bytes per stream element and residual GEMM, same fp32 values.  This family has no LayerNorm in front of
        out_proj / c_proj, so the split kernel gets the identity statistics (mean 0, rstd 1, column sums 0: x + 1 * (acc - 0 * 0) + b)."""
        ops, cfg = self.ops, self.cfg
        C, Hd, H, eps = cfg.width, cfg.hidden, cfg.heads, cfg.ln_eps
        b = f"{self.prefix}{self.BLOCK_TAG}{i}."
        M = B * N
        f = self.fold[i]
        qkv = ops.empty((M, 3 * C), BF16)
        first = xb is None
        if first:
            ln1 = ops.empty((M, C), BF16)
            ops.layernorm_fwd(x, self.p[b + "ln_1.weight"], self.p[b + "ln_1.bias"], ln1, None, None, eps)
            ops.gemm_nt(ln1, self.w[b + "attn.in_proj_weight"], qkv, bias=self.p[b + "attn.in_proj_bias"], epi=EPI_BF16)
        else:
            Wq, cq, dq = f["qkv"]
            ops.gemm_nt_ln(xb, Wq, qkv, bias=dq, ln_mean=st[0], ln_rstd=st[1], ln_colsum=cq, epi=EPI_BF16)
        att = ops.empty((M, C), BF16)
        ops.attn_fwd(qkv, cos, sin, att, None, B, N, H, cfg.head_width ** -0.5)
        part = ops.empty(((C + 63) // 64, M, 2), F32)
        xb2 = xb if (lo is not None and not first) else ops.empty((M, C), BF16)
        ident = self._identity_stats(M, C) if lo is not None else None
        wo, bo = self.w[b + "attn.out_proj.weight"], self.p[b + "attn.out_proj.bias"]
        if lo is not None:
            ops.gemm_nt_ln_split(att, wo, xb2, lo, bo, *ident, x_in=x if first else None, stats_part=part)
        else:
            ops.gemm_nt_ln(att, wo, x, bias=bo, extra=x, stats_part=part, xb_out=xb2, epi=EPI_RESID_F32)
        mean, rstd = ops.empty((M,), F32), ops.empty((M,), F32)
        ops.ln_stats_finalize(part, 64, C, mean, rstd, eps)
        Wf, cf, df = f["fc"]
        hid = ops.empty((M, Hd), BF16)
        ops.gemm_nt_ln(xb2, Wf, hid, bias=df, ln_mean=mean, ln_rstd=rstd, ln_colsum=cf, epi=EPI_QGELU_BF16 if cfg.quick_gelu else EPI_GELU_BF16)
        wp, bp = self.w[b + "mlp.c_proj.weight"], self.p[b + "mlp.c_proj.bias"]
        if lo is not None:
            if not emit_next:
                ops.gemm_nt_ln_split(hid, wp, xb2, lo, bp, *ident, x_out=x)
                return None, None
            ops.gemm_nt_ln_split(hid, wp, xb2, lo, bp, *ident, stats_part=part)
        else:
            if not emit_next:
                ops.gemm_nt(hid, wp, x, bias=bp, extra=x, epi=EPI_RESID_F32)
                return None, None
            ops.gemm_nt_ln(hid, wp, x, bias=bp, extra=x, stats_part=part, xb_out=xb2, epi=EPI_RESID_F32)
        ops.ln_stats_finalize(part, 64, C, mean, rstd, eps)
        return xb2, (mean, rstd)

    def _identity_stats(self, M, C):
        """(mean = 0 [M], rstd = 1 [M], column sums = 0 [C]): statistics under which a folded-LayerNorm epilogue is the plain one, bit for bit."""
        key = ("ident", M, C)
        if key not in self._tables:
            self._tables[key] = (torch.zeros(M, dtype=F32, device=self.device), torch.ones(M, dtype=F32, device=self.device),
                                 torch.zeros(C, dtype=F32, device=self.device))
        return self._tables[key]

    def _block_fwd_cls(self, i, x, B, N, cos, sin, xb=None, st=None, lo=None):
        """Last teacher block restricted to what forward() consumes: the CLS row.  x fp32 [B*N, C] -> fp32 [B, C]; keys and values still
        come from every token.  Row-for-row the same arithmetic as _block_fwd.  With xb / st (/ lo) -- the bf16 operand view, ln_1 statistics
        (and low plane) the previous folded block left -- ln_1 is folded into the K|V GEMM like in every other block (no LayerNorm pass over
        the whole stream, which never returns to fp32) and only the B CLS rows are rebuilt in fp32 (EvaEngine._block_fwd_cls)."""
        ops, cfg = self.ops, self.cfg
        C, Hd, H, eps = cfg.width, cfg.hidden, cfg.heads, cfg.ln_eps
        b = f"{self.prefix}{self.BLOCK_TAG}{i}."
        M = B * N
        wqkv, bqkv = self.w[b + "attn.in_proj_weight"], self.p[b + "attn.in_proj_bias"]
        kv = ops.empty((M, 2 * C), BF16)
        q = ops.empty((B, C), BF16)
        if xb is not None:
            Wq, cq, dq = self.fold[i]["qkv"]
            ops.gemm_nt_ln(xb, Wq[C:], kv, bias=dq[C:], ln_mean=st[0], ln_rstd=st[1], ln_colsum=cq[C:], epi=EPI_BF16)
            xc = (self._join_planes(xb.view(B, N, C)[:, 0, :], lo.view(B, N, C)[:, 0, :]) if lo is not None
                  else x.view(B, N, C)[:, 0, :].contiguous())
            ln1c = ops.empty((B, C), BF16)
            ops.layernorm_fwd(xc, self.p[b + "ln_1.weight"], self.p[b + "ln_1.bias"], ln1c, None, None, eps)
            ops.gemm_nt(ln1c, wqkv[:C], q, bias=bqkv[:C], epi=EPI_BF16)
        else:
            ln1 = ops.empty((M, C), BF16)
            ops.layernorm_fwd(x, self.p[b + "ln_1.weight"], self.p[b + "ln_1.bias"], ln1, None, None, eps)
            ops.gemm_nt(ln1, wqkv[C:], kv, bias=bqkv[C:], epi=EPI_BF16)
            ops.gemm_nt(ln1.view(B, N, C)[:, 0, :], wqkv[:C], q, bias=bqkv[:C], epi=EPI_BF16)
            xc = x.view(B, N, C)[:, 0, :].contiguous()
        att = ops.empty((B, C), BF16)
        ops.attn_cls_fwd(q, kv, cos, sin, att, B, N, H, cfg.head_width ** -0.5)
        ops.gemm_nt(att, self.w[b + "attn.out_proj.weight"], xc, bias=self.p[b + "attn.out_proj.bias"], extra=xc, epi=EPI_RESID_F32)
        ln2 = ops.empty((B, C), BF16)
        ops.layernorm_fwd(xc, self.p[b + "ln_2.weight"], self.p[b + "ln_2.bias"], ln2, None, None, eps)
        hid = ops.empty((B, Hd), BF16)
        ops.gemm_nt(ln2, self.w[b + "mlp.c_fc.weight"], hid, bias=self.p[b + "mlp.c_fc.bias"], epi=EPI_QGELU_BF16 if cfg.quick_gelu else EPI_GELU_BF16)
        ops.gemm_nt(hid, self.w[b + "mlp.c_proj.weight"], xc, bias=self.p[b + "mlp.c_proj.bias"], extra=xc, epi=EPI_RESID_F32)
        return xc

    def _head(self, rows, out):
        """out[M,E] f32 = bf16(rows) . proj   (no bias: transformer.py:492-493,583-584)."""
        self.ops.gemm_nt(rows, self.wt["head_fwd"][:, :self.cfg.width], out, epi=EPI_F32)

    # ------------------------------------------------------------------------------------------ teacher
    def encode_image(self, images, chunk: int = 256):
        """Frozen-teacher path: every block with attention, ln_post on the CLS row, proj.  [K,3,S,S] -> fp32 [K,E]."""
        ops, cfg, P = self.ops, self.cfg, self.prefix
        K = images.shape[0]
        out = ops.empty((K, cfg.embed_dim), F32)
        fold_blocks = self.block_folds_active(images)          # the switch + the row-statistics guard of the current weights (EvaEngine)
        for k0 in range(0, K, chunk):
            img = images[k0:k0 + chunk]
            B = img.shape[0]
            x, g = self._stem(img)
            N = g * g + 1
            cos, sin = self.rope_tables(g)
            xf = x.view(B * N, cfg.width)
            last = cfg.layers - 1 if self.cls_only_last_block else cfg.layers
            xb = st = None
            cls_folded = fold_blocks and self.fold_cls_block and last < cfg.layers and last > 0      # the CLS-only block takes the planes + statistics as they are
            lo = ops.empty((B * N, cfg.width), torch.int16) if fold_blocks and self.split_stream and (last > 1 or cls_folded) else None
            try:
                for i in range(last):
                    self._rccl_window_step(i, k0)                # EvaEngine: leading blocks of a prefetched pass leave CUs to RCCL
                    if fold_blocks:
                        xb, st = self._teacher_block_folded(i, xf, xb, st, B, N, cos, sin, emit_next=i + 1 < last or cls_folded, lo=lo)
                    else:
                        self._block_fwd(i, xf, B, N, cos, sin, True, None, True)
            finally:
                self._rccl_window_close(k0)
            if last < cfg.layers:
                xc = self._block_fwd_cls(last, xf, B, N, cos, sin, xb if cls_folded else None, st, lo)
            else:
                xc = x[:, 0, :]
            cls = ops.empty((B, cfg.width), BF16)
            ops.layernorm_fwd(xc, self.p[P + "ln_post.weight"], self.p[P + "ln_post.bias"], cls, None, None, cfg.ln_eps)
            self._head(cls, out[k0:k0 + B])
        return out

    # ------------------------------------------------------------------------------------------ mask-attention pooling (inference)
    def mask_attn_pool(self, images, masks, chunk: int = 64):
        """VisionTransformer.mask_attn_pool / _mask_attn_pool (transformer.py:736-834), the pooling behind extract_type='v1' (:660-671) and
        encode_masks(mask_attn=True).  masks: list over images of bool [n_i, g, g] on the token grid.  Every mask is an extra token -- a copy
        of the image's CLS embedding after ln_pre -- that runs through ALL blocks; the reference's attention mask hides the extra tokens from
        everybody and lets token q see the CLS token plus the image tokens inside its mask, so the image tokens are exactly forward()'s and
        an extra token is a query-only passenger.  Per block: the image tokens' usual q|k|v GEMM, whose k|v columns also serve the Q extra
        queries of the image (cs_attn_query_fwd), then out_proj / MLP on the [B*Q, C] passenger stream with the same GEMM epilogues as the
        blocks.  Images with fewer masks are padded with see-everything tokens whose rows are dropped (:793-795,823-824).  -> fp32 [sum n_i, E]"""
        ops, cfg, P = self.ops, self.cfg, self.prefix
        C, Hd, H, eps, E = cfg.width, cfg.hidden, cfg.heads, cfg.ln_eps, cfg.embed_dim
        counts = [int(m.shape[0]) for m in masks]
        assert len(masks) == images.shape[0], "one mask list per image"
        Q = max(counts) if counts else 0
        if Q == 0:                                   # no image has a mask: transformer.py:823-824 returns zero rows
            return torch.zeros((0, E), dtype=F32, device=self.device)
        outs = []                                    # an image WITHOUT masks is all padding: see-everything queries whose rows are dropped (:793-795)
        act = EPI_QGELU_BF16 if cfg.quick_gelu else EPI_GELU_BF16
        for k0 in range(0, images.shape[0], chunk):
            img = images[k0:k0 + chunk]
            B = img.shape[0]
            x, g = self._stem(img)
            N = g * g + 1
            cos, sin = self.rope_tables(g)
            allow = torch.ones((B, Q, N), dtype=torch.uint8, device=self.device)
            for b, m in enumerate(masks[k0:k0 + B]):
                assert tuple(m.shape[1:]) == (g, g), f"masks live on the {g}x{g} token grid, got {tuple(m.shape[1:])}"
                allow[b, :m.shape[0], 1:] = m.reshape(m.shape[0], -1).to(device=self.device, dtype=torch.uint8)
            allow = allow.view(B * Q, N)
            xf = x.view(B * N, C)
            xm = x[:, :1, :].expand(B, Q, C).reshape(B * Q, C).contiguous()          # fp32 passenger stream
            MQ = B * Q
            for i in range(cfg.layers):
                b_ = f"{P}{self.BLOCK_TAG}{i}."
                wqkv, bqkv = self.w[b_ + "attn.in_proj_weight"], self.p[b_ + "attn.in_proj_bias"]
                ln1 = ops.empty((B * N, C), BF16)
                ops.layernorm_fwd(xf, self.p[b_ + "ln_1.weight"], self.p[b_ + "ln_1.bias"], ln1, None, None, eps)
                qkv = ops.empty((B * N, 3 * C), BF16)
                ops.gemm_nt(ln1, wqkv, qkv, bias=bqkv, epi=EPI_BF16)
                lnm = ops.empty((MQ, C), BF16)
                ops.layernorm_fwd(xm, self.p[b_ + "ln_1.weight"], self.p[b_ + "ln_1.bias"], lnm, None, None, eps)
                qm = ops.empty((MQ, C), BF16)
                ops.gemm_nt(lnm, wqkv[:C], qm, bias=bqkv[:C], epi=EPI_BF16)
                attm = ops.empty((MQ, C), BF16)
                ops.attn_query_fwd(qm, qkv[:, C:], allow, attm, B, Q, N, H, cfg.head_width ** -0.5)
                ops.gemm_nt(attm, self.w[b_ + "attn.out_proj.weight"], xm, bias=self.p[b_ + "attn.out_proj.bias"], extra=xm, epi=EPI_RESID_F32)
                ln2 = ops.empty((MQ, C), BF16)
                ops.layernorm_fwd(xm, self.p[b_ + "ln_2.weight"], self.p[b_ + "ln_2.bias"], ln2, None, None, eps)
                hid = ops.empty((MQ, Hd), BF16)
                ops.gemm_nt(ln2, self.w[b_ + "mlp.c_fc.weight"], hid, bias=self.p[b_ + "mlp.c_fc.bias"], epi=act)
                ops.gemm_nt(hid, self.w[b_ + "mlp.c_proj.weight"], xm, bias=self.p[b_ + "mlp.c_proj.bias"], extra=xm, epi=EPI_RESID_F32)
                if i + 1 < cfg.layers:                      # the image tokens move on (they never see the passengers); not needed after the last block
                    att = ops.empty((B * N, C), BF16)
                    ops.attn_fwd(qkv, cos, sin, att, None, B, N, H, cfg.head_width ** -0.5)
                    ops.gemm_nt(att, self.w[b_ + "attn.out_proj.weight"], xf, bias=self.p[b_ + "attn.out_proj.bias"], extra=xf, epi=EPI_RESID_F32)
                    ln2i = ops.empty((B * N, C), BF16)
                    ops.layernorm_fwd(xf, self.p[b_ + "ln_2.weight"], self.p[b_ + "ln_2.bias"], ln2i, None, None, eps)
                    hidi = ops.empty((B * N, Hd), BF16)
                    ops.gemm_nt(ln2i, self.w[b_ + "mlp.c_fc.weight"], hidi, bias=self.p[b_ + "mlp.c_fc.bias"], epi=act)
                    ops.gemm_nt(hidi, self.w[b_ + "mlp.c_proj.weight"], xf, bias=self.p[b_ + "mlp.c_proj.bias"], extra=xf, epi=EPI_RESID_F32)
            lnp = ops.empty((MQ, C), BF16)
            ops.layernorm_fwd(xm, self.p[P + "ln_post.weight"], self.p[P + "ln_post.bias"], lnp, None, None, eps)
            pooled = ops.empty((MQ, E), F32)
            self._head(lnp, pooled)
            pooled = pooled.view(B, Q, E)
            outs += [pooled[b, :n] for b, n in enumerate(counts[k0:k0 + B])]
        return torch.cat(outs)

    # ------------------------------------------------------------------------------------------ student
    def encode_dense(self, images, need_grad: bool = False):
        ops, cfg, P = self.ops, self.cfg, self.prefix
        B = images.shape[0]
        stem_keep = {} if (need_grad and self.stem_level > 0) else None
        x, g = self._stem(images, stem_keep)
        N, C, E = g * g + 1, cfg.width, cfg.embed_dim
        cos, sin = self.rope_tables(g)
        xf = x.view(B * N, C)
        saves = {}
        for i in range(cfg.layers):
            keep = need_grad and i >= self.first_trainable
            save = {} if keep else None
            xf = self._block_fwd(i, xf, B, N, cos, sin, with_attn=(i < cfg.layers - 1), save=save, inplace=not keep)
            if keep:
                saves[i] = save
        M = B * N
        lnf = ops.empty((M, C), BF16)
        mean = ops.empty((M,), F32) if need_grad else None
        rstd = ops.empty((M,), F32) if need_grad else None
        ops.layernorm_fwd(xf, self.p[P + "ln_post.weight"], self.p[P + "ln_post.bias"], lnf, mean, rstd, cfg.ln_eps)
        feats = ops.empty((M, E), F32)
        self._head(lnf, feats)
        dense = ops.empty((M, E), F32)
        inv = ops.empty((M,), F32)
        ops.l2norm_fwd(feats, dense, inv)
        if need_grad:
            self._ctx = dict(B=B, N=N, g=g, saves=saves, xL=xf, stf=(mean, rstd), dense=dense, inv=inv, cos=cos, sin=sin, stem=stem_keep,
                             lnf=lnf if self.train_all else None)
        return dense.view(B, N, E), g

    # ------------------------------------------------------------------------------------------ backward
    def _block_bwd(self, i, s, g, gb, B, N, cos, sin, ws, next_bias=None):
        """g: fp32 [M,C] gradient w.r.t. the block output; updated in place to the gradient w.r.t. its input.  gb / next_bias: as in
        EvaEngine._block_bwd (the bf16 copy of g and its column sums come out of the LayerNorm backwards)."""
        ops, cfg = self.ops, self.cfg
        C, Hd, H = cfg.width, cfg.hidden, cfg.heads
        b = f"{self.prefix}{self.BLOCK_TAG}{i}."
        M = B * N
        G = self.g
        # ---- MLP: x2 = x1 + c_proj(act(c_fc(ln_2 x1))) --------------------------------------------
        self._wgrad(gb, s["hid"], G[b + "mlp.c_proj.weight"])
        d_hid = ops.empty((M, Hd), BF16)
        ops.gemm_nt(gb, self.wt[(i, "cproj")][:, :C], d_hid, epi=EPI_BF16)                   # [M,C] . c_proj[C,Hd]
        d_fc = ops.empty((M, Hd), BF16)
        ops.gelu_bwd(d_hid, s["fc"], d_fc, cfg.quick_gelu)
        ops.colsum_bf16(d_fc, G[b + "mlp.c_fc.bias"], ws[1])
        self._wgrad(d_fc, s["ln2"], G[b + "mlp.c_fc.weight"])
        d_ln2 = ops.empty((M, C), BF16)
        ops.gemm_nt(d_fc, self.wt[(i, "fc")][:, :Hd], d_ln2, epi=EPI_BF16)                   # [M,Hd] . c_fc[Hd,C]
        ops.layernorm_bwd(d_ln2, s["x1"], self.p[b + "ln_2.weight"], *s["st2"], g, DX_F32_ACCUM,
                          G[b + "ln_2.weight"], G[b + "ln_2.bias"], True, ws[0], dx_copy=gb, copy_colsum=G[b + "attn.out_proj.bias"])
        # ---- attention branch: x1 = x0 + out_proj(att) ------------------------------------------
        self._wgrad(gb, s["att"], G[b + "attn.out_proj.weight"])
        d_att = ops.empty((M, C), BF16)
        ops.gemm_nt(gb, self.wt[(i, "proj")][:, :C], d_att, epi=EPI_BF16)
        d_ln1 = ops.empty((M, C), BF16)
        Gw, Gb = G[b + "attn.in_proj_weight"], G[b + "attn.in_proj_bias"]
        if s["with_attn"]:
            d_qkv = ops.empty((M, 3 * C), BF16)
            ops.attn_bwd(s["qkv"], s["att"], d_att, s["lse"], cos, sin, d_qkv, ws[0], B, N, H, cfg.head_width ** -0.5)
            ops.colsum_bf16(d_qkv, Gb, ws[1])                                                # q, k and v all carry a bias here
            self._wgrad(d_qkv, s["ln1"], Gw)
            ops.gemm_nt(d_qkv, self.wt[(i, "qkv")][:, :3 * C], d_ln1, epi=EPI_BF16)
        else:
            ops.colsum_bf16(d_att, Gb[2 * C:], ws[1])                                        # q/k rows keep their zero gradient
            self._wgrad(d_att, s["ln1"], Gw[2 * C:])
            ops.gemm_nt(d_att, self.wt[(i, "qkv")][:, 2 * C:3 * C], d_ln1, epi=EPI_BF16)
        ops.layernorm_bwd(d_ln1, s["x0"], self.p[b + "ln_1.weight"], *s["st1"], g, DX_F32_ACCUM,
                          G[b + "ln_1.weight"], G[b + "ln_1.bias"], True, ws[0], dx_copy=gb if next_bias is not None else None,
                          copy_colsum=next_bias)

    def backward_dense(self, d_dense):
        ops, cfg, P = self.ops, self.cfg, self.prefix
        c = self._ctx
        if c is None:
            raise RuntimeError("backward_dense() without a preceding encode_dense(need_grad=True)")
        self._ctx = None
        B, N, C, E = c["B"], c["N"], cfg.width, cfg.embed_dim
        M = B * N
        d_feats = ops.empty((M, E), BF16)
        ops.l2norm_bwd(d_dense.reshape(M, E), c["dense"], c["inv"], d_feats)
        d_lnf = ops.empty((M, C), BF16)
        ops.gemm_nt(d_feats, self.w[P + "proj"], d_lnf, epi=EPI_BF16)                        # [M,E] . proj^T: proj [C,E] is already "W^T"; frozen
        g = ops.empty((M, C), F32)
        gb = ops.empty((M, C), BF16)
        ws_bytes = max(ops.layernorm_bwd_workspace(M, C), ops.attn_bwd_workspace(B, N, cfg.heads))
        ws = (ops.empty((ws_bytes,), torch.uint8), ops.empty((max(ops.colsum_workspace(M, max(cfg.hidden, 3 * C)), 4),), torch.uint8))
        L, first = cfg.layers, self.first_trainable
        cproj_bias = lambda i: self.g[f"{P}{self.BLOCK_TAG}{i}.mlp.c_proj.bias"] if i >= first else None
        if self.train_all:
            # proj [C,E] (transformer.py:583-584: tokens @ proj) and ln_post train: dproj = LN(x)^T . dfeats; the CLS rows of d_feats are exact
            # zeros (the dense map drops them), so they add nothing
            self._wgrad(c["lnf"], d_feats, self.g[P + "proj"])
            ops.layernorm_bwd(d_lnf, c["xL"], self.p[P + "ln_post.weight"], *c["stf"], g, DX_F32_ASSIGN, self.g[P + "ln_post.weight"],
                              self.g[P + "ln_post.bias"], True, ws[0], dx_copy=gb, copy_colsum=cproj_bias(L - 1))
            if self.grad_ready_hook is not None:
                self.grad_ready_hook("head")
        else:
            ops.layernorm_bwd(d_lnf, c["xL"], self.p[P + "ln_post.weight"], *c["stf"], g, DX_F32_ASSIGN, None, None, True, ws[0],   # ln_post frozen
                              dx_copy=gb if first < L else None, copy_colsum=cproj_bias(L - 1))                                    # (transformer.py:405)
        for i in range(L - 1, first - 1, -1):
            self._block_bwd(i, c["saves"].pop(i), g, gb, B, N, c["cos"], c["sin"], ws, next_bias=cproj_bias(i - 1) if i > 0 else None)
            if self.grad_ready_hook is not None:
                self.grad_ready_hook(i)
        if c["stem"] is not None:
            self._stem_bwd(g, c["stem"], B, N, c["g"], ws[0])
            if self.grad_ready_hook is not None:
                self.grad_ready_hook("stem")

    def _stem_bwd(self, g, keep, B, N, grid, ws):
        """Gradients of the stem from g = d loss / d (ln_pre output) fp32 [B*N, C]  (transformer.py:551-569: x = ln_pre(cat(class_embedding,
        conv1(img)) + positional_embedding)).  ln_pre backward (its gamma / beta at stem level 2), then: positional_embedding <- sum over images
        (through the bicubic rescale for a non-native grid, :724-734), class_embedding <- the CLS rows, conv1.weight <- dY^T . im2row(images) (no
        bias).  Once per step on [B*N, C]; the row bookkeeping is tensor code, the LayerNorm backward and the contraction are the kernels."""
        ops, cfg, P = self.ops, self.cfg, self.prefix
        C, M = cfg.width, B * N
        lvl2 = self.stem_level >= 2
        d_pre = ops.empty((M, C), F32)
        ops.layernorm_bwd(g.to(BF16), keep["x_pre"], self.p[P + "ln_pre.weight"], *keep["st"], d_pre, DX_F32_ASSIGN,
                          self.g[P + "ln_pre.weight"] if lvl2 else None, self.g[P + "ln_pre.bias"] if lvl2 else None, True, ws)
        d3 = d_pre.view(B, N, C)
        d_pos = d3.sum(dim=0)                                                    # [N, C]
        gpos = self.g[P + "positional_embedding"]                                # [native N, C]
        if grid == cfg.grid:
            gpos.add_(d_pos)
        else:
            gpos[0].add_(d_pos[0])
            with torch.enable_grad():
                pe = self.p[P + "positional_embedding"].detach()[1:].T.reshape(1, C, cfg.grid, cfg.grid).clone().requires_grad_(True)
                out = F.interpolate(pe, (grid, grid), mode="bicubic", align_corners=False)
                (d_pe,) = torch.autograd.grad(out, pe, d_pos[1:].T.reshape(1, C, grid, grid))
            gpos[1:].add_(d_pe.reshape(C, cfg.grid * cfg.grid).T)
        if lvl2:
            self.g[P + "class_embedding"].view(C).add_(d_pos[0])
            gp = d3[:, 1:, :].to(BF16).reshape(B * (N - 1), C)                   # patch rows, in the im2row matrix's row order
            self._wgrad(gp, keep["patches"], self.storage_of(self.grad, P + "conv1.weight"))
