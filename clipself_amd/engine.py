"""Step engine of the EVA02 vision tower on the CLIPSelf hot path: an explicit forward / hand-written backward
schedule over the C-ABI kernels (clipself_amd/hip.py), with all parameters in flat buffers.

No tracing compiler and no autograd graph inside the tower: the schedule below *is* the program.  What each stage
computes, and where the reference does it (paths under /root/reference/src/open_clip/eva_clip/):

  stem            eva_vit_model.py:537-544   patch-embed conv (im2row + GEMM) + cls_token + pos_embed
  block           eva_vit_model.py:300-307   x += attn(norm1(x)); x += mlp(norm2(x))
  attention       eva_vit_model.py:174-247   q/k/v proj (+q_bias, none, +v_bias) -> RoPE(q,k) -> softmax(qk^T/8)v
                                              -> inner_attn_ln -> proj                 (rope.py:148-164)
  last dense blk  eva_vit_model.py:249-256,317-324   proj(inner_attn_ln(v_proj(norm1 x)+v_bias)), no attention
  SwiGLU          eva_vit_model.py:98-105    w3(ffn_ln(silu(w1 x) * (w2 x)))
  teacher head    eva_vit_model.py:565-569,585   head(norm(x)[:,0])
  dense head      eva_vit_model.py:615-623   normalize(head(norm(x[:,1:])))
  RoI pooling     eva_vit_model.py:625-629,655-664

Memory layout (MI355X: 288 GB HBM3E -- nothing is recomputed, nothing is re-laid-out):
  * one fp32 master buffer for every parameter of the tower, tensors back-to-back on 64-element boundaries, in
    layer order; [Wq;Wk;Wv], [q_bias;0;v_bias], [W1;W2], [b1;b2] are *adjacent* so the fused QKV / SwiGLU GEMMs read
    them as single matrices without any packing step;
  * a same-layout bf16 shadow (MFMA operand), and for the student same-layout fp32 grad / exp_avg / exp_avg_sq (one
    flat AdamW launch, one contiguous all-reduce bucket per block) plus transposed bf16 shadows for the dgrad GEMMs;
  * residual stream fp32 [B*N, C]; every GEMM operand bf16; LayerNorm / softmax statistics fp32.

The engine is backend-agnostic on purpose: `ops` is clipself_amd.hip.HipOps in the product; the CPU test-suite
injects the per-kernel references (oracle/ops_ref.py) to verify this schedule -- in particular the hand-written
backward -- against the monolithic autograd oracle without a GPU.  The product never constructs anything but HipOps.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict

import torch
import torch.nn.functional as F

from .config import TowerCfg

BF16, F32 = torch.bfloat16, torch.float32
EPI_BF16, EPI_F32, EPI_RESID_F32, EPI_SWIGLU_BF16, EPI_ATOMIC_F32, EPI_PATCH_F32, EPI_RESID_LN_F32, EPI_GELU_BF16, EPI_QGELU_BF16 = range(9)
DX_BF16, DX_F32_ASSIGN, DX_F32_ACCUM = range(3)
ALIGN = 64


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def padded_hidden(cfg: TowerCfg) -> int:
    return _round_up(cfg.hidden, 64)


def param_groups_layout(cfg: TowerCfg, prefix: str = "visual."):
    """Allocation groups (tensors of a group are contiguous, group starts are 64-aligned), in layer order.
    Entries are (name, logical shape, storage shape): names/logical shapes are the reference's state-dict entries; the
    storage of the SwiGLU hidden dimension and of the patch-embed contraction is zero-padded to a multiple of 64 so
    that every GEMM sees K % 64 == 0 and 16-byte rows (EVA02-L/14: hidden 2730 -> 2752, 3*14*14 = 588 -> 640; for
    B/16 both are already multiples of 64 and storage == logical).  Names containing '._' are private padding."""
    C, Hd, E, p = cfg.width, cfg.hidden, cfg.embed_dim, cfg.patch_size
    Hp, Kpe, Kp = padded_hidden(cfg), 3 * p * p, _round_up(3 * p * p, 64)
    t = lambda name, shape, storage=None: (name, shape, storage if storage is not None else shape)
    groups = [[t(prefix + "cls_token", (1, 1, C))], [t(prefix + "pos_embed", (1, cfg.tokens, C))],
              [t(prefix + "patch_embed.proj.weight", (C, 3, p, p), (C, Kp))], [t(prefix + "patch_embed.proj.bias", (C,))]]
    for i in range(cfg.layers):
        b = f"{prefix}blocks.{i}."
        groups += [
            [t(b + "norm1.weight", (C,))], [t(b + "norm1.bias", (C,))],
            [t(b + "attn.q_proj.weight", (C, C)), t(b + "attn.k_proj.weight", (C, C)), t(b + "attn.v_proj.weight", (C, C))],
            [t(b + "attn.q_bias", (C,)), t(b + "attn._k_bias_zero", (C,)), t(b + "attn.v_bias", (C,))],
            [t(b + "attn.inner_attn_ln.weight", (C,))], [t(b + "attn.inner_attn_ln.bias", (C,))],
            [t(b + "attn.proj.weight", (C, C))], [t(b + "attn.proj.bias", (C,))],
            [t(b + "norm2.weight", (C,))], [t(b + "norm2.bias", (C,))],
            [t(b + "mlp.w1.weight", (Hd, C), (Hp, C)), t(b + "mlp.w2.weight", (Hd, C), (Hp, C))],
            [t(b + "mlp.w1.bias", (Hd,), (Hp,)), t(b + "mlp.w2.bias", (Hd,), (Hp,))],
            [t(b + "mlp.ffn_ln.weight", (Hd,), (Hp,))], [t(b + "mlp.ffn_ln.bias", (Hd,), (Hp,))],
            [t(b + "mlp.w3.weight", (C, Hd), (C, Hp))], [t(b + "mlp.w3.bias", (C,))],
        ]
    groups += [[t(prefix + "norm.weight", (C,))], [t(prefix + "norm.bias", (C,))],
               [t(prefix + "head.weight", (E, C))], [t(prefix + "head.bias", (E,))]]
    return groups


def is_no_decay(name: str, ndim: int) -> bool:
    """AdamW grouping rule of the reference (src/training/main.py:199)."""
    return ndim < 2 or "bn" in name or "ln" in name or "bias" in name or "logit_scale" in name


class EvaEngine:
    BLOCK_TAG = "blocks."                     # state-dict name of the block list below the tower prefix
    FP8_MAX_ROW = 8192                        # widest row cs_quant_rows_fp8 quantises (one row per wave, held in registers)

    def _layout(self):
        return param_groups_layout(self.cfg, self.prefix)

    def block_index(self, name: str):
        """Index of the transformer block a state-dict name belongs to, None for stem / head tensors."""
        tag = self.prefix + self.BLOCK_TAG
        return int(name[len(tag):].split(".")[0]) if name.startswith(tag) else None

    def _never_reached(self, i: int, name: str) -> bool:
        """Tensors of block i the dense path never differentiates.  The last block runs without attention, so its q/k projections
        and q_bias never get a gradient and torch's AdamW skips them (no decay either) -- SURVEY.md D7."""
        return i == self.cfg.layers - 1 and (name.rsplit(".", 2)[-2:] in (["q_proj", "weight"], ["k_proj", "weight"])
                                             or name.endswith("attn.q_bias"))

    def __init__(self, cfg: TowerCfg, ops, trainable: bool = False, prefix: str = "visual."):
        self.cfg, self.ops, self.prefix, self.trainable = cfg, ops, prefix, trainable
        self.offsets = OrderedDict()          # name -> (offset, storage shape)
        self.logical = {}                     # name -> logical (reference) shape
        off = 0
        self.block_ranges = []                # flat [begin, end) of each block (contiguous all-reduce buckets)
        cur_block, blk_begin = None, 0
        for grp in self._layout():
            off = _round_up(off, ALIGN)
            blk = self.block_index(grp[0][0])
            if blk != cur_block:
                if cur_block is not None:
                    self.block_ranges.append((blk_begin, off))
                cur_block, blk_begin = blk, off
            for name, shape, storage in grp:
                self.offsets[name] = (off, storage)
                self.logical[name] = shape
                off += math.prod(storage)
        self.numel = _round_up(off, 256)
        # stem (cls_token, pos_embed, patch embedding) and head (final norm, head) slices of the flat store: two more gradient buckets
        # when the whole tower trains (training without --lock-image)
        self.stem_range = (0, self.block_ranges[0][0]) if self.block_ranges else (0, 0)
        self.head_range = (self.block_ranges[-1][1], self.numel) if self.block_ranges else (0, self.numel)
        self.master = ops.zeros((self.numel,), F32)
        self.shadow = ops.zeros((self.numel,), BF16)
        self.device = self.master.device
        self.Hp = padded_hidden(cfg)
        self.Kpe = _round_up(3 * cfg.patch_size * cfg.patch_size, 64)
        self.p = {n: self.view_of(self.master, n) for n in self.offsets}
        self.w = {n: self.view_of(self.shadow, n) for n in self.offsets}
        self._tables = {}
        self._pos_cache = {}
        self.grad = self.exp_avg = self.exp_avg_sq = self.flags = None
        self.g = {}
        self.wt = {}
        self.first_trainable = cfg.layers      # no block trainable until lock()/unlock is applied
        self.train_all = False                 # stem + final norm + head train as well (set_trainable_all: training without --lock-image)
        self.flags_version = 0                 # bumped whenever the trainable / decay flag bytes are rewritten (_set_flags)
        self.grad_ready_hook = None            # callable(block_index) fired when a block's grads are complete
        self._ctx = None
        self._wgrad_ws = None
        # encode_image() consumes only the CLS row, so the last block runs its query/proj/MLP for that row alone (keys and
        # values still span all tokens); False runs the last block over every token -- same outputs, ~1/L more work.
        self.cls_only_last_block = True
        # Frozen towers fold the two sub-LayerNorms (inner_attn_ln ahead of proj, ffn_ln ahead of w3) into the following GEMM:
        # gamma goes into a bf16 copy of the weight, beta and the row statistics into the GEMM epilogue, and the statistics
        # come out of the producing kernels' epilogues -- the LN passes over [M,C] and [M,hidden] disappear (_block_post_folded).
        self.fold_sub_ln = not trainable
        # ... and, in encode_image(), the block LayerNorms norm1 / norm2 as well: the residual GEMMs also emit a bf16 copy of the new
        # stream and its row statistics, the q|k|v and W1|W2 GEMMs apply the normalisation in their epilogues (_teacher_block_folded)
        self.fold_block_ln = not trainable
        # Guard of that fold.  The folded norm1 / norm2 contract the UN-centred bf16 row and remove the mean afterwards in fp32
        # (rstd * (bf16(x) . W gamma - mean * colsum)): the bf16 rounding of x is relative to |x|, not to |x - mean|, so the error of the
        # normalised row grows like sqrt(1 + (mean / sigma)^2).  Measured against the plain schedule on weights with trained-like statistics
        # (oracle/stress_weights.py, profiles/r04_parity.md): outlier channels x200 alone -- folded is CLOSER to fp32 than the plain bf16
        # schedule; |mean| / sigma = 2 -- equal; 3 -- 1.6x; 5 -- 2.8x.  So the first encode_image() after a weight load measures
        # mean_rows(|row mean| / row sigma) of the stream entering every block on a few crops (block_fold_statistic, one host read-back)
        # and keeps norm1 / norm2 as LayerNorm kernels when it exceeds block_fold_limit; the sub-LayerNorm folds are never worse than
        # the plain schedule (their input is a stored bf16 tensor either way) and stay.
        self.block_fold_guard = not trainable
        self.block_fold_limit = 2.0
        self.block_fold_ratio = None           # the measured statistic (None: not measured since the last weight load)
        # ... with the residual stream between those GEMMs held as two 16-bit planes (bf16 view + remainder, exact; cs_gemm_nt_ln_split):
        # 8 instead of 10 bytes of HBM traffic per stream element and residual GEMM
        self.split_stream = not trainable
        self.fold = {}
        # BASELINE configs[4] "fp8 MFMA weights": the forward linears of the non-folded (training / dense) schedule run on e4m3 operands --
        # weight shadows quantised per output row (refreshed after every AdamW step), activations per token row by cs_quant_rows_fp8,
        # contraction by the block-scaled fp8 MFMA, fp32 accumulate; backward (dgrad / wgrad) keeps the bf16 operands.  enable_fp8_forward().
        self.fp8_forward = False
        self.w8 = {}
        # enable_fp8_forward(dgrad=True): the four dgrad GEMMs of a block (dX = dY . W) also contract e4m3 x e4m3 -- dY quantised per token
        # row, the transposed weight shadows per input-feature row, both scales factor out of the contraction over the output features;
        # wgrad (contraction over tokens: per-row scales do not factor out) stays bf16
        self.fp8_dgrad = False
        self.wt8 = {}
        # (leading blocks, CUs): encode_image()'s persistent GEMMs leave that many compute units free in its first blocks -- set by
        # CLIPSelf.prefetch_teacher in data-parallel runs, where those blocks run beside the student's gradient all-reduce
        self.rccl_window = (0, 0)
        self.wgrad_tn = True                   # weight gradients from the token-major operands (no transposed copies) where the shape allows
        # the SwiGLU backward also reduces its output's columns (the w1 | w2 bias gradients): bit-identical to the separate colsum pass
        # (A/B switch CLIPSELF_NO_FUSED_SWIGLU_COLSUM=1)
        self.fused_swiglu_colsum = os.environ.get("CLIPSELF_NO_FUSED_SWIGLU_COLSUM") != "1"
        if trainable:
            self.grad = ops.zeros((self.numel,), F32)
            self.exp_avg = ops.zeros((self.numel,), F32)
            self.exp_avg_sq = ops.zeros((self.numel,), F32)
            self.g = {n: self.view_of(self.grad, n) for n in self.offsets}
            self.flags = torch.zeros(self.numel // 64, dtype=torch.uint8, device=self.device)

    # ------------------------------------------------------------------------------------------ parameters
    def storage_of(self, buf, name):
        """The (possibly zero-padded) storage block of `name` inside a flat buffer, in its storage shape."""
        o, st = self.offsets[name]
        return buf[o:o + math.prod(st)].view(st)

    def view_of(self, buf, name):
        """The reference-shaped tensor of `name` inside a flat buffer: a plain view, or a strided view of the leading
        block when the storage is padded."""
        o, st = self.offsets[name]
        shape = self.logical[name]
        flat = buf[o:o + math.prod(st)]
        if tuple(st) == tuple(shape) or math.prod(st) == math.prod(shape):
            return flat.view(shape)
        if len(st) == 1:
            return flat[:shape[0]]
        if len(shape) == 2:
            return flat.view(st)[:shape[0], :shape[1]]
        inner = [1]
        for d in reversed(shape[2:]):
            inner.insert(0, inner[0] * d)                 # contiguous strides of the trailing dims
        return torch.as_strided(flat, shape, (st[1], *inner))

    def public_names(self):
        return [n for n in self.offsets if "._" not in n]

    def load_state(self, sd: dict, strict: bool = True):
        missing = []
        with torch.no_grad():
            for n in self.public_names():
                if n in sd:
                    self.p[n].copy_(sd[n].to(self.device, F32).reshape(self.p[n].shape))
                else:
                    missing.append(n)
        if strict and missing:
            raise KeyError(f"missing keys: {missing[:5]}... ({len(missing)})")
        self.sync_shadow()
        return missing

    def sync_shadow(self):
        """bf16 MFMA operands from the fp32 masters (after a load; AdamW refreshes them itself each step)."""
        self.ops.cast_f32_bf16(self.master, self.shadow)
        self._pos_cache.clear()
        self.block_fold_ratio = None
        if self.trainable:
            self.sync_transposed()
        if self.fold_sub_ln:
            self._build_folds()
        if self.fp8_forward:
            self.sync_fp8()

    def _build_folds(self):
        """gamma (.) W in bf16, its row sums, and W.beta + b for proj and w3 of every block (one-time, after a weight load)."""
        cfg, C, Hl = self.cfg, self.cfg.width, self.cfg.hidden
        self.fold = {}
        with torch.no_grad():
            for i in range(cfg.layers):
                b = f"{self.prefix}blocks.{i}."
                out = {}
                for key, wname, ln, bname, K in (("proj", "attn.proj.weight", "attn.inner_attn_ln", "attn.proj.bias", C),
                                                 ("w3", "mlp.w3.weight", "mlp.ffn_ln", "mlp.w3.bias", Hl)):
                    W = self.storage_of(self.master, b + wname)                    # [C, K padded]
                    g, beta = self.p[b + ln + ".weight"][:K], self.p[b + ln + ".bias"][:K]
                    Wf = torch.zeros_like(W, dtype=BF16)
                    Wf[:, :K] = (W[:, :K] * g[None, :]).to(BF16)
                    out[key] = (Wf, Wf.float().sum(dim=1).contiguous(), (W[:, :K] @ beta + self.p[b + bname]).contiguous())
                if self.fold_block_ln:
                    # norm1 -> [Wq;Wk;Wv] and norm2 -> [W1;W2]: the stacked matrices and biases are adjacent in the flat store
                    Hd = self.Hp
                    for key, wname, bias_name, ln, rows in (("qkv", "attn.q_proj.weight", "attn.q_bias", "norm1", 3 * C),
                                                           ("w12", "mlp.w1.weight", "mlp.w1.bias", "norm2", 2 * Hd)):
                        o, ob = self.offsets[b + wname][0], self.offsets[b + bias_name][0]
                        W, bias = self.master[o:o + rows * C].view(rows, C), self.master[ob:ob + rows]
                        g, beta = self.p[b + ln + ".weight"], self.p[b + ln + ".bias"]
                        Wf = (W * g[None, :]).to(BF16).contiguous()
                        out[key] = (Wf, Wf.float().sum(dim=1).contiguous(), (W @ beta + bias).contiguous())
                self.fold[i] = out

    def _wt_alloc(self, key, rows, cols):
        t = self.wt.get(key)
        if t is None:
            t = self.ops.zeros((cols, _round_up(rows, 64)), BF16)
            self.wt[key] = t
        return t

    def sync_transposed(self, blocks=None):
        """W^T shadows for the dgrad GEMMs (dx = dy . W needs W with the contraction dimension contiguous)."""
        cfg, C, Hd = self.cfg, self.cfg.width, self.Hp
        pairs = []
        for i in (range(self.first_trainable, cfg.layers) if blocks is None else blocks):
            b = f"{self.prefix}blocks.{i}."
            o = self.offsets[b + "attn.q_proj.weight"][0]
            pairs.append((self.shadow[o:o + 3 * C * C].view(3 * C, C), self._wt_alloc((i, "qkv"), 3 * C, C)))
            pairs.append((self.w[b + "attn.proj.weight"], self._wt_alloc((i, "proj"), C, C)))
            o = self.offsets[b + "mlp.w1.weight"][0]
            pairs.append((self.shadow[o:o + 2 * Hd * C].view(2 * Hd, C), self._wt_alloc((i, "w12"), 2 * Hd, C)))
            pairs.append((self.storage_of(self.shadow, b + "mlp.w3.weight"), self._wt_alloc((i, "w3"), C, Hd)))
        pairs.append((self.w[self.prefix + "head.weight"], self._wt_alloc("head", self.cfg.embed_dim, C)))
        self.ops.transpose_bf16_batched(pairs)          # one launch (49 matrices per step for B/16)

    # ------------------------------------------------------------------------------------------ fp8 forward operands
    def _fp8_rows(self, X):
        """bf16 [M,K] -> (e4m3 bytes [M, K padded to 128], fp32 row scales [M])."""
        M, K = X.shape
        q = self.ops.empty((M, _round_up(K, 128)), torch.uint8)
        sc = self.ops.empty((M,), F32)
        self.ops.quant_rows_fp8(X, q, sc)
        return q, sc

    def sync_fp8(self, blocks=None):
        """e4m3 shadows (+ per-output-row scales) of the four weight matrices of every block, from the bf16 shadows."""
        cfg, C, Hd = self.cfg, self.cfg.width, self.Hp
        for i in (range(cfg.layers) if blocks is None else blocks):
            b = f"{self.prefix}blocks.{i}."
            o = self.offsets[b + "attn.q_proj.weight"][0]
            self.w8[(i, "qkv")] = self._fp8_rows(self.shadow[o:o + 3 * C * C].view(3 * C, C))
            self.w8[(i, "proj")] = self._fp8_rows(self.w[b + "attn.proj.weight"])
            o = self.offsets[b + "mlp.w1.weight"][0]
            self.w8[(i, "w12")] = self._fp8_rows(self.shadow[o:o + 2 * Hd * C].view(2 * Hd, C))
            self.w8[(i, "w3")] = self._fp8_rows(self.storage_of(self.shadow, b + "mlp.w3.weight"))
            if self.fp8_dgrad and i >= self.first_trainable:
                for key, n in (("qkv", 3 * C), ("proj", C), ("w12", 2 * Hd), ("w3", C)):
                    self.wt8[(i, key)] = self._fp8_rows(self.wt[(i, key)][:, :n])

    def _dgrad(self, i, key, dY, out, cols=None, q=None):
        """out (bf16) = dY . W for the linear `key` of block i, through the transposed shadow (contraction over the output features, or over
        the column range `cols` of them); with fp8_dgrad on e4m3 operands."""
        n = dY.shape[1]
        lo, hi = cols if cols is not None else (0, n)
        if not self.fp8_dgrad or lo % 128 or (hi - lo) % 8:
            self.ops.gemm_nt(dY, self.wt[(i, key)][:, lo:hi], out, epi=EPI_BF16)
            return
        q, sy = q if q is not None else self._fp8_rows(dY)            # q: (codes, scales) already produced by the kernel that wrote dY
        w8, sw = self.wt8[(i, key)]
        self.ops.gemm_nt_f8(q, w8[:, lo:lo + q.shape[1]], out, sy, sw, epi=EPI_BF16)

    def enable_fp8_forward(self, on: bool = True, dgrad: bool = False):
        if on and not self.trainable:
            raise RuntimeError("fp8 forward operands belong to the training schedule; a frozen tower keeps its bf16 operands "
                               "(teacher targets and evaluation features must not depend on the run's precision flag)")
        widest = max(self.cfg.width, self.Hp) if not dgrad else max(3 * self.cfg.width, 2 * self.Hp)
        if on and widest > self.FP8_MAX_ROW:
            raise NotImplementedError(f"amp_fp8: cs_quant_rows_fp8 keeps a row in registers and covers rows up to {self.FP8_MAX_ROW} wide; "
                                      f"this tower's widest GEMM operand is {widest}")
        self.fp8_forward = bool(on)
        self.fp8_dgrad = bool(on and dgrad)
        self.w8, self.wt8 = {}, {}
        if on:
            self.sync_fp8()

    def _ln(self, x, gamma, beta, y, mean, rstd, eps):
        """LayerNorm forward; under fp8_forward also the e4m3 copy of its output (cs_layernorm_fwd_q8) for the linear that consumes it.
        Returns (q8, scale) or None."""
        if not self.fp8_forward:
            self.ops.layernorm_fwd(x, gamma, beta, y, mean, rstd, eps)
            return None
        M = x.shape[0]
        q = self.ops.empty((M, _round_up(y.shape[1], 128)), torch.uint8)
        sc = self.ops.empty((M,), F32)
        self.ops.layernorm_fwd(x, gamma, beta, y, mean, rstd, eps, q8=q, q_scale=sc)
        return q, sc

    def _linear(self, i, key, X, W, out, bias, extra=None, epi=EPI_BF16, rows=None, xq=None):
        """out = X . W^T + bias (+ extra): the bf16 MFMA GEMM, or -- fp8_forward -- the e4m3 GEMM on the quantised copy of X (xq: already
        produced by the LayerNorm that wrote X) and the weight's e4m3 shadow (`rows` = row range of the stacked weight that W is a slice of)."""
        if not self.fp8_forward:
            self.ops.gemm_nt(X, W, out, bias=bias, extra=extra, epi=epi)
            return
        xq, sx = xq if xq is not None else self._fp8_rows(X)
        w8, sw = self.w8[(i, key)]
        if rows is not None:
            w8, sw = w8[rows[0]:rows[1]], sw[rows[0]:rows[1]]
        self.ops.gemm_nt_f8(xq, w8, out, sx, sw, bias=bias, extra=extra, epi=epi)

    def set_trainable_blocks(self, unlocked_groups: int):
        """visual.lock(unlocked_groups) (eva_vit_model.py:500-516): only the last n blocks train
        (blocks[-0:] is the whole list, as in the reference)."""
        L = self.cfg.layers
        self.first_trainable = L - unlocked_groups if 0 < unlocked_groups <= L else 0
        self.train_all = False
        self._set_flags()

    def set_trainable_all(self):
        """No lock at all (training.main without --lock-image, src/training/main.py:161-166): every parameter of the visual tower trains --
        besides the blocks the stem (cls_token, pos_embed, patch_embed.proj), the final norm and the head, which the dense path
        differentiates as well (eva_vit_model.py:537-544,615-623)."""
        self.first_trainable = 0
        self.train_all = True
        self._set_flags()

    def _set_flags(self):
        if not self.trainable:
            return
        self.flags.zero_()
        names = list(self.offsets)
        for k, name in enumerate(names):
            o, s = self.offsets[name]
            i = self.block_index(name)
            if "._" in name or (i is None and not self._nonblock_trains(name)) or (i is not None and (i < self.first_trainable or self._never_reached(i, name))):
                continue
            n = math.prod(s)
            nxt = self.offsets[names[k + 1]][0] if k + 1 < len(names) else self.numel
            n64 = _round_up(n, 64)            # a tensor that ends its 64-aligned allocation group short of a flag granule (tiny head.bias)
            assert o % 64 == 0 and (n % 64 == 0 or o + n64 <= nxt), f"{name}: flag granularity"
            self.flags[o // 64:(o + n64) // 64] = 1 | (0 if is_no_decay(name, len(self.logical[name])) else 2)
        self.flags_version += 1
        # shadows that only trainable blocks need: drop those of blocks that are frozen now (one bf16 + one e4m3 copy of a block's weights
        # each), build the missing ones of blocks that train now
        for key in [k for k in self.wt if isinstance(k, tuple) and k[0] < self.first_trainable]:
            del self.wt[key]
        for key in [k for k in self.wt8 if k[0] < self.first_trainable]:
            del self.wt8[key]
        self.sync_transposed()
        if self.fp8_forward and self.fp8_dgrad:
            self.sync_fp8(range(self.first_trainable, self.cfg.layers))

    def _nonblock_trains(self, name):
        """Does a tensor outside the blocks (stem, final norm, head) train?  EVA02: only without --lock-image (set_trainable_all)."""
        return self.train_all

    def _pos_trains(self):
        return self.train_all

    def trainable_names(self):
        if self.train_all:
            return self.public_names()
        return [n for n in self.public_names()
                if (self.block_index(n) >= self.first_trainable if self.block_index(n) is not None else self._nonblock_trains(n))]

    def bucket_range(self, key):
        """Flat [begin, end) of a gradient bucket: a block index, "head" (final norm + head) or "stem" (cls_token, pos_embed, patch embedding)."""
        return self.head_range if key == "head" else self.stem_range if key == "stem" else self.block_ranges[key]

    # ------------------------------------------------------------------------------------------ tables
    def rope_tables(self, grid: int):
        """cos/sin [grid*grid, 64] (rope.py:118-142,179-214): 16 frequencies theta^(-2i/32), positions
        arange(grid)/grid*pt_seq_len, each repeated twice, row block then column block."""
        key = ("rope", grid)
        if key not in self._tables:
            half = self.cfg.head_width // 2
            freqs = 1.0 / (10000.0 ** (torch.arange(0, half, 2)[: half // 2].float() / half))
            t = torch.arange(grid).float() / grid * self.cfg.pt_hw_seq_len
            ang = (t[:, None] * freqs[None, :]).repeat_interleave(2, dim=-1)
            full = torch.cat([ang[:, None, :].expand(grid, grid, half), ang[None, :, :].expand(grid, grid, half)], dim=-1)
            full = full.reshape(grid * grid, self.cfg.head_width)
            self._tables[key] = (full.cos().contiguous().to(self.device), full.sin().contiguous().to(self.device))
        return self._tables[key]

    def pos_for(self, grid: int):
        """pos_embed [N, C] fp32, bicubic-rescaled for a non-native grid (eva_vit_model.py:631-643).  One-time
        host-side table preparation per grid size, cached."""
        if grid not in self._pos_cache:
            pe = self.p[self.prefix + "pos_embed"][0]
            if grid != self.cfg.grid:
                C = pe.shape[1]
                pe2 = pe[1:].T.contiguous().view(1, C, self.cfg.grid, self.cfg.grid)
                pe2 = F.interpolate(pe2, (grid, grid), mode="bicubic", align_corners=False).view(C, grid * grid)
                pe = torch.cat([pe[:1], pe2.T], dim=0)
            self._pos_cache[grid] = pe.contiguous()
        return self._pos_cache[grid]

    # ------------------------------------------------------------------------------------------ forward pieces
    def _stem(self, images, keep=None):
        ops, cfg, P = self.ops, self.cfg, self.prefix
        B, _, S, _ = images.shape
        p, C = cfg.patch_size, cfg.width
        g = S // p
        N, Kpe = g * g + 1, self.Kpe                       # contraction zero-padded to a multiple of 64 (3*14*14 -> 640)
        A = ops.empty((B * g * g, Kpe), BF16)
        ops.im2row(images.contiguous(), A, p)              # writes the zero padding columns too
        x = ops.empty((B, N, C), F32)
        pos = self.pos_for(g)
        ops.gemm_nt(A, self.storage_of(self.shadow, P + "patch_embed.proj.weight"), x.view(B * N, C),
                    bias=self.p[P + "patch_embed.proj.bias"], extra=pos, epi=EPI_PATCH_F32, group=g * g)
        ops.cls_row(x, self.p[P + "cls_token"].view(C), pos)
        if keep is not None:
            keep["patches"] = A                  # operand of the patch-embedding weight gradient
        return x, g

    def _qkv_w(self, b):
        C = self.cfg.width
        o = self.offsets[b + "attn.q_proj.weight"][0]
        ob = self.offsets[b + "attn.q_bias"][0]
        return self.shadow[o:o + 3 * C * C].view(3 * C, C), self.master[ob:ob + 3 * C]

    def _w12(self, b):
        C, Hd = self.cfg.width, self.Hp
        o = self.offsets[b + "mlp.w1.weight"][0]
        ob = self.offsets[b + "mlp.w1.bias"][0]
        return self.shadow[o:o + 2 * Hd * C].view(2 * Hd, C), self.master[ob:ob + 2 * Hd]

    def _block_fwd(self, i, x, B, N, cos, sin, with_attn=True, save=None, inplace=True):
        """x: fp32 [B*N, C].  Returns the block output (x itself when inplace)."""
        ops, cfg = self.ops, self.cfg
        C, Hd, Hl, H, eps = cfg.width, self.Hp, cfg.hidden, cfg.heads, cfg.ln_eps      # Hd: padded storage width, Hl: logical
        padded = Hd != Hl
        b = f"{self.prefix}blocks.{i}."
        M = B * N
        keep = save is not None
        st = (lambda: (ops.empty((M,), F32), ops.empty((M,), F32))) if keep else (lambda: (None, None))

        ln1 = ops.empty((M, C), BF16)
        m1, r1 = st()
        q1 = self._ln(x, self.p[b + "norm1.weight"], self.p[b + "norm1.bias"], ln1, m1, r1, eps)
        wqkv, bqkv = self._qkv_w(b)
        qkv = lse = None
        if with_attn and not keep and inplace and self.fold_sub_ln:
            qkv = ops.empty((M, 3 * C), BF16)
            ops.gemm_nt(ln1, wqkv, qkv, bias=bqkv, epi=EPI_BF16)
            att = ops.empty((M, C), BF16)
            part = ops.empty((H, M, 2), F32)
            ops.attn_fwd_stats(qkv, cos, sin, att, None, part, B, N, H, cfg.head_width ** -0.5)
            return self._block_post_folded(i, b, x, att, part, M)
        if with_attn:
            qkv = ops.empty((M, 3 * C), BF16)
            self._linear(i, "qkv", ln1, wqkv, qkv, bqkv, xq=q1)
            att = ops.empty((M, C), BF16)
            lse = ops.empty((B * H, N), F32) if keep else None
            ops.attn_fwd(qkv, cos, sin, att, lse, B, N, H, cfg.head_width ** -0.5)
        else:
            att = ops.empty((M, C), BF16)      # v only: every token "attends" to itself (proj_without_attn)
            self._linear(i, "qkv", ln1, wqkv[2 * C:], att, bqkv[2 * C:], rows=(2 * C, 3 * C), xq=q1)
        x2 = self._block_post(i, b, x, att, M, st, save, inplace)
        if keep:
            save.update(x0=x, ln1=ln1, st1=(m1, r1), qkv=qkv, lse=lse, att=att, with_attn=with_attn)
        return x2

    def _block_post(self, i, b, x, att, M, st, save, inplace):
        """Everything after the attention core: inner_attn_ln -> proj (+x) -> norm2 -> SwiGLU -> ffn_ln -> w3 (+x1)."""
        ops, cfg = self.ops, self.cfg
        C, Hd, Hl, eps = cfg.width, self.Hp, cfg.hidden, cfg.ln_eps
        padded = Hd != Hl
        keep = save is not None
        iln = ops.empty((M, C), BF16)
        m2, r2 = st()
        q2 = self._ln(att, self.p[b + "attn.inner_attn_ln.weight"], self.p[b + "attn.inner_attn_ln.bias"], iln, m2, r2, eps)
        x1 = x if inplace else ops.empty((M, C), F32)
        self._linear(i, "proj", iln, self.w[b + "attn.proj.weight"], x1, self.p[b + "attn.proj.bias"], extra=x, epi=EPI_RESID_F32, xq=q2)

        ln2 = ops.empty((M, C), BF16)
        m3, r3 = st()
        q3 = self._ln(x1, self.p[b + "norm2.weight"], self.p[b + "norm2.bias"], ln2, m3, r3, eps)
        w12, b12 = self._w12(b)
        hid = ops.empty((M, Hd), BF16)
        x12 = None
        if keep or self.fp8_forward:
            x12 = ops.empty((M, 2 * Hd), BF16)
            # (a W1|W2 epilogue that stores x1 | x2 AND silu(x1) * x2 was built and measured in round 4: 129.8 us against 112.3 us for GEMM +
            # cs_swiglu_fwd at 12 608 rows -- the un-overlapped epilogue costs more than the HBM-rate elementwise pass it replaces)
            self._linear(i, "w12", ln2, w12, x12, b12, xq=q3)
            ops.swiglu_fwd(x12, hid)
        else:
            ops.gemm_nt(ln2, w12, hid, bias=b12, epi=EPI_SWIGLU_BF16, group=Hd)
        fln = (ops.zeros if padded else ops.empty)((M, Hd), BF16)     # padding columns feed the W3 GEMM: must be exact zeros
        m4, r4 = st()
        q4 = self._ln(hid[:, :Hl], self.p[b + "mlp.ffn_ln.weight"], self.p[b + "mlp.ffn_ln.bias"], fln[:, :Hl], m4, r4, eps)
        x2 = x1 if inplace else ops.empty((M, C), F32)
        self._linear(i, "w3", fln, self.storage_of(self.shadow, b + "mlp.w3.weight"), x2, self.p[b + "mlp.w3.bias"], extra=x1,
                     epi=EPI_RESID_F32, xq=q4)
        if keep:
            save.update(iln=iln, st2=(m2, r2), x1=x1, ln2=ln2, st3=(m3, r3), x12=x12, hid=hid, fln=fln, st4=(m4, r4))
        return x2

    def _block_post_folded(self, i, b, x, att, att_part, M):
        """_block_post for a frozen tower with both sub-LayerNorms folded into proj / w3 (in place on x)."""
        ops, cfg = self.ops, self.cfg
        C, Hd, Hl, eps = cfg.width, self.Hp, cfg.hidden, cfg.ln_eps
        f = self.fold[i]
        mean, rstd = ops.empty((M,), F32), ops.empty((M,), F32)
        ops.ln_stats_finalize(att_part, 64, C, mean, rstd, eps)
        Wp, cp, dp = f["proj"]
        ops.gemm_nt_ln(att, Wp, x, bias=dp, extra=x, ln_mean=mean, ln_rstd=rstd, ln_colsum=cp, epi=EPI_RESID_LN_F32)

        ln2 = ops.empty((M, C), BF16)
        ops.layernorm_fwd(x, self.p[b + "norm2.weight"], self.p[b + "norm2.bias"], ln2, None, None, eps)
        w12, b12 = self._w12(b)
        hid = ops.empty((M, Hd), BF16)
        part = ops.empty((4 * ((Hd + 127) // 128), M, 2), F32)
        ops.gemm_nt_ln(ln2, w12, hid, bias=b12, stats_part=part, epi=EPI_SWIGLU_BF16, group=Hd)
        ops.ln_stats_finalize(part, 32, Hl, mean, rstd, eps)
        W3, c3, d3 = f["w3"]
        ops.gemm_nt_ln(hid, W3, x, bias=d3, extra=x, ln_mean=mean, ln_rstd=rstd, ln_colsum=c3, epi=EPI_RESID_LN_F32)
        return x

    def _teacher_block_folded(self, i, x, xb, st, B, N, cos, sin, emit_next, lo=None):
        """One frozen-tower block with all four LayerNorms folded into the GEMMs (in place on x).  xb / st = bf16 copy and (mean, rstd)
        of x for norm1 as left by the previous block's w3 GEMM, or None (first block: plain norm1 kernel).  Returns (xb, st) for the
        next block when emit_next.  With lo (int16 [M, C]) the stream lives in the planes (xb, lo) between the first residual GEMM of the
        tower, which reads fp32 x, and the last one (emit_next False), which writes fp32 x again (cs_gemm_nt_ln_split)."""
        ops, cfg = self.ops, self.cfg
        C, Hd, Hl, H, eps = cfg.width, self.Hp, cfg.hidden, cfg.heads, cfg.ln_eps
        b = f"{self.prefix}blocks.{i}."
        M = B * N
        f = self.fold[i]
        qkv = ops.empty((M, 3 * C), BF16)
        first = xb is None
        if first:
            ln1 = ops.empty((M, C), BF16)
            ops.layernorm_fwd(x, self.p[b + "norm1.weight"], self.p[b + "norm1.bias"], ln1, None, None, eps)
            wqkv, bqkv = self._qkv_w(b)
            ops.gemm_nt(ln1, wqkv, qkv, bias=bqkv, epi=EPI_BF16)
        else:
            Wq, cq, dq = f["qkv"]
            ops.gemm_nt_ln(xb, Wq, qkv, bias=dq, ln_mean=st[0], ln_rstd=st[1], ln_colsum=cq, epi=EPI_BF16)
        att = ops.empty((M, C), BF16)
        part_a = ops.empty((H, M, 2), F32)
        ops.attn_fwd_stats(qkv, cos, sin, att, None, part_a, B, N, H, cfg.head_width ** -0.5)
        mean, rstd = ops.empty((M,), F32), ops.empty((M,), F32)
        ops.ln_stats_finalize(part_a, 64, C, mean, rstd, eps)
        Wp, cp, dp = f["proj"]
        part_x = ops.empty(((C + 63) // 64, M, 2), F32)
        xb2 = xb if (lo is not None and not first) else ops.empty((M, C), BF16)
        if lo is not None:
            ops.gemm_nt_ln_split(att, Wp, xb2, lo, dp, mean, rstd, cp, x_in=x if first else None, stats_part=part_x)
        else:
            ops.gemm_nt_ln(att, Wp, x, bias=dp, extra=x, ln_mean=mean, ln_rstd=rstd, ln_colsum=cp, stats_part=part_x, xb_out=xb2,
                           epi=EPI_RESID_LN_F32)
        mean2, rstd2 = ops.empty((M,), F32), ops.empty((M,), F32)
        ops.ln_stats_finalize(part_x, 64, C, mean2, rstd2, eps)
        W12, c12, d12 = f["w12"]
        hid = ops.empty((M, Hd), BF16)
        part_h = ops.empty((4 * ((Hd + 127) // 128), M, 2), F32)
        ops.gemm_nt_ln(xb2, W12, hid, bias=d12, ln_mean=mean2, ln_rstd=rstd2, ln_colsum=c12, stats_part=part_h, epi=EPI_SWIGLU_BF16,
                       group=Hd)
        ops.ln_stats_finalize(part_h, 32, Hl, mean, rstd, eps)
        W3, c3, d3 = f["w3"]
        if lo is not None:
            if not emit_next:
                ops.gemm_nt_ln_split(hid, W3, xb2, lo, d3, mean, rstd, c3, x_out=x)
                return None, None
            ops.gemm_nt_ln_split(hid, W3, xb2, lo, d3, mean, rstd, c3, stats_part=part_x)
        else:
            if not emit_next:
                ops.gemm_nt_ln(hid, W3, x, bias=d3, extra=x, ln_mean=mean, ln_rstd=rstd, ln_colsum=c3, epi=EPI_RESID_LN_F32)
                return None, None
            ops.gemm_nt_ln(hid, W3, x, bias=d3, extra=x, ln_mean=mean, ln_rstd=rstd, ln_colsum=c3, stats_part=part_x, xb_out=xb2,
                           epi=EPI_RESID_LN_F32)
        ops.ln_stats_finalize(part_x, 64, C, mean2, rstd2, eps)
        return xb2, (mean2, rstd2)

    @staticmethod
    def _join_planes(hi, lo):
        """fp32 values of a split stream (cs_gemm_nt_ln_split: hi | lo = the halves of bits(x) + 0x8000); used on the B CLS rows only."""
        y = (hi.contiguous().view(torch.int16).to(torch.int32) << 16) | (lo.to(torch.int32) & 0xFFFF)
        return (y - 0x8000).view(torch.float32)

    def _block_fwd_cls(self, i, x, B, N, cos, sin, xb=None, st=None, lo=None):
        """Last teacher block restricted to what encode_image() consumes: the CLS row.  x fp32 [B*N, C] -> fp32 [B, C].
        forward_features() returns x[:, 0] after the final norm (eva_vit_model.py:505-519), so only the CLS *query* of the
        last block is live; keys and values still come from every token.  Row-for-row the same arithmetic as _block_fwd.
        With xb / st (/ lo) -- the bf16 operand view, norm1 statistics (and low plane) the previous folded block left -- norm1 is folded into
        the K|V GEMM like in every other block of the tower (no LayerNorm pass over the 403 456-row stream, the stream never returns to
        fp32) and only the B CLS rows are rebuilt in fp32 for the query, the residual adds and the MLP."""
        ops, cfg = self.ops, self.cfg
        C, H, eps = cfg.width, cfg.heads, cfg.ln_eps
        b = f"{self.prefix}blocks.{i}."
        M = B * N
        wqkv, bqkv = self._qkv_w(b)
        kv = ops.empty((M, 2 * C), BF16)
        q = ops.empty((B, C), BF16)
        if xb is not None:
            Wq, cq, dq = self.fold[i]["qkv"]
            ops.gemm_nt_ln(xb, Wq[C:], kv, bias=dq[C:], ln_mean=st[0], ln_rstd=st[1], ln_colsum=cq[C:], epi=EPI_BF16)
            xc = (self._join_planes(xb.view(B, N, C)[:, 0, :], lo.view(B, N, C)[:, 0, :]) if lo is not None
                  else x.view(B, N, C)[:, 0, :].contiguous())
            ln1c = ops.empty((B, C), BF16)
            ops.layernorm_fwd(xc, self.p[b + "norm1.weight"], self.p[b + "norm1.bias"], ln1c, None, None, eps)
            ops.gemm_nt(ln1c, wqkv[:C], q, bias=bqkv[:C], epi=EPI_BF16)
        else:
            ln1 = ops.empty((M, C), BF16)
            ops.layernorm_fwd(x, self.p[b + "norm1.weight"], self.p[b + "norm1.bias"], ln1, None, None, eps)
            ops.gemm_nt(ln1, wqkv[C:], kv, bias=bqkv[C:], epi=EPI_BF16)
            ops.gemm_nt(ln1.view(B, N, C)[:, 0, :], wqkv[:C], q, bias=bqkv[:C], epi=EPI_BF16)
            xc = x.view(B, N, C)[:, 0, :].contiguous()
        att = ops.empty((B, C), BF16)
        ops.attn_cls_fwd(q, kv, cos, sin, att, B, N, H, cfg.head_width ** -0.5)
        return self._block_post(i, b, xc, att, B, lambda: (None, None), None, True)

    # ------------------------------------------------------------------------------------------ teacher
    def fold_probe(self, crops: int = 16, kind: str = "white"):
        """The crops the guard is calibrated on, seeded (numpy PCG64, version-stable) at the tower's native size -- the same on every rank, in
        every run and whatever the first batch holds, so that all ranks of a data-parallel job choose the same teacher schedule and two runs of
        one checkpoint produce the same distillation targets.  (The statistic is a property of the weights -- massive-activation channels,
        bias-driven row means -- far more than of the pixels; oracle/stress_weights.py builds it from them.)  Two kinds (white
        noise alone can under-estimate a common-mode row mean that real crops excite): "white" = N(0, 1) pixels; "natural" = what
        normalised photographs look like to a patch embedding -- a 1/f amplitude spectrum (smooth regions, few edges) around a per-image,
        per-channel mean of N(0, 1), i.e. crops that are mostly one colour.  The guard takes the larger statistic of the two."""
        import numpy as np
        S = self.cfg.image_size
        g = np.random.Generator(np.random.PCG64(20250927 if kind == "white" else 20251001))
        x = g.standard_normal((crops, 3, S, S), dtype=np.float32)
        if kind == "natural":
            f = np.hypot(np.fft.fftfreq(S)[:, None], np.fft.rfftfreq(S)[None, :]).astype(np.float32)
            f[0, 0] = 1.0
            x = np.fft.irfft2(np.fft.rfft2(x) / f, s=(S, S)).astype(np.float32)
            x -= x.mean(axis=(2, 3), keepdims=True)
            x /= x.std(axis=(2, 3), keepdims=True) + 1e-6
            x = 0.5 * x + g.standard_normal((crops, 3, 1, 1), dtype=np.float32)
        elif kind != "white":
            raise ValueError(kind)
        return torch.from_numpy(np.ascontiguousarray(x)).to(self.device)

    def block_fold_statistic(self, images=None, crops: int = 16):
        """max over blocks of mean over rows of |row mean| / row sigma of the residual stream entering the block, on `images[:crops]`
        (default: the larger of the two fold_probe() kinds) through the plain block schedule.  One-time calibration after a weight load (a
        few torch reductions and one host read-back per probe, not part of the step)."""
        if images is None:
            return max(self.block_fold_statistic(self.fold_probe(crops, kind), crops) for kind in ("white", "natural"))
        with torch.no_grad():
            img = images[:crops]
            B = img.shape[0]
            x, g = self._stem(img)
            N = g * g + 1
            cos, sin = self.rope_tables(g)
            xf = x.view(B * N, self.cfg.width)
            worst = xf.new_zeros(())
            for i in range(self.cfg.layers):
                worst = torch.maximum(worst, (xf.mean(-1).abs() / xf.std(-1).clamp_min(1e-30)).mean())
                if i + 1 < self.cfg.layers:
                    self._block_fwd(i, xf, B, N, cos, sin, True, None, True)
            return float(worst)

    @staticmethod
    def _fold_group_active() -> bool:
        import torch.distributed as dist
        # (CLIPSELF_FORCE_DIST=1: the one-rank rehearsal of the N-rank path takes the collective too -- the only way to run this RCCL call on a one-GPU box)
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("CLIPSELF_FORCE_DIST") == "1")

    def calibrate_block_folds(self, collective: bool = True) -> bool:
        """Measure the guard's statistic of the current weights on the seeded probes and decide whether norm1 / norm2 are folded.  In a process
        group (collective=True) every rank takes the MAX over ranks -- one scalar all-reduce --, so the ranks cannot disagree even if their
        devices rounded differently.  That makes this a COLLECTIVE call: every rank must reach it at the same point of the program --
        training.main does right after a frozen tower's weights are in place (FrozenDataParallel.__init__, and after the evaluation model's
        load_state_dict), never from inside a forward pass that some rank might skip (an empty evaluation shard used to hang the
        job there).  An error of the collective is an error of the job and propagates.  The statistic and the decision are logged once per
        calibration."""
        import logging
        ratio = self.block_fold_statistic()
        agreed = ""
        if collective and self._fold_group_active():
            import torch.distributed as dist
            t = torch.tensor([ratio], dtype=torch.float64, device=self.device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ratio = float(t[0])
            agreed = f", MAX over {dist.get_world_size()} ranks"
        self.block_fold_ratio = ratio
        folded = ratio <= self.block_fold_limit
        near = abs(ratio - self.block_fold_limit) <= 0.1 * self.block_fold_limit
        logging.log(logging.WARNING if (near or not folded) else logging.INFO,
                    "frozen tower: mean |row mean| / row sigma of the residual stream = %.3f on the seeded probes%s (limit %.1f%s) -- norm1 / norm2 %s",
                    ratio, agreed, self.block_fold_limit, ", within 10 % of it" if near else "",
                    "folded into the q|k|v and W1|W2 GEMMs" if folded else
                    "stay LayerNorm kernels (the folded form would lose precision on these weights)")
        return folded

    def block_folds_active(self, images=None) -> bool:
        """Whether encode_image() folds norm1 / norm2 into the q|k|v and W1|W2 GEMMs: the switch, and -- with the guard armed -- the
        calibration of the current weights (calibrate_block_folds).  A caller with data in hand (`images` only says that the tower is about
        to run; its content does not enter the decision) that finds the weights uncalibrated calibrates them on the spot WITHOUT the
        collective: the probes are seeded, so ranks holding the same weights compute the same statistic up to device rounding; a job that
        wants the agreed value calls calibrate_block_folds() at a point every rank reaches (a warning says so in a multi-rank group)."""
        if not self.fold_block_ln:
            return False
        if not self.block_fold_guard:
            return True
        if self.block_fold_ratio is None:
            if images is None:
                return True
            if self._fold_group_active():
                import logging
                logging.warning("frozen tower: fold guard calibrated lazily inside a forward pass of a multi-rank job -- local value, no rank "
                                "agreement; call engine.calibrate_block_folds() on every rank after the weight load")
            self.calibrate_block_folds(collective=False)
        return self.block_fold_ratio <= self.block_fold_limit

    def _rccl_window_step(self, i, k0):
        """rccl_window = (leading blocks, CUs): the persistent GEMMs of the first chunk's leading blocks leave those CUs to RCCL's kernels
        (a prefetched pass runs them beside the student's gradient buckets); shared by every engine's encode_image()."""
        win, cus = self.rccl_window
        if win and k0 == 0 and i in (0, win) and hasattr(self.ops, "reserve_compute_units"):
            self.ops.reserve_compute_units(cus if i < win else 0)

    def _rccl_window_close(self, k0):
        """... and whatever happens inside the window (an exception included), the reservation does not outlive the pass."""
        if self.rccl_window[0] and k0 == 0 and hasattr(self.ops, "reserve_compute_units"):
            self.ops.reserve_compute_units(0)

    def encode_image(self, images, chunk: int = 256):
        """Frozen-teacher path: full ViT, final LN on the CLS row, head.  [K,3,S,S] -> fp32 [K,E].
        Activation-free: crops are streamed in chunks, blocks update the residual stream in place."""
        ops, cfg, P = self.ops, self.cfg, self.prefix
        K = images.shape[0]
        out = ops.empty((K, cfg.embed_dim), F32)
        fold_blocks = self.fold_sub_ln and self.block_folds_active(images)
        for k0 in range(0, K, chunk):
            img = images[k0:k0 + chunk]
            B = img.shape[0]
            x, g = self._stem(img)
            N = g * g + 1
            cos, sin = self.rope_tables(g)
            xf = x.view(B * N, cfg.width)
            last = cfg.layers - 1 if self.cls_only_last_block else cfg.layers
            xb = st = None
            folded = self.fold_sub_ln and fold_blocks
            lo = ops.empty((B * N, cfg.width), torch.int16) if folded and self.split_stream and last > 0 else None
            cls_folded = folded and last < cfg.layers and last > 0          # the CLS-only block takes the planes + statistics as they are
            try:
                for i in range(last):
                    self._rccl_window_step(i, k0)
                    if folded:
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
            ops.layernorm_fwd(xc, self.p[P + "norm.weight"], self.p[P + "norm.bias"], cls, None, None, cfg.ln_eps)
            ops.gemm_nt(cls, self.w[P + "head.weight"], out[k0:k0 + B], bias=self.p[P + "head.bias"], epi=EPI_F32)
        return out

    # ------------------------------------------------------------------------------------------ student
    def encode_dense(self, images, need_grad: bool = False):
        """Dense path -> L2-normalised token map fp32 [B, N, E] (row 0 of each image is the unused CLS slot).
        With need_grad the activations of the trainable blocks are kept for backward_dense()."""
        ops, cfg, P = self.ops, self.cfg, self.prefix
        B = images.shape[0]
        stem_keep = {} if (need_grad and self.train_all) else None
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
        ops.layernorm_fwd(xf, self.p[P + "norm.weight"], self.p[P + "norm.bias"], lnf, mean, rstd, cfg.ln_eps)
        feats = ops.empty((M, E), F32)
        ops.gemm_nt(lnf, self.w[P + "head.weight"], feats, bias=self.p[P + "head.bias"], epi=EPI_F32)
        dense = ops.empty((M, E), F32)
        inv = ops.empty((M,), F32)
        ops.l2norm_fwd(feats, dense, inv)
        if need_grad:
            self._ctx = dict(B=B, N=N, g=g, saves=saves, xL=xf, stf=(mean, rstd), dense=dense, inv=inv, cos=cos, sin=sin,
                             lnf=lnf if self.train_all else None, patches=stem_keep["patches"] if stem_keep is not None else None)
        return dense.view(B, N, E), g

    def roi_pool(self, dense, rois, g):
        """rois [K,5] = (image index, x0,y0,x1,y1 normalised to [0,1]) -> fp32 [K,E]."""
        pooled = self.ops.empty((rois.shape[0], dense.shape[2]), F32)
        if rois.shape[0]:
            self.ops.roialign_fwd(dense, rois, pooled, g, g, 1)
        return pooled

    # ------------------------------------------------------------------------------------------ backward
    def zero_grad(self):
        self.grad.zero_()

    def _wgrad(self, dY, X, dW):
        """dW[N,K] += dY^T X, dY [M,N] and X [M,K] token-major bf16.  cs_gemm_wgrad_tn contracts them as they are (transposing LDS
        reads); shapes it does not cover (N or K not a multiple of 8) go through explicit transposes + the NT split-K kernel."""
        ops = self.ops
        M, N = dY.shape
        K = X.shape[1]
        need = ops.gemm_wgrad_tn_workspace(N, K, M) if self.wgrad_tn else 0
        if need:
            if self._wgrad_ws is None or self._wgrad_ws.numel() < need:
                self._wgrad_ws = ops.empty((need,), torch.uint8)
            ops.gemm_wgrad_tn(dY, X, dW, self._wgrad_ws)
            return
        Xt = self._transposed(X)
        Mp = Xt.shape[1]
        dYt = ops.empty((N, Mp), BF16)
        ops.transpose_bf16(dY, dYt)
        need = ops.gemm_wgrad_workspace(N, Xt.shape[0], Mp)
        if self._wgrad_ws is None or self._wgrad_ws.numel() < need:
            self._wgrad_ws = ops.empty((need,), torch.uint8)
        ops.gemm_wgrad(dYt, Xt, dW, self._wgrad_ws)                      # split-K through partial buffers, then dW += sum

    def _transposed(self, X):
        M, K = X.shape
        Xt = self.ops.empty((K, _round_up(M, 64)), BF16)
        self.ops.transpose_bf16(X, Xt)
        return Xt

    def _block_bwd(self, i, s, g, gb, B, N, cos, sin, ws, next_bias=None, gq=None):
        """g: fp32 [M,C] gradient w.r.t. the block output; updated in place to the gradient w.r.t. its input.  gb: its bf16 copy, already
        summed into this block's w3 bias gradient by the LayerNorm backward that produced it (the final norm's, or norm1's of block i+1);
        on return gb is the copy of the new g and its column sums have gone to `next_bias` (block i-1's w3 bias gradient, or None).
        gq (fp8_dgrad): (e4m3 codes, row scales) of gb, written by the same LayerNorm backwards (cs_layernorm_bwd_q8)."""
        q8a = dict(q8=gq[0], q_scale=gq[1]) if gq is not None else {}
        ops, cfg = self.ops, self.cfg
        C, Hd, Hl, H = cfg.width, self.Hp, cfg.hidden, cfg.heads
        padded = Hd != Hl
        b = f"{self.prefix}blocks.{i}."
        M = B * N
        G = self.g
        # ---- MLP: x2 = x1 + w3(ffn_ln(silu(x1')*x2')) ------------------------------------------
        self._wgrad(gb, s["fln"], self.storage_of(self.grad, b + "mlp.w3.weight"))
        d_fln = ops.empty((M, Hd), BF16)
        self._dgrad(i, "w3", gb, d_fln, q=gq)                                               # [M,C] . W3[C,Hd]
        d_hid = (ops.zeros if padded else ops.empty)((M, Hd), BF16)
        ops.layernorm_bwd(d_fln[:, :Hl], s["hid"][:, :Hl], self.p[b + "mlp.ffn_ln.weight"], *s["st4"], d_hid[:, :Hl], DX_BF16,
                          G[b + "mlp.ffn_ln.weight"], G[b + "mlp.ffn_ln.bias"], True, ws[0])
        d_x12 = ops.empty((M, 2 * Hd), BF16)
        ob = self.offsets[b + "mlp.w1.bias"][0]
        q12 = None
        if self.fp8_dgrad and Hd <= 4096:
            q12 = (ops.empty((M, _round_up(2 * Hd, 128)), torch.uint8), ops.empty((M,), F32))
            ops.swiglu_bwd(d_hid, s["x12"], d_x12, q8=q12[0], q_scale=q12[1])
            ops.colsum_bf16(d_x12, self.grad[ob:ob + 2 * Hd], ws[1])
        elif self.fused_swiglu_colsum:
            ops.swiglu_bwd_colsum(d_hid, s["x12"], d_x12, self.grad[ob:ob + 2 * Hd], ws[1])      # d x1|x2 and the w1 | w2 bias gradients in one pass
        else:
            ops.swiglu_bwd(d_hid, s["x12"], d_x12)
            ops.colsum_bf16(d_x12, self.grad[ob:ob + 2 * Hd], ws[1])
        ow = self.offsets[b + "mlp.w1.weight"][0]
        self._wgrad(d_x12, s["ln2"], self.grad[ow:ow + 2 * Hd * C].view(2 * Hd, C))
        d_ln2 = ops.empty((M, C), BF16)
        self._dgrad(i, "w12", d_x12, d_ln2, q=q12)                                          # [M,2Hd] . W12[2Hd,C]
        # norm2's backward adds into the stream gradient and hands back its bf16 copy + column sums (= the proj bias gradient)
        ops.layernorm_bwd(d_ln2, s["x1"], self.p[b + "norm2.weight"], *s["st3"], g, DX_F32_ACCUM,
                          G[b + "norm2.weight"], G[b + "norm2.bias"], True, ws[0], dx_copy=gb, copy_colsum=G[b + "attn.proj.bias"], **q8a)
        # ---- attention branch: x1 = x0 + proj(inner_ln(att)) -------------------------------------
        self._wgrad(gb, s["iln"], G[b + "attn.proj.weight"])
        d_iln = ops.empty((M, C), BF16)
        self._dgrad(i, "proj", gb, d_iln, q=gq)
        d_att = ops.empty((M, C), BF16)
        ops.layernorm_bwd(d_iln, s["att"], self.p[b + "attn.inner_attn_ln.weight"], *s["st2"], d_att, DX_BF16,
                          G[b + "attn.inner_attn_ln.weight"], G[b + "attn.inner_attn_ln.bias"], True, ws[0])
        oq = self.offsets[b + "attn.q_proj.weight"][0]
        d_ln1 = ops.empty((M, C), BF16)
        if s["with_attn"]:
            d_qkv = ops.empty((M, 3 * C), BF16)
            ops.attn_bwd(s["qkv"], s["att"], d_att, s["lse"], cos, sin, d_qkv, ws[0], B, N, H, cfg.head_width ** -0.5)
            # one pass over d_q|d_k|d_v into the adjacent [q_bias; (k: no bias, eva_vit_model.py:178); v_bias] gradient slots; the middle slot
            # belongs to no parameter and goes back to zero (the flat gradient's norm must be the parameters' gradient norm)
            obq = self.offsets[b + "attn.q_bias"][0]
            ops.colsum_bf16(d_qkv, self.grad[obq:obq + 3 * C], ws[1])
            self.grad[obq + C:obq + 2 * C].zero_()
            self._wgrad(d_qkv, s["ln1"], self.grad[oq:oq + 3 * C * C].view(3 * C, C))
            self._dgrad(i, "qkv", d_qkv, d_ln1)
        else:
            ops.colsum_bf16(d_att, G[b + "attn.v_bias"], ws[1])
            self._wgrad(d_att, s["ln1"], G[b + "attn.v_proj.weight"])
            self._dgrad(i, "qkv", d_att, d_ln1, cols=(2 * C, 3 * C))
        ops.layernorm_bwd(d_ln1, s["x0"], self.p[b + "norm1.weight"], *s["st1"], g, DX_F32_ACCUM,
                          G[b + "norm1.weight"], G[b + "norm1.bias"], True, ws[0], dx_copy=gb if next_bias is not None else None,
                          copy_colsum=next_bias, **(q8a if next_bias is not None else {}))

    def backward_dense(self, d_dense):
        """d_dense: fp32 [B, N, E] gradient w.r.t. the normalised token map (CLS rows zero).  Accumulates every
        trainable-block gradient into the flat grad buffer; fires grad_ready_hook(block) as blocks complete."""
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
        ops.gemm_nt(d_feats, self.wt["head"][:, :E], d_lnf, epi=EPI_BF16)                  # dgrad through the head
        g = ops.empty((M, C), F32)
        gb = ops.empty((M, C), BF16)                  # bf16 copy of g, written by the LayerNorm backwards that update g
        gq = (ops.empty((M, _round_up(C, 128)), torch.uint8), ops.empty((M,), F32)) if self.fp8_dgrad and C <= 3072 else None
        q8a = dict(q8=gq[0], q_scale=gq[1]) if gq is not None else {}
        ws_bytes = max(ops.layernorm_bwd_workspace(M, max(C, self.Hp)), ops.attn_bwd_workspace(B, N, cfg.heads))
        ws = (ops.empty((ws_bytes,), torch.uint8), ops.empty((max(ops.colsum_workspace(M, max(2 * self.Hp, 3 * C)), 4),), torch.uint8))
        L, first = cfg.layers, self.first_trainable
        w3_bias = lambda i: self.g[f"{P}blocks.{i}.mlp.w3.bias"] if i >= first else None
        if self.train_all:
            # head (eva_vit_model.py:617) and final norm (:616) train: bias = column sums, weight = dY^T . LN(x), LayerNorm gamma / beta.
            # The CLS rows of d_feats are exact zeros (the dense map drops them, :615), so they add nothing to any of the sums.
            ops.colsum_bf16(d_feats, self.g[P + "head.bias"], ws[1])
            self._wgrad(d_feats, c["lnf"], self.g[P + "head.weight"])
            ops.layernorm_bwd(d_lnf, c["xL"], self.p[P + "norm.weight"], *c["stf"], g, DX_F32_ASSIGN,
                              self.g[P + "norm.weight"], self.g[P + "norm.bias"], True, ws[0], dx_copy=gb, copy_colsum=w3_bias(L - 1), **q8a)
            if self.grad_ready_hook is not None:
                self.grad_ready_hook("head")
        else:                                                                                   # head and final norm frozen
            ops.layernorm_bwd(d_lnf, c["xL"], self.p[P + "norm.weight"], *c["stf"], g, DX_F32_ASSIGN, None, None, True, ws[0],
                              dx_copy=gb if first < L else None, copy_colsum=w3_bias(L - 1), **(q8a if first < L else {}))
        for i in range(L - 1, first - 1, -1):
            self._block_bwd(i, c["saves"].pop(i), g, gb, B, N, c["cos"], c["sin"], ws, next_bias=w3_bias(i - 1) if i > 0 else None, gq=gq)
            if self.grad_ready_hook is not None:
                self.grad_ready_hook(i)
        if self.train_all:
            self._stem_bwd(g, c["patches"], B, N, c["g"])
            if self.grad_ready_hook is not None:
                self.grad_ready_hook("stem")

    def _stem_bwd(self, g, patches, B, N, grid):
        """Gradients of the stem from g = d loss / d (stem output) fp32 [B*N, C]  (eva_vit_model.py:537-544: x = cat(cls, conv(img)) + pos):
        pos_embed <- sum over images (through the bicubic rescale for a non-native grid, :631-643), cls_token <- the CLS rows,
        patch_embed.proj <- bias = column sums of the patch rows, weight = dY^T . im2row(images).  Runs once per step on [B*N, C]
        tensors; the row bookkeeping (dropping the CLS rows, the sum over images) is plain tensor code, the contraction is the wgrad kernel."""
        ops, cfg, P = self.ops, self.cfg, self.prefix
        C = cfg.width
        g3 = g.view(B, N, C)
        d_pos = g3.sum(dim=0)                                                   # [N, C]
        self.g[P + "cls_token"].view(C).add_(d_pos[0])
        gpos = self.g[P + "pos_embed"][0]                                       # [native N, C]
        if grid == cfg.grid:
            gpos.add_(d_pos)
        else:
            gpos[0].add_(d_pos[0])
            with torch.enable_grad():
                pe = self.p[P + "pos_embed"].detach()[0, 1:].T.reshape(1, C, cfg.grid, cfg.grid).clone().requires_grad_(True)
                out = F.interpolate(pe, (grid, grid), mode="bicubic", align_corners=False)
                (d_pe,) = torch.autograd.grad(out, pe, d_pos[1:].T.reshape(1, C, grid, grid))
            gpos[1:].add_(d_pe.reshape(C, cfg.grid * cfg.grid).T)
        gp = g3[:, 1:, :].to(BF16).reshape(B * (N - 1), C)                      # patch rows, in the im2row matrix's row order
        ops.colsum_bf16(gp, self.g[P + "patch_embed.proj.bias"])          # (allocates its own row-block workspace: once per step)
        self._wgrad(gp, patches, self.storage_of(self.grad, P + "patch_embed.proj.weight"))

    def roi_pool_backward(self, d_pooled, rois, B, N, g):
        d_dense = self.ops.zeros((B, N, self.cfg.embed_dim), F32)
        if rois.shape[0]:
            self.ops.roialign_bwd(d_pooled.contiguous(), rois, d_dense, g, g, 1)
        return d_dense

    # ------------------------------------------------------------------------------------------ optimizer
    def adamw_step(self, step: int, lr: float, wd: float, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale: float = 1.0):
        """One flat AdamW launch over every trainable tensor (fp32 master + bf16 shadow refresh), then the W^T shadows."""
        self.ops.adamw_step(self.master, self.grad, self.exp_avg, self.exp_avg_sq, self.shadow, self.flags,
                            lr, beta1, beta2, eps, wd, step, grad_scale)
        self.sync_transposed()
        if self._pos_trains():
            self._pos_cache.clear()                # pos_embed moved: drop the rescaled copies of non-native grids
        if self.fp8_forward:
            self.sync_fp8(range(self.first_trainable, self.cfg.layers))
