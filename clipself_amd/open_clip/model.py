"""Model objects behind the reference's `open_clip` API for the EVA02 towers (and, SURVEY.md §8 N4, the OpenAI-CLIP ViT family:
`CLIP` / `ClipVisionTower` at the end of this file), backed by the HIP step engines.

Surface mirrored (reference: src/open_clip/eva_clip/model.py:272-346 `CustomCLIP`,
src/open_clip/eva_clip/eva_vit_model.py:396-711 `EVAVisionTransformer`):
  model.encode_image(x, normalize=False)                               -> [K, E]
  model.encode_dense(x, normalize=False, keep_shape=False)             -> [B, hw, E] | [B, E, h, w]
  model.encode_pseudo_boxes(x, list[Tensor[k_i,4]], normalize=False, extract_type='v2') -> [K, E]
  model.encode_masks(x, masks, normalize=True)                         -> [sum masks, E]
  model.lock_image_tower(unlocked_groups, freeze_bn_stats), .set_grad_checkpointing(), .logit_scale,
  model.visual.image_size / image_mean / image_std, .train() / .eval(), .state_dict() with the reference's keys.

Every `visual.*` parameter is an nn.Parameter *view* into the engine's flat fp32 master buffer, and its `.grad`
is a view into the flat grad buffer, so torch-side tools (state_dict, optimizers, checkpoints) see ordinary
parameters while the kernels see four contiguous arrays.  Gradients enter torch autograd through two
autograd.Functions (dense map, RoI pooling); everything between them is the engine's explicit schedule.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import TowerCfg
from ..engine import EvaEngine, F32
from ..engine_openai import ClipVitEngine

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)


def _default_ops():
    from ..hip import HipOps          # raises if the HIP library or the GPU is missing: no fallback
    return HipOps()


class _DenseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, tower, images):
        dense, grid = tower.engine.encode_dense(images, need_grad=True)
        ctx.tower = tower
        return dense

    @staticmethod
    def backward(ctx, d_dense):
        tower = ctx.tower
        tower._prepare_grads()
        tower.engine.backward_dense(d_dense.contiguous())
        tower._attach_grads()
        return None, None, None


class _RoiFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dense, tower, rois, grid):
        ctx.tower, ctx.rois, ctx.grid, ctx.shape = tower, rois, grid, dense.shape
        return tower.engine.roi_pool(dense.contiguous(), rois, grid)

    @staticmethod
    def backward(ctx, d_pooled):
        B, N, _ = ctx.shape
        return ctx.tower.engine.roi_pool_backward(d_pooled, ctx.rois, B, N, ctx.grid), None, None, None


class _Node(nn.Module):
    """Structural holder so that parameters get the reference's dotted names (visual.blocks.0.attn.q_proj.weight ...)."""

    def child(self, name):
        if name not in self._modules:
            self.add_module(name, _Node())
        return self._modules[name]

    def _apply(self, fn, recurse=True):
        return self


class EVAVisionTower(_Node):
    """`model.visual`: EVA02 ViT (RoPE + SwiGLU + sub-LN) executing on the HIP engine."""
    ENGINE = EvaEngine
    UNLOCKED_TRAINS_ALL = True                 # a fresh model trains its whole visual tower, like the reference's before lock() (both families)

    def __init__(self, cfg: TowerCfg, ops=None, trainable: bool = True, teacher_chunk: int = 2048):
        super().__init__()
        self.cfg = cfg
        self.image_size = cfg.image_size
        self.image_mean, self.image_std = OPENAI_DATASET_MEAN, OPENAI_DATASET_STD
        self.num_heads, self.embed_dim, self.num_classes = cfg.heads, cfg.width, cfg.embed_dim
        self.engine = self.ENGINE(cfg, ops if ops is not None else _default_ops(), trainable=trainable, prefix="visual.")
        self.teacher_chunk = teacher_chunk
        self._flat = {}                                   # full name -> nn.Parameter (view of the flat master)
        for full in self.engine.public_names():
            parts = full[len("visual."):].split(".")
            node = self
            for part in parts[:-1]:
                node = node.child(part)
            param = nn.Parameter(self.engine.p[full], requires_grad=trainable)
            node.register_parameter(parts[-1], param)
            self._flat[full] = param
        self._register_tables()
        self._anchor = torch.zeros((), device=self.engine.device, requires_grad=True)
        self.grad_checkpointing = False
        if trainable and hasattr(self.engine, "set_trainable_all") and self.UNLOCKED_TRAINS_ALL:
            self.engine.set_trainable_all()       # the reference's state before lock_image_tower(): the whole visual tower trains

    def _register_tables(self):
        # the reference registers the RoPE tables as buffers of the tower and (shared module) of every attention
        # (rope.py:138-139); kept for checkpoint-key compatibility
        cfg = self.cfg
        cos, sin = self.engine.rope_tables(cfg.grid)
        for node in [self.child("rope")] + [self.child("blocks").child(str(i)).child("attn").child("rope") for i in range(cfg.layers)]:
            node.register_buffer("freqs_cos", cos)
            node.register_buffer("freqs_sin", sin)

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self.engine.sync_shadow()
        return out

    # ---- training-state plumbing --------------------------------------------------------------------
    def lock(self, unlocked_groups=0, freeze_bn_stats=False):
        """eva_vit_model.py:500-516: freeze everything, then unfreeze the last `unlocked_groups` blocks."""
        self.engine.set_trainable_blocks(unlocked_groups)
        first = self.engine.first_trainable
        for full, p in self._flat.items():
            blk = self.engine.block_index(full)
            p.requires_grad = blk >= first if blk is not None else self.engine._nonblock_trains(full)

    def unlock(self):
        """Undo lock(): the whole tower trains again (the state a freshly built model is in, like the reference's)."""
        if not self.UNLOCKED_TRAINS_ALL:
            raise NotImplementedError("this tower family only differentiates its transformer blocks: lock it with unlocked_groups <= layers")
        self.engine.set_trainable_all()
        for p in self._flat.values():
            p.requires_grad = True

    def set_grad_checkpointing(self, enable=True):
        self.grad_checkpointing = enable       # activations are kept; 288 GB of HBM makes recompute pointless here

    def _trainable(self):
        return self.engine.trainable and any(p.requires_grad for p in self._flat.values())

    def _prepare_grads(self):
        """torch semantics: a parameter whose .grad is None starts from zero, otherwise gradients accumulate."""
        active = self._active_cpu
        if any(p.requires_grad and p.grad is None and active[self.engine.offsets[n][0] // 64] for n, p in self._flat.items()):
            self.engine.zero_grad()

    def _attach_grads(self):
        active = self._active_cpu
        for full, p in self._flat.items():
            # parameters the dense path never reaches keep grad=None, exactly like the reference
            if p.requires_grad and p.grad is None and active[self.engine.offsets[full][0] // 64]:
                p.grad = self.engine.g[full]

    @property
    def _active_cpu(self):
        # the flag bytes are rewritten in place by every lock() / unlock(): key on the engine's flag version, not on the buffer's identity
        # (lock(all blocks) and unlock() share first_trainable == 0 but differ in the stem / head flags)
        key = (id(self.engine.flags), self.engine.flags_version)
        if getattr(self, "_active_key", None) != key:
            self._active_cache = (self.engine.flags & 1).cpu().numpy()
            self._active_key = key
        return self._active_cache

    # ---- forward paths ---------------------------------------------------------------------------------
    def forward(self, x, return_all_features=False):
        if return_all_features:
            raise NotImplementedError("return_all_features is not on the CLIPSelf hot path")
        return self.engine.encode_image(x.to(self.engine.device), chunk=self.teacher_chunk)

    def _dense(self, x):
        x = x.to(self.engine.device)
        if torch.is_grad_enabled() and self._trainable():
            dense = _DenseFn.apply(self._anchor, self, x)
        else:
            dense, _ = self.engine.encode_dense(x, need_grad=False)
        return dense, x.shape[2] // self.cfg.patch_size

    def encode_dense(self, x, keep_shape=True):
        dense, g = self._dense(x)
        feats = dense[:, 1:]
        if keep_shape:
            return feats.reshape(x.shape[0], g, g, -1).permute(0, 3, 1, 2)
        return feats

    def extract_roi_features(self, x, normed_boxes, **kwargs):
        dense, g = self._dense(x)
        rois = boxes_to_rois(normed_boxes, self.engine.device)
        if dense.requires_grad:
            return _RoiFn.apply(dense, self, rois, g)
        return self.engine.roi_pool(dense, rois, g)

    def mask_pool(self, x, masks):
        """eva_vit_model.py:645-653 (evaluation-time helper, plain tensor math on the dense map)."""
        feature_map = self.encode_dense(x, keep_shape=False)
        counts = [len(m) for m in masks]
        m = torch.cat(masks).float().flatten(-2, -1).to(feature_map.device)
        fm = torch.repeat_interleave(feature_map, torch.tensor(counts, device=feature_map.device), dim=0)
        return (fm * m.unsqueeze(-1)).sum(1) / (m.sum(1, keepdim=True) + 1e-12)


def boxes_to_grid_masks(normed_boxes, grid_h: int, grid_w: int):
    """VisionTransformer._generate_masks_per_image (open_clip/transformer.py:636-646): box * (w, h, w, h), truncated towards zero, rows
    y0:y1 and columns x0:x1 of a [k, grid_h, grid_w] bool mask set (an empty slice leaves the mask empty)."""
    scaled = (normed_boxes.detach().to("cpu", F32)[:, :4] * torch.tensor([[grid_w, grid_h, grid_w, grid_h]], dtype=F32)).long().tolist()
    masks = torch.zeros(len(scaled), grid_h, grid_w, dtype=torch.bool)
    for i, (x0, y0, x1, y1) in enumerate(scaled):
        masks[i, y0:y1, x0:x1] = True
    return masks


def boxes_to_rois(normed_boxes, device):
    """list[Tensor[k_i, 4]] (x0,y0,x1,y1 in [0,1]) -> [K,5] with the image index in column 0 (the layout
    torchvision.roi_align builds from a box list)."""
    if isinstance(normed_boxes, torch.Tensor) and normed_boxes.dim() == 2 and normed_boxes.shape[1] == 5:
        return normed_boxes.to(device, F32).contiguous()
    rows = [torch.cat([torch.full((len(b), 1), float(i), device=b.device, dtype=F32), b.to(F32)[:, :4]], dim=1)
            for i, b in enumerate(normed_boxes)]
    out = torch.cat(rows) if rows else torch.zeros(0, 5)
    return out.to(device).contiguous()


class FrozenTextTower(nn.Module):
    """The reference constructs, freezes and checkpoints a TextTransformer that the distillation step never runs
    (eva_clip/model.py:284-288).  Only its state-dict keys/shapes matter; they are held here as
    frozen parameters so checkpoints round-trip."""

    def __init__(self, cfg: TowerCfg, device, mask_in_state_dict: bool = True):
        super().__init__()
        self.mask_in_state_dict = mask_in_state_dict
        W, L, E = cfg.text_width, cfg.text_layers, cfg.embed_dim
        shapes = {"positional_embedding": (cfg.text_context, W), "text_projection": (W, E),
                  "token_embedding.weight": (cfg.text_vocab, W), "ln_final.weight": (W,), "ln_final.bias": (W,)}
        for i in range(L):
            r = f"transformer.resblocks.{i}."
            shapes.update({r + "ln_1.weight": (W,), r + "ln_1.bias": (W,), r + "attn.in_proj_weight": (3 * W, W),
                           r + "attn.in_proj_bias": (3 * W,), r + "attn.out_proj.weight": (W, W), r + "attn.out_proj.bias": (W,),
                           r + "ln_2.weight": (W,), r + "ln_2.bias": (W,), r + "mlp.c_fc.weight": (4 * W, W),
                           r + "mlp.c_fc.bias": (4 * W,), r + "mlp.c_proj.weight": (W, 4 * W), r + "mlp.c_proj.bias": (W,)})
        self._names = {}
        for k, s in shapes.items():
            safe = k.replace(".", "__")
            self.register_parameter(safe, nn.Parameter(torch.zeros(s, device=device), requires_grad=False))
            self._names[safe] = k
        self.register_buffer("attn_mask", torch.full((cfg.text_context, cfg.text_context), float("-inf"), device=device).triu_(1),
                             persistent=False)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for safe, k in self._names.items():
            p = self._parameters[safe]
            destination[prefix + k] = p if keep_vars else p.detach()
        if self.mask_in_state_dict:
            destination[prefix + "attn_mask"] = self.attn_mask

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        with torch.no_grad():
            for safe, k in self._names.items():
                if prefix + k in state_dict:
                    self._parameters[safe].copy_(state_dict[prefix + k])
                else:
                    missing_keys.append(prefix + k)

    def forward(self, text):
        raise NotImplementedError("the text tower is not on the CLIPSelf hot path (never executed by the reference's training step)")


class CustomCLIP(nn.Module):
    def __init__(self, cfg: TowerCfg, ops=None, trainable: bool = True, with_text: bool = True):
        super().__init__()
        self.visual = EVAVisionTower(cfg, ops=ops, trainable=trainable)
        self.text = FrozenTextTower(cfg, self.visual.engine.device) if with_text else None
        self.embed_dim = cfg.embed_dim
        self.logit_scale = nn.Parameter(torch.ones([], device=self.visual.engine.device) * np.log(1 / 0.07))

    def train(self, mode=True):
        super().train(mode)
        if self.text is not None:
            self.text.train(False)
        return self

    def _apply(self, fn, recurse=True):
        return self            # tensors already live on the engine's device as views of flat buffers

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self.visual.engine.sync_shadow()
        return out

    def lock_image_tower(self, unlocked_groups=0, freeze_bn_stats=False, **kwargs):
        self.visual.lock(unlocked_groups=unlocked_groups, freeze_bn_stats=freeze_bn_stats)

    def set_grad_checkpointing(self, enable=True):
        self.visual.set_grad_checkpointing(enable)

    def no_weight_decay(self):
        return {"logit_scale"}

    def encode_image(self, image, normalize: bool = False):
        features = self.visual(image)
        return F.normalize(features, dim=-1) if normalize else features

    def encode_text(self, text, normalize: bool = False):
        raise NotImplementedError("text encoding is outside the CLIPSelf hot path")

    def encode_dense(self, image, normalize: bool = False, keep_shape=False):
        features = self.visual.encode_dense(image, keep_shape=keep_shape)
        if normalize:
            features = F.normalize(features, dim=1 if keep_shape else -1)
        return features

    def encode_pseudo_boxes(self, image, normed_boxes, normalize: bool = False, extract_type="v1"):
        features = self.visual.extract_roi_features(image, normed_boxes, extract_type=extract_type)
        if normalize:
            features = F.normalize(features, dim=-1)
        return features

    def encode_masks(self, image, masks, normalize=True, mask_attn=False):
        # open_clip/model.py:245-247: the OpenAI-CLIP family pools through extra query tokens when mask_attn is set; the EVA02 model takes the
        # argument and ignores it (eva_clip/model.py:342-346: always mask_pool) -- and so does this one
        if mask_attn and hasattr(self.visual, "mask_attn_pool"):
            mask_pooled = self.visual.mask_attn_pool(image, masks)
        else:
            mask_pooled = self.visual.mask_pool(image, masks)
        if normalize:
            mask_pooled = F.normalize(mask_pooled, dim=-1)
        return mask_pooled


# ------------------------------------------------------------------------------------------------------------------
# OpenAI-CLIP ViT family: reference src/open_clip/model.py:181-311 `CLIP`, transformer.py:318-734
# `VisionTransformer` -- same API surface as above on the ClipVitEngine schedule.
# ------------------------------------------------------------------------------------------------------------------
class ClipVisionTower(EVAVisionTower):
    """`model.visual` of the OpenAI-CLIP family: class/positional embeddings, ln_pre, fused-QKV blocks with a GELU MLP, ln_post, proj."""
    ENGINE = ClipVitEngine

    def __init__(self, cfg: TowerCfg, ops=None, trainable: bool = True, teacher_chunk: int = 2048):
        super().__init__(cfg, ops=ops, trainable=trainable, teacher_chunk=teacher_chunk)
        self.output_dim = cfg.embed_dim
        self.grid_size = (cfg.grid, cfg.grid)
        self.patch_size = (cfg.patch_size, cfg.patch_size)

    def extract_roi_features(self, x, normed_boxes, extract_type="v2", **kwargs):
        """The reference dispatches `extract_type` for this family (open_clip/transformer.py:515-521): 'v2' = the dense map + RoIAlign
        (differentiable), 'v1' = every box rasterised on the token grid and pooled by an extra query token (:660-671; inference)."""
        if extract_type == "v1":
            g = x.shape[-1] // self.cfg.patch_size
            return self.mask_attn_pool(x, [boxes_to_grid_masks(b, g, g) for b in normed_boxes])
        if extract_type != "v2":
            raise NotImplementedError(f"extract_type={extract_type!r}: the reference builds 'v1' and 'v2' (transformer.py:515-521)")
        return super().extract_roi_features(x, normed_boxes)

    def mask_attn_pool(self, image, masks):
        """VisionTransformer.mask_attn_pool (transformer.py:785-834): one extra query token per mask through every block.  Forward only: the
        hand-written backward does not cover the extra tokens, so a call that would need their gradient raises instead of silently detaching."""
        if torch.is_grad_enabled() and self._trainable():
            raise NotImplementedError("mask-attention pooling (extract_type='v1' / mask_attn=True) is built for inference: call it under "
                                      "torch.no_grad() or on a frozen tower; the training step differentiates extract_type='v2' only")
        return self.engine.mask_attn_pool(image.to(self.engine.device), list(masks))

    def _register_tables(self):
        pass                                                # no rotary tables in this family

    def lock(self, unlocked_groups=0, freeze_bn_stats=False):
        """transformer.py:391-422: of [[conv1, class_embedding, ln_pre], positional_embedding, block 0 .. L-1] the last `unlocked_groups`
        train (0 = all frozen; L + 1 adds the positional embedding, >= L + 2 the stem: ClipVitEngine._stem_bwd)."""
        super().lock(unlocked_groups=unlocked_groups, freeze_bn_stats=freeze_bn_stats)


class CLIP(CustomCLIP):
    """model.py:181-311: the text tower's tensors sit at the top level of the state dict (`transformer.*`, `token_embedding.weight`,
    `positional_embedding`, `ln_final.*`, `text_projection`), frozen and never executed by the distillation step."""

    def __init__(self, cfg: TowerCfg, ops=None, trainable: bool = True, with_text: bool = True):
        nn.Module.__init__(self)
        self.visual = ClipVisionTower(cfg, ops=ops, trainable=trainable)
        self.text = None
        text = FrozenTextTower(cfg, self.visual.engine.device, mask_in_state_dict=False) if with_text else None
        object.__setattr__(self, "_text", text)             # not a registered child: its keys carry no prefix
        self.embed_dim, self.vocab_size = cfg.embed_dim, cfg.text_vocab
        self.logit_scale = nn.Parameter(torch.ones([], device=self.visual.engine.device) * np.log(1 / 0.07))

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        if self._text is not None:
            self._text._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        if self._text is not None:
            own = {prefix + k for k in self._text._names.values()}
            self._text._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, [], error_msgs)
            state_dict = {k: v for k, v in state_dict.items() if k not in own}     # not "unexpected" for the parameters of this module
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
