"""Seeded, version-stable parameter and batch recipes.

No pretrained EVA checkpoints exist offline, so both towers start from this
recipe: every tensor is drawn from a numpy PCG64 stream keyed by
crc32(parameter name) ^ seed -- independent of torch's RNG, so the same
numbers are produced in the survey container (where the reference is imported
to make golden vectors) and on the GPU box.

Key names are the reference's state-dict names
(src/open_clip/eva_clip/eva_vit_model.py:411-453, :119-167, :82-105).
"""
from __future__ import annotations

import math
import zlib

import numpy as np
import torch

from .config import TowerCfg


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64((zlib.crc32(name.encode()) << 16) ^ (seed & 0xFFFF)))


def clip_vit_param_shapes(cfg: TowerCfg, prefix: str = "visual.") -> "dict[str, tuple]":
    """Shapes of every parameter of the OpenAI-CLIP vision transformer, in the reference's registration order
    (src/open_clip/transformer.py:355-389 and :190-215)."""
    C, Hd, E, p = cfg.width, cfg.hidden, cfg.embed_dim, cfg.patch_size
    out = {prefix + "class_embedding": (C,), prefix + "positional_embedding": (cfg.tokens, C), prefix + "proj": (C, E),
           prefix + "conv1.weight": (C, 3, p, p), prefix + "ln_pre.weight": (C,), prefix + "ln_pre.bias": (C,)}
    for i in range(cfg.layers):
        b = f"{prefix}transformer.resblocks.{i}."
        out.update({b + "ln_1.weight": (C,), b + "ln_1.bias": (C,), b + "attn.in_proj_weight": (3 * C, C), b + "attn.in_proj_bias": (3 * C,),
                    b + "attn.out_proj.weight": (C, C), b + "attn.out_proj.bias": (C,), b + "ln_2.weight": (C,), b + "ln_2.bias": (C,),
                    b + "mlp.c_fc.weight": (Hd, C), b + "mlp.c_fc.bias": (Hd,), b + "mlp.c_proj.weight": (C, Hd), b + "mlp.c_proj.bias": (C,)})
    out[prefix + "ln_post.weight"] = (C,)
    out[prefix + "ln_post.bias"] = (C,)
    return out


def visual_param_shapes(cfg: TowerCfg, prefix: str = "visual.") -> "dict[str, tuple]":
    """Shapes of every parameter of the vision tower, in the reference's registration order (EVA02: eva_vit_model.py:411-453)."""
    if getattr(cfg, "arch", "eva02") == "openai":
        return clip_vit_param_shapes(cfg, prefix)
    C, Hd, E, p = cfg.width, cfg.hidden, cfg.embed_dim, cfg.patch_size
    out = {}
    out[prefix + "cls_token"] = (1, 1, C)
    out[prefix + "pos_embed"] = (1, cfg.tokens, C)
    out[prefix + "patch_embed.proj.weight"] = (C, 3, p, p)
    out[prefix + "patch_embed.proj.bias"] = (C,)
    for i in range(cfg.layers):
        b = f"{prefix}blocks.{i}."
        out[b + "norm1.weight"] = (C,)
        out[b + "norm1.bias"] = (C,)
        out[b + "attn.q_bias"] = (C,)
        out[b + "attn.v_bias"] = (C,)
        out[b + "attn.q_proj.weight"] = (C, C)
        out[b + "attn.k_proj.weight"] = (C, C)
        out[b + "attn.v_proj.weight"] = (C, C)
        out[b + "attn.inner_attn_ln.weight"] = (C,)
        out[b + "attn.inner_attn_ln.bias"] = (C,)
        out[b + "attn.proj.weight"] = (C, C)
        out[b + "attn.proj.bias"] = (C,)
        out[b + "norm2.weight"] = (C,)
        out[b + "norm2.bias"] = (C,)
        out[b + "mlp.w1.weight"] = (Hd, C)
        out[b + "mlp.w1.bias"] = (Hd,)
        out[b + "mlp.w2.weight"] = (Hd, C)
        out[b + "mlp.w2.bias"] = (Hd,)
        out[b + "mlp.ffn_ln.weight"] = (Hd,)
        out[b + "mlp.ffn_ln.bias"] = (Hd,)
        out[b + "mlp.w3.weight"] = (C, Hd)
        out[b + "mlp.w3.bias"] = (C,)
    out[prefix + "norm.weight"] = (C,)
    out[prefix + "norm.bias"] = (C,)
    out[prefix + "head.weight"] = (E, C)
    out[prefix + "head.bias"] = (E,)
    return out


def seeded_visual_state(cfg: TowerCfg, seed: int = 0, prefix: str = "visual.") -> "dict[str, torch.Tensor]":
    """fp32 CPU tensors for every vision-tower parameter.

    Matrices ~ N(0, 0.02); proj / w3 are divided by sqrt(2*(layer+1)) like the
    reference's fix_init_weight (eva_vit_model.py:474-483).  Unlike the
    reference's constant init, biases and LayerNorm affine terms are *non-trivial*
    (N(0,0.02) and 1+N(0,0.1)) so that a dropped bias/gain shows up in parity tests.
    """
    sd = {}
    for name, shape in visual_param_shapes(cfg, prefix).items():
        g = _rng(name, seed)
        if name.endswith(".weight") and len(shape) == 1:          # LayerNorm gain
            t = 1.0 + 0.1 * g.standard_normal(shape)
        elif len(shape) == 1:                                      # biases
            t = 0.02 * g.standard_normal(shape)
        else:
            t = 0.02 * g.standard_normal(shape)
            if ".blocks." in name and (name.endswith("attn.proj.weight") or name.endswith("mlp.w3.weight")):
                layer = int(name.split(".blocks.")[1].split(".")[0])
                t = t / math.sqrt(2.0 * (layer + 1))
            elif name.endswith("visual.proj") or name.endswith("positional_embedding"):
                t = t * (50.0 * cfg.width ** -0.5)          # scale * randn with scale = width^-0.5 (transformer.py:360-362,387)
        sd[name] = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32))
    return sd


def synthetic_batch(batch: int, boxes_per_image: int, image_size: int, crop_size: int,
                    seed: int = 1234, rank: int = 0, valid_prob: float = 1.0):
    """SURVEY.md §8 M2 synthetic step input with the reference batch contract
    (src/training/data.py:247-281): images [B,3,S,S], normed_boxes [B,k,5]
    = (x0,y0,x1,y1 in [0,1], valid), image_crops [B,k,3,Sc,Sc]; fp32 on CPU."""
    g = np.random.Generator(np.random.PCG64(seed + rank))
    images = g.standard_normal((batch, 3, image_size, image_size), dtype=np.float32)
    crops = g.standard_normal((batch, boxes_per_image, 3, crop_size, crop_size), dtype=np.float32)
    xy0 = g.uniform(0.0, 0.6, size=(batch, boxes_per_image, 2))
    wh = g.uniform(0.1, 0.4, size=(batch, boxes_per_image, 2))
    xy1 = np.minimum(xy0 + wh, 1.0)
    if valid_prob >= 1.0:
        valid = np.ones((batch, boxes_per_image, 1))
    else:
        valid = (g.uniform(size=(batch, boxes_per_image, 1)) < valid_prob).astype(np.float64)
        valid[:, 0, 0] = 1.0                    # at least one valid box per image
    boxes = np.concatenate([xy0, xy1, valid], axis=-1).astype(np.float32)
    crops = crops * valid[..., None, None].astype(np.float32)   # zero rows where invalid (data.py:262-281)
    return torch.from_numpy(images), torch.from_numpy(boxes), torch.from_numpy(crops)
