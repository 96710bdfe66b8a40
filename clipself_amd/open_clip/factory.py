"""`open_clip` factory surface for the CLIPSelf hot path.

Mirrors the call signatures of the reference (src/open_clip/factory.py:111-264 `create_model`, :267-350
`create_model_and_transforms`; src/open_clip/eva_clip/factory.py:211-355) for the EVA towers
(`pretrained='eva'`, checkpoint path passed as `cache_dir`) and the plain OpenAI-CLIP ViT configs (`ViT-B-16`, `ViT-L-14`,
`ViT-L-14-336`; `pretrained='openai'` or a checkpoint path).  Anything else the reference's zoo can build
(ResNets, timm, CoCa, HF text towers) is outside the hot path and raises.
"""
from __future__ import annotations

import logging
import os
from typing import Optional

import torch

import dataclasses

from ..config import get_tower_cfg, list_models as _list_models
from .model import CLIP, CustomCLIP


def list_models():
    return _list_models()


def get_cast_dtype(precision: str):
    # eva_clip/model.py:83-89
    if precision == "bf16":
        return torch.bfloat16
    if precision == "fp16":
        return torch.float16
    return None


def load_state_dict(checkpoint_path: str, map_location="cpu", model_key="model|module|state_dict"):
    """eva_clip/factory.py:80-106: pick the state dict out of a checkpoint, strip 'module.', drop RoPE tables."""
    checkpoint = torch.load(checkpoint_path, map_location=map_location, weights_only=False)
    state_dict = checkpoint
    for mk in model_key.split("|"):
        if isinstance(checkpoint, dict) and mk in checkpoint:
            state_dict = checkpoint[mk]
            break
    if next(iter(state_dict.items()))[0].startswith("module"):
        state_dict = {k[7:]: v for k, v in state_dict.items()}
    return {k: v for k, v in state_dict.items() if "freqs_cos" not in k and "freqs_sin" not in k}


def _resize_pos_embed(state_dict, model):
    """eva_clip/utils.py:78-106 (resize_evaclip_pos_embed): bicubic resize of a checkpoint's grid to the model's."""
    key = "visual.pos_embed"
    if key not in state_dict:
        return
    pe = state_dict[key].float()
    cfg = model.visual.cfg
    new_n = cfg.grid * cfg.grid
    old_n = pe.shape[1] - 1
    if old_n == new_n:
        return
    old_g = int(old_n ** 0.5)
    body = pe[:, 1:].reshape(1, old_g, old_g, -1).permute(0, 3, 1, 2)
    body = torch.nn.functional.interpolate(body, size=(cfg.grid, cfg.grid), mode="bicubic", align_corners=False)
    state_dict[key] = torch.cat([pe[:, :1], body.permute(0, 2, 3, 1).flatten(1, 2)], dim=1)


def load_checkpoint(model, checkpoint_path, strict=False):
    sd = load_state_dict(checkpoint_path)
    if "text.logit_scale" in sd:
        sd["logit_scale"] = sd.pop("text.logit_scale")
    _resize_pos_embed(sd, model)
    return model.load_state_dict(sd, strict=strict)


def create_model(model_name: str, pretrained: Optional[str] = None, precision: str = "fp32", device="cpu", jit: bool = False,
                 force_quick_gelu: bool = False, force_custom_text: bool = False, force_patch_dropout=None,
                 force_image_size=None, pretrained_image: bool = False, pretrained_hf: bool = True,
                 cache_dir: Optional[str] = None, output_dict: Optional[bool] = None, require_pretrained: bool = False,
                 ops=None, trainable: bool = True):
    """Returns a CustomCLIP whose vision tower runs on the HIP engine.

    `precision`: every reference value (amp, amp_bf16, bf16, fp16, fp32) maps to the one numerics of the HIP engine -- bf16 MFMA
    operands, fp32 accumulation / statistics / master weights; "amp_fp8" (or "fp8") additionally runs the forward
    linears of the training schedule on e4m3 operands (BASELINE configs[4] "fp8 MFMA weights", EvaEngine.enable_fp8_forward).
    `device` must be a ROCm device.
    """
    model_name = model_name.replace("/", "-")
    if jit:
        raise NotImplementedError("torchscript is not supported by the HIP engine")
    cfg = get_tower_cfg(model_name)
    fp8 = precision in ("fp8", "amp_fp8", "amp_fp8_dgrad")
    if cfg.arch == "openai":
        if fp8:
            raise NotImplementedError("precision='amp_fp8' exists for the EVA02 towers (RegionCLIP configuration) only")
        return _create_openai_vit(cfg, model_name, pretrained, force_quick_gelu, cache_dir, require_pretrained, ops, trainable)
    if pretrained not in ("eva", None, ""):
        raise NotImplementedError(f"pretrained={pretrained!r}: the EVA towers load with pretrained='eva' (checkpoint path in cache_dir)")
    os.environ["RoPE"] = "1"                      # side effect of the reference factory (eva_clip/factory.py:249-253)
    model = CustomCLIP(cfg, ops=ops, trainable=trainable)
    if cache_dir and os.path.exists(cache_dir) and os.path.isfile(cache_dir):
        logging.info(f"Loading pretrained {model_name} weights ({cache_dir}).")
        load_checkpoint(model, cache_dir, strict=False)
    elif cache_dir and require_pretrained:
        raise RuntimeError(f"Pretrained weights ({cache_dir}) not found for model {model_name}.")
    else:
        from ..init import seeded_visual_state
        logging.info(f"No checkpoint at {cache_dir!r}: {model_name} starts from the seeded random initialisation")
        model.visual.engine.load_state(seeded_visual_state(cfg, seed=0))
    if fp8 and trainable:
        # amp_fp8 applies to the TRAINING forward schedule (the student of configs[4]); frozen towers -- the CLIPSelf teacher, the
        # end-of-epoch evaluation copy -- stay on bf16 operands, so distillation targets and evaluation features do not depend on
        # the precision flag of the run
        # "amp_fp8_dgrad" (or CLIPSELF_FP8_DGRAD=1): the dgrad GEMMs of the backward contract e4m3 operands as well (wgrad stays bf16)
        model.visual.engine.enable_fp8_forward(dgrad=precision == "amp_fp8_dgrad" or os.environ.get("CLIPSELF_FP8_DGRAD") == "1")
    return model


def load_openai_state_dict(path: str):
    """State dict of an OpenAI release file (a TorchScript archive, or a plain state dict): the keys are already the `CLIP` names
    (src/open_clip/openai.py:44-76, model.py:417-474); the three shape scalars are dropped like the reference does."""
    try:
        sd = torch.jit.load(path, map_location="cpu").state_dict()
    except RuntimeError:
        sd = torch.load(path, map_location="cpu", weights_only=False)
        sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd.state_dict()
    return {k: v.float() for k, v in sd.items() if k not in ("input_resolution", "context_length", "vocab_size")}


def _create_openai_vit(cfg, model_name, pretrained, force_quick_gelu, cache_dir, require_pretrained, ops, trainable):
    """factory.py:150-232: `pretrained='openai'` = the OpenAI release (QuickGELU, file <cache_dir>/<model>.pt -- there is no network
    to download it from); another non-empty `pretrained` = path of an open_clip checkpoint; otherwise seeded random initialisation."""
    openai = bool(pretrained) and pretrained.lower() == "openai"
    if openai or force_quick_gelu:
        cfg = dataclasses.replace(cfg, quick_gelu=True)
    model = CLIP(cfg, ops=ops, trainable=trainable)
    path = None
    if openai:
        path = os.path.join(cache_dir or "", f"{model_name}.pt")
    elif pretrained:
        path = pretrained
    if path and os.path.isfile(path):
        logging.info(f"Loading pretrained {model_name} weights ({path}).")
        sd = load_openai_state_dict(path) if openai else load_state_dict(path)
        model.load_state_dict(sd, strict=False)
    elif path and require_pretrained:
        raise RuntimeError(f"Pretrained weights ({path}) not found for model {model_name}.")
    else:
        from ..init import seeded_visual_state
        logging.info(f"No checkpoint at {path!r}: {model_name} starts from the seeded random initialisation")
        model.visual.engine.load_state(seeded_visual_state(cfg, seed=0))
    return model


def create_model_and_transforms(model_name: str, pretrained: Optional[str] = None, precision: str = "fp32", device="cpu",
                                jit: bool = False, force_quick_gelu: bool = False, force_custom_text: bool = False,
                                force_patch_dropout=None, force_image_size=None, pretrained_image: bool = False,
                                pretrained_hf: bool = True, image_mean=None, image_std=None, aug_cfg=None,
                                cache_dir: Optional[str] = None, output_dict: Optional[bool] = None,
                                det_image_size=1024, dataset_type=None, ops=None):
    model = create_model(model_name, pretrained, precision=precision, device=device, jit=jit,
                         force_quick_gelu=force_quick_gelu, force_custom_text=force_custom_text,
                         force_patch_dropout=force_patch_dropout, force_image_size=force_image_size,
                         pretrained_image=pretrained_image, pretrained_hf=pretrained_hf, cache_dir=cache_dir,
                         output_dict=output_dict, ops=ops)
    # factory.py:303-350 of the reference: [det transform (ResizeLongest), crop transform (ResizeMaxSize)] for the distillation /
    # region_clip datasets, the train-augmentation transform otherwise; the validation pair is always [det, crop]
    from .transform import det_image_transform, image_transform
    image_mean = image_mean or getattr(model.visual, "image_mean", None)
    image_std = image_std or getattr(model.visual, "image_std", None)
    kops = ops if ops is not None else getattr(getattr(model.visual, "engine", None), "ops", None)
    val_det = det_image_transform(det_image_size, is_train=False, mean=image_mean, std=image_std, ops=kops)
    val_img = image_transform(model.visual.image_size, is_train=False, mean=image_mean, std=image_std, resize_longest_max=True, ops=kops)
    if dataset_type == "sanity_check":
        train = image_transform(det_image_size, is_train=True, mean=image_mean, std=image_std, aug_cfg=aug_cfg)
    elif dataset_type is not None and ("distill" in dataset_type or dataset_type in ("region_clip", "clipself", "clipself_proposals", "coop")):
        train = [val_det, val_img]
    else:
        train = image_transform(model.visual.image_size, is_train=True, mean=image_mean, std=image_std, aug_cfg=aug_cfg)
    return model, train, [val_det, val_img]


def get_tokenizer(model_name):
    def _no_text(*a, **k):
        raise NotImplementedError("tokenisation / the text tower are outside the CLIPSelf hot path")
    return _no_text
