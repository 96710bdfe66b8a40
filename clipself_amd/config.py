"""Architecture records for the vision towers on the CLIPSelf hot path.

The numbers mirror the reference's JSON model configs
(reference: src/open_clip/eva_clip/model_configs/EVA02-CLIP-B-16.json,
EVA02-CLIP-L-14-336.json) and the wiring in
src/open_clip/eva_clip/model.py:92-131 (``_build_vision_tower``) and
src/open_clip/eva_clip/eva_vit_model.py:396-470; for the OpenAI-CLIP ViT family
(``arch == "openai"``: learned positional embedding, fused-QKV attention, GELU MLP, ln_pre / ln_post / proj)
src/open_clip/model_configs/ViT-*.json, src/open_clip/model.py:24-49,77-139 and
src/open_clip/transformer.py:318-389.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, asdict
from pathlib import Path


@dataclass(frozen=True)
class TowerCfg:
    name: str
    embed_dim: int          # E: CLIP embedding width (head output)
    image_size: int         # native square input
    patch_size: int
    width: int              # C
    layers: int             # L
    head_width: int = 64
    mlp_ratio: float = 2.6667
    pt_hw_seq_len: int = 16  # RoPE pre-training grid (rope.py:96-142)
    ln_eps: float = 1e-6     # eva_clip/model.py:123
    # text tower census only (state-dict compat; never executed on the hot path)
    text_width: int = 512
    text_heads: int = 8
    text_layers: int = 12
    text_context: int = 77
    text_vocab: int = 49408
    arch: str = "eva02"      # "eva02" (RoPE + SwiGLU + sub-LN) | "openai" (OpenAI-CLIP ViT)
    quick_gelu: bool = False  # openai arch only: QuickGELU instead of nn.GELU (model.py:77-90)

    @property
    def heads(self) -> int:
        return self.width // self.head_width

    @property
    def hidden(self) -> int:
        # eva_vit_model.py:274  int(dim * mlp_ratio)
        return int(self.width * self.mlp_ratio)

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def tokens(self) -> int:
        return self.grid * self.grid + 1

    def grid_for(self, image_hw: int) -> int:
        return image_hw // self.patch_size


_CFG_DIR = Path(__file__).parent / "open_clip" / "model_configs"


_EVA_FLAGS = ("rope", "naiveswiglu", "subln", "intp_freq")


def _from_json(name: str, blob: dict) -> TowerCfg:
    v, t = blob["vision_cfg"], blob["text_cfg"]
    if not any(f in v for f in _EVA_FLAGS):
        # plain OpenAI-CLIP ViT config (model.py:24-49): integer depth, no timm / attentional-pool / patch-norm variants
        odd = [k for k in ("timm_model_name", "attentional_pool", "global_average_pool", "input_patchnorm", "ls_init_value") if v.get(k)]
        if odd or not isinstance(v.get("layers"), int):
            raise NotImplementedError(f"{name}: vision_cfg options {odd or 'layers'} are outside the CLIPSelf hot path")
        return TowerCfg(
            name=name, embed_dim=blob["embed_dim"], image_size=v["image_size"], patch_size=v["patch_size"], width=v["width"],
            layers=v["layers"], head_width=v.get("head_width", 64), mlp_ratio=v.get("mlp_ratio", 4.0), ln_eps=1e-5,
            text_width=t["width"], text_heads=t["heads"], text_layers=t["layers"], text_context=t.get("context_length", 77),
            text_vocab=t.get("vocab_size", 49408), arch="openai", quick_gelu=bool(blob.get("quick_gelu", False)))
    for flag in _EVA_FLAGS:
        if not v.get(flag, False):
            raise NotImplementedError(
                f"{name}: of the EVA family only the EVA02 (rope+swiglu+subln) towers are on the hot path")
    return TowerCfg(
        name=name, embed_dim=blob["embed_dim"], image_size=v["image_size"],
        patch_size=v["patch_size"], width=v["width"], layers=v["layers"],
        head_width=v.get("head_width", 64), mlp_ratio=v.get("mlp_ratio", 4.0),
        pt_hw_seq_len=v.get("pt_hw_seq_len", 16),
        text_width=t["width"], text_heads=t["heads"], text_layers=t["layers"],
        text_context=t.get("context_length", 77), text_vocab=t.get("vocab_size", 49408))


def list_models():
    return sorted(p.stem for p in _CFG_DIR.glob("*.json"))


def get_tower_cfg(model_name: str) -> TowerCfg:
    model_name = model_name.replace("/", "-")
    path = _CFG_DIR / f"{model_name}.json"
    if not path.exists():
        raise RuntimeError(f"Model config for {model_name} not found; available models {list_models()}.")
    return _from_json(model_name, json.loads(path.read_text()))


def tiny_cfg() -> TowerCfg:
    """The small EVA02-shaped tower (head dim 64 like both shipped towers) used for full-tensor golden vectors
    (SURVEY.md Appendix B item 9)."""
    return TowerCfg(name="EVA02-tiny-test", embed_dim=64, image_size=32, patch_size=8,
                    width=128, layers=2, head_width=64, mlp_ratio=2.0,
                    text_width=32, text_heads=2, text_layers=1)


def tiny14_cfg() -> TowerCfg:
    """A small tower with the awkward dimensions of EVA02-CLIP-L-14-336: patch 14 (3*14*14 = 588, padded to 640 in
    storage) and mlp_ratio 2.6667 (hidden int(128*2.6667) = 341, padded to 384) -- exercises every zero-padding path."""
    return TowerCfg(name="EVA02-tiny14-test", embed_dim=64, image_size=42, patch_size=14, width=128, layers=2,
                    head_width=64, mlp_ratio=2.6667, text_width=32, text_heads=2, text_layers=1)


def tiny_openai_cfg(quick_gelu: bool = False) -> TowerCfg:
    """A small OpenAI-CLIP-shaped ViT (head dim 64, mlp_ratio 4, LayerNorm eps 1e-5) for full-tensor golden vectors."""
    return TowerCfg(name="ViT-tiny-test" + ("-quickgelu" if quick_gelu else ""), embed_dim=64, image_size=32, patch_size=8, width=128,
                    layers=2, head_width=64, mlp_ratio=4.0, ln_eps=1e-5, text_width=32, text_heads=2, text_layers=1, text_context=8,
                    text_vocab=64, arch="openai", quick_gelu=quick_gelu)


def tiny_openai14_cfg() -> TowerCfg:
    """ViT-L/14-shaped miniature of the OpenAI family: patch 14 (3*14*14 = 588 -> conv1 stored with K padded to 640), 3x3 grid."""
    return TowerCfg(name="ViT-tiny14-test", embed_dim=64, image_size=42, patch_size=14, width=128, layers=2, head_width=64, mlp_ratio=4.0,
                    ln_eps=1e-5, text_width=32, text_heads=2, text_layers=1, text_context=8, text_vocab=64, arch="openai")


def cfg_dict(cfg: TowerCfg) -> dict:
    return asdict(cfg)
