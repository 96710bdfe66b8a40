"""TEST INFRASTRUCTURE -- CPU oracle, never imported by the product path.

fp32 CPU restatement (plain torch) of the OpenAI-CLIP vision transformer on the CLIPSelf hot path:

  CLIP.encode_image / encode_dense / encode_pseudo_boxes     src/open_clip/model.py:224-243
  VisionTransformer.forward / encode_dense / _extract_roi_features_v2
                                                              src/open_clip/transformer.py:443-494,550-589,685-722
  rescale_positional_embedding / _denormalize_boxes           transformer.py:724-734,649-657
  Transformer.forward / extract_feature_map                   transformer.py:288-306
  ResidualAttentionBlock(V2).forward / forward_without_attn   transformer.py:232-260
  nn.MultiheadAttention (fused in_proj, bias on q, k and v; 1/sqrt(head_dim) on q)   as called at transformer.py:203,217-230
  QuickGELU                                                   transformer.py:31-34
  VisionTransformer.lock                                      transformer.py:391-422
  _extract_roi_features_v1 / mask_attn_pool / _mask_attn_pool  transformer.py:636-646,660-671,736-834 (inference)

Functional (a dict of tensors keyed by the reference's state-dict names), like oracle/eva_ref.py whose helpers it reuses.
Pinned against the reference itself: oracle/gen_golden.py (--openai) imports the real reference in the build container and
tests/test_oracle_vs_golden.py checks this restatement against the captured outputs (tests/golden/tiny_openai_step.npz).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .eva_ref import _Round, layer_norm, rois_from_list
from .roi_align_ref import roi_align_1x1


def positional_embedding_for(sd, cfg, grid: int, prefix="visual."):
    pe = sd[prefix + "positional_embedding"]
    if grid == cfg.grid:
        return pe
    C = pe.shape[1]                                            # transformer.py:724-734: bicubic, align_corners=False, class slot copied
    pe2 = pe[1:].T.contiguous().view(1, C, cfg.grid, cfg.grid)
    pe2 = F.interpolate(pe2, (grid, grid), mode="bicubic", align_corners=False).view(C, grid * grid)
    return torch.cat([pe[:1], pe2.T], dim=0)


def stem(sd, cfg, images, rq, prefix="visual."):
    """conv1 (no bias) as unfold-GEMM + class/positional embeddings + ln_pre (transformer.py:551-569)."""
    B, _, Hh, _ = images.shape
    p = cfg.patch_size
    g = Hh // p
    w = sd[prefix + "conv1.weight"].reshape(cfg.width, -1)
    patches = images.reshape(B, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * p * p)
    x = rq(patches) @ rq(w).T
    cls = sd[prefix + "class_embedding"].expand(B, 1, -1)
    x = torch.cat((cls, x), dim=1) + positional_embedding_for(sd, cfg, g, prefix)
    return layer_norm(x, sd[prefix + "ln_pre.weight"], sd[prefix + "ln_pre.bias"], cfg.ln_eps), g


def activation(x, quick: bool):
    return x * torch.sigmoid(1.702 * x) if quick else F.gelu(x)


def attention(sd, cfg, x, blk, rq):
    B, N, C = x.shape
    H, d = cfg.heads, cfg.head_width
    qkv = rq(rq(x) @ rq(sd[blk + "attn.in_proj_weight"]).T + sd[blk + "attn.in_proj_bias"])
    q, k, v = (t.reshape(B, N, H, d).permute(0, 2, 1, 3) for t in qkv.split(C, dim=-1))
    att = ((q * d ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
    o = rq((rq(att) @ v).transpose(1, 2).reshape(B, N, C))
    return o @ rq(sd[blk + "attn.out_proj.weight"]).T + sd[blk + "attn.out_proj.bias"]


def proj_without_attn(sd, cfg, x, blk, rq):
    C = cfg.width                                              # transformer.py:248-255: the value third of in_proj, then out_proj
    v = rq(rq(x) @ rq(sd[blk + "attn.in_proj_weight"][2 * C:]).T + sd[blk + "attn.in_proj_bias"][2 * C:])
    return v @ rq(sd[blk + "attn.out_proj.weight"]).T + sd[blk + "attn.out_proj.bias"]


def block(sd, cfg, x, i, rq, with_attn=True, prefix="visual."):
    blk = f"{prefix}transformer.resblocks.{i}."
    n1 = rq(layer_norm(x, sd[blk + "ln_1.weight"], sd[blk + "ln_1.bias"], cfg.ln_eps))
    x = x + (attention(sd, cfg, n1, blk, rq) if with_attn else proj_without_attn(sd, cfg, n1, blk, rq))
    n2 = rq(layer_norm(x, sd[blk + "ln_2.weight"], sd[blk + "ln_2.bias"], cfg.ln_eps))
    fc = rq(n2 @ rq(sd[blk + "mlp.c_fc.weight"]).T + sd[blk + "mlp.c_fc.bias"])
    return x + rq(activation(fc, cfg.quick_gelu)) @ rq(sd[blk + "mlp.c_proj.weight"]).T + sd[blk + "mlp.c_proj.bias"]


def encode_image(sd, cfg, images, emulate_bf16=False, prefix="visual."):
    """forward(): every block with attention, ln_post on the class token, @ proj (transformer.py:443-494)."""
    rq = _Round(emulate_bf16)
    x, _ = stem(sd, cfg, images, rq, prefix)
    for i in range(cfg.layers):
        x = block(sd, cfg, x, i, rq, True, prefix)
    pooled = rq(layer_norm(x[:, 0], sd[prefix + "ln_post.weight"], sd[prefix + "ln_post.bias"], cfg.ln_eps))
    return pooled @ rq(sd[prefix + "proj"])


def encode_dense(sd, cfg, images, emulate_bf16=False, prefix="visual."):
    """encode_dense(): last block without attention, ln_post on the patch tokens, @ proj, L2 normalise (transformer.py:550-589)."""
    rq = _Round(emulate_bf16)
    x, g = stem(sd, cfg, images, rq, prefix)
    for i in range(cfg.layers - 1):
        x = block(sd, cfg, x, i, rq, True, prefix)
    x = block(sd, cfg, x, cfg.layers - 1, rq, False, prefix)[:, 1:]
    x = rq(layer_norm(x, sd[prefix + "ln_post.weight"], sd[prefix + "ln_post.bias"], cfg.ln_eps)) @ rq(sd[prefix + "proj"])
    return F.normalize(x, dim=-1), g


def encode_pseudo_boxes(sd, cfg, images, normed_boxes_list, emulate_bf16=False, prefix="visual."):
    dense, g = encode_dense(sd, cfg, images, emulate_bf16, prefix)
    return roi_align_1x1(dense.reshape(images.shape[0], g, g, -1), rois_from_list(normed_boxes_list, g))


def boxes_to_masks(normed_boxes, grid_h: int, grid_w: int):
    """_generate_masks_per_image (transformer.py:636-646): box * (w, h, w, h), truncated to integers, rows y0:y1 / columns x0:x1 set."""
    boxes = normed_boxes * torch.tensor([[grid_w, grid_h, grid_w, grid_h]], dtype=normed_boxes.dtype)
    masks = torch.zeros(len(normed_boxes), grid_h, grid_w, dtype=torch.bool)
    for i, box in enumerate(boxes):
        x0, y0, x1, y1 = box.long().tolist()
        masks[i, y0:y1, x0:x1] = True
    return masks


def mask_attn_pool(sd, cfg, images, masks, emulate_bf16=False, prefix="visual."):
    """mask_attn_pool / _mask_attn_pool (transformer.py:736-834).  masks: list over images of bool [n_i, g, g].  Every mask is an extra token --
    a copy of the image's CLS embedding after ln_pre -- that runs through all blocks; the attention mask (:806-818) hides the extra tokens from
    everybody (column block [:, :Q] masked) and lets token q see the CLS token plus the image tokens inside its mask, so the image tokens evolve
    exactly as in forward() and each extra token is a query-only passenger: x_q += out_proj(softmax(q_q K^T / 8 | allowed) V), then the MLP.
    Images with fewer masks are padded with see-everything tokens whose rows are dropped (:793-795,823-824).  Returns [sum n_i, E]."""
    rq = _Round(emulate_bf16)
    x, g = stem(sd, cfg, images, rq, prefix)                      # [B, N, C]
    B, N, C = x.shape
    H, d = cfg.heads, cfg.head_width
    counts = [int(m.shape[0]) for m in masks]
    Q = max(counts)
    allow = torch.ones(B, Q, N, dtype=torch.bool)                 # key allowed?  column 0 = CLS: always
    for b, m in enumerate(masks):
        allow[b, :m.shape[0], 1:] = m.reshape(m.shape[0], -1)
    xm = x[:, :1].expand(B, Q, C).clone()
    for i in range(cfg.layers):
        blk = f"{prefix}transformer.resblocks.{i}."
        Wqkv, bqkv = sd[blk + "attn.in_proj_weight"], sd[blk + "attn.in_proj_bias"]
        n1 = rq(layer_norm(x, sd[blk + "ln_1.weight"], sd[blk + "ln_1.bias"], cfg.ln_eps))
        kv = rq(rq(n1) @ rq(Wqkv[C:]).T + bqkv[C:])               # keys / values of the image's own tokens at this depth
        k, v = (t.reshape(B, N, H, d).permute(0, 2, 1, 3) for t in kv.split(C, dim=-1))
        nm = rq(layer_norm(xm, sd[blk + "ln_1.weight"], sd[blk + "ln_1.bias"], cfg.ln_eps))
        q = rq(rq(nm) @ rq(Wqkv[:C]).T + bqkv[:C]).reshape(B, Q, H, d).permute(0, 2, 1, 3)
        sc = (q * d ** -0.5) @ k.transpose(-2, -1)                # [B, H, Q, N]
        att = sc.masked_fill(~allow[:, None], float("-inf")).softmax(dim=-1)
        om = rq((rq(att) @ v).transpose(1, 2).reshape(B, Q, C))
        xm = xm + om @ rq(sd[blk + "attn.out_proj.weight"]).T + sd[blk + "attn.out_proj.bias"]
        n2 = rq(layer_norm(xm, sd[blk + "ln_2.weight"], sd[blk + "ln_2.bias"], cfg.ln_eps))
        fc = rq(n2 @ rq(sd[blk + "mlp.c_fc.weight"]).T + sd[blk + "mlp.c_fc.bias"])
        xm = xm + rq(activation(fc, cfg.quick_gelu)) @ rq(sd[blk + "mlp.c_proj.weight"]).T + sd[blk + "mlp.c_proj.bias"]
        if i + 1 < cfg.layers:
            x = block(sd, cfg, x, i, rq, True, prefix)            # the image tokens never see the extra ones
    pooled = rq(layer_norm(xm, sd[prefix + "ln_post.weight"], sd[prefix + "ln_post.bias"], cfg.ln_eps)) @ rq(sd[prefix + "proj"])
    return torch.cat([pooled[b, :n] for b, n in enumerate(counts)])


def extract_roi_features_v1(sd, cfg, images, normed_boxes_list, emulate_bf16=False, prefix="visual."):
    """_extract_roi_features_v1 (transformer.py:660-671): the boxes rasterised on the token grid, then mask_attn_pool."""
    g = images.shape[-1] // cfg.patch_size
    return mask_attn_pool(sd, cfg, images, [boxes_to_masks(b, g, g) for b in normed_boxes_list], emulate_bf16, prefix)


def trainable_names(sd, cfg, unlocked_groups: int, prefix="visual."):
    """lock (transformer.py:391-422): groups [stem, positional_embedding, block 0 .. L-1]; the last n train, n = 0 freezes all;
    ln_post and proj stay frozen; logit_scale trains (model.py:214)."""
    groups = [[prefix + "conv1.weight", prefix + "class_embedding", prefix + "ln_pre.weight", prefix + "ln_pre.bias"],
              [prefix + "positional_embedding"]]
    for i in range(cfg.layers):
        tag = f"{prefix}transformer.resblocks.{i}."
        groups.append([n for n in sd if n.startswith(tag)])
    if unlocked_groups < 0:         # no lock_image_tower() call at all (training.main without --lock-image): the whole visual tower trains
        groups.append([prefix + "ln_post.weight", prefix + "ln_post.bias", prefix + "proj"])
        unlocked_groups = len(groups)
    keep = [n for grp in (groups[-unlocked_groups:] if unlocked_groups else []) for n in grp]
    return keep + (["logit_scale"] if "logit_scale" in sd else [])
