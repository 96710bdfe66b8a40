"""Zero-shot region classification (reference: src/training/zero_shot.py:11-193).

Each validation batch follows the panoptic val dataset's contract (src/training/data.py:331-387):
    images [B,3,S,S], bboxes [B,K,8] = (x0,y0,x1,y1 in [0,1], class, valid, area, is_thing), image_crops [B,K,3,Sc,Sc],
    gt_masks [B,K,h,w] (feature-map resolution), masked_image_crops [B,K,3,Sc,Sc]
and `dataloader.dataset.embeddings` holds one text embedding per class.  Three region descriptors are scored against the class
embeddings -- RoIAlign over the dense map, mask pooling of the dense map, the plain image embedding of the crop -- and reported as
mean per-class top-1 / top-5 accuracy, things and stuff apart.  Features come from the HIP engine (encode_pseudo_boxes /
encode_masks / encode_image); the scoring itself is a few small torch ops outside the training hot path.
"""
import logging

import torch
import torch.distributed as dist
import torch.nn.functional as F

from ..open_clip import get_cast_dtype
from .precision import get_autocast


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def _topk_hits(logits, labels):
    return logits.topk(5).indices == labels.view(-1, 1)


def run(model, dataloader, args):
    cls_emb = torch.as_tensor(dataloader.dataset.embeddings).float()
    cls_emb = F.normalize(cls_emb, dim=-1).to(args.device)
    cast_dtype = get_cast_dtype(args.precision)
    if cast_dtype is not None:
        cls_emb = cls_emb.to(cast_dtype)
    module = _unwrap(model)
    acc = {k: [] for k in ("hit_rois", "hit_crops", "hit_maskpool", "sim_rois", "sim_crops", "sim_maskpool", "size", "thing", "label")}
    with torch.no_grad():
        for images, bboxes, image_crops, gt_masks, masked_image_crops in dataloader:
            images, bboxes, image_crops, gt_masks = (t.to(args.device) for t in (images, bboxes, image_crops, gt_masks))
            if cast_dtype is not None:
                images, bboxes, image_crops, gt_masks = (t.to(cast_dtype) for t in (images, bboxes, image_crops, gt_masks))
            rois, masks, crops, labels = [], [], [], []
            for boxes_i, crops_i, masks_i in zip(bboxes, image_crops, gt_masks):
                keep = boxes_i[:, 5] > 0.5
                rois.append(boxes_i[keep, :4])
                labels.append(boxes_i[keep, 4])
                crops.append(crops_i[keep])
                masks.append(masks_i[keep])
                acc["size"].append(boxes_i[keep, 6].float())
                acc["thing"].append(boxes_i[keep, 7])
            labels = torch.cat(labels).to(torch.long)
            if labels.numel() == 0:
                continue
            with get_autocast(args.precision)():
                feats = {
                    "rois": module.encode_pseudo_boxes(images, rois, normalize=True, extract_type=args.extract_type),
                    "maskpool": module.encode_masks(images, masks, normalize=True, mask_attn=args.extract_type == "v1"),
                }
                crops = torch.cat(crops)
                if getattr(args, "image_ave_pool", False):
                    fmap = module.visual.encode_dense(crops, keep_shape=True)
                    feats["crops"] = F.normalize(fmap.mean(dim=(-2, -1)), dim=-1)
                else:
                    feats["crops"] = module.encode_image(crops, normalize=True)
                for key, f in feats.items():
                    logits = (f.to(cast_dtype) if cast_dtype is not None else f) @ cls_emb.T
                    acc["hit_" + key].append(_topk_hits(logits, labels))
                    acc["sim_" + key].append(torch.gather(logits, 1, labels.view(-1, 1))[:, 0])
            acc["label"].append(labels)
    out = {k: torch.cat(v).float() if k != "label" else torch.cat(v) for k, v in acc.items()}
    if getattr(args, "distributed", False) and not getattr(args, "horovod", False):
        out = {k: multi_gpu_sync(v) for k, v in out.items()}
    return out


def multi_gpu_sync(x):
    """Concatenate per-rank results (object all-gather, as training/dist_utils.py:135-155 does)."""
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, x.cpu())
    return torch.cat([g.to(x.device) for g in gathered])


def macc_with_is_thing(hits, is_thing, labels, prefix):
    """Mean per-class accuracy over the classes that occur (zero_shot.py:135-173), top-1 and top-5, things vs stuff; the
    per-class means go through fp16 like the reference's `.mean().half().item()`."""
    def macc(correct, cls):
        if cls.numel() == 0:
            return float("nan")
        per_class = [correct[cls == c].mean().half().item() for c in range(int(cls.min()), int(cls.max()) + 1) if (cls == c).any()]
        return sum(per_class) / len(per_class)

    res = {}
    for name, sel in (("thing", is_thing > 0), ("stuff", is_thing < 1)):
        h, c = hits[sel], labels[sel].long()
        res[f"{prefix}.{name}.macc1"] = macc(h[:, 0], c)
        res[f"{prefix}.{name}.macc5"] = macc(h.sum(-1), c)
    return res


def zero_shot_eval(model, data, epoch, args):
    if "val" not in data or args.zeroshot_frequency == 0:
        return {}
    if (epoch % args.zeroshot_frequency) != 0 and epoch != args.epochs:
        return {}
    logging.info("Region classifier")
    r = run(model, data["val"].dataloader, args)
    results = {}
    for key in ("rois", "crops", "maskpool"):
        results.update(macc_with_is_thing(r["hit_" + key], r["thing"], r["label"], key))
    return results
