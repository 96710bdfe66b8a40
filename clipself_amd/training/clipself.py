"""CLIPSelf distillation method -- same call contract as the reference's src/training/clipself.py:6-49:

    losses, batch_size, logit_scale = CLIPSelf()(batch, model, dist_model, loss, device, cast_dtype, distributed, args)

batch = (images [B,3,S,S], normed_boxes [B,max_boxes,5] = (x0,y0,x1,y1 in [0,1], valid), image_crops [B,max_boxes,3,Sc,Sc]).
The teacher forward, the student dense forward, RoIAlign, both L2 normalisations, the cosine loss and the whole
backward run as HIP kernels; the tensors handed back (`loss_cosine`) are ordinary autograd leaves of that path.
"""
import random

import torch
import torch.nn.functional as F


class _CosineDistillFn(torch.autograd.Function):
    """loss = w * (1 - mean_k <s_k/|s_k|, t_k/|t_k|>)   (clipself.py:42-47), one fused fwd + one fused bwd kernel."""

    @staticmethod
    def forward(ctx, student, teacher, ops, weight):
        K, E = student.shape
        stats = ops.empty((K, 3), torch.float32)
        loss = ops.empty((1,), torch.float32)
        student, teacher = student.contiguous(), teacher.contiguous().float()
        ops.cosine_loss_fwd(student, teacher, stats, loss, weight)
        ctx.save_for_backward(student, teacher, stats)
        ctx.ops, ctx.weight = ops, weight
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        student, teacher, stats = ctx.saved_tensors
        d = ctx.ops.empty(tuple(student.shape), torch.float32)
        ctx.ops.cosine_loss_bwd(student, teacher, stats, d, ctx.weight, 1.0, grad_out.contiguous().float().reshape(1))
        return d, None, None, None


def cosine_distill_loss(student, teacher, ops, weight=1.0):
    return _CosineDistillFn.apply(student, teacher, ops, float(weight))


class CLIPSelf:
    def __call__(self, batch, model, dist_model, loss, device, cast_dtype, distributed, args):
        if distributed:
            model = model.module
            dist_model = dist_model.module
        images, normed_boxes, image_crops = batch       # note texts are not paired with images

        images = images.to(device=device, dtype=cast_dtype, non_blocking=True)
        normed_boxes = normed_boxes.to(device=device, dtype=torch.float32, non_blocking=True)
        image_crops = image_crops.to(device=device, dtype=cast_dtype, non_blocking=True)

        if getattr(args, "multiscale", False):
            side = images.shape[2]
            assert side == images.shape[3]
            choices = {1024: [320, 640, 896, 1024], 896: [336, 448, 672, 896]}.get(side)
            if choices is None:
                raise NotImplementedError
            tar = random.choice(choices)
            images = F.interpolate(images, size=(tar, tar), mode="bilinear")

        valid = normed_boxes[..., -1] > 0.5                                   # [B, max_boxes]
        if bool(valid.all()):                                                 # dense batch: no gather needed
            B, k = valid.shape
            idx = torch.arange(B, device=normed_boxes.device, dtype=torch.float32).repeat_interleave(k)[:, None]
            rois = torch.cat([idx, normed_boxes[..., :4].reshape(B * k, 4)], dim=1)
            crops = image_crops.reshape(B * k, *image_crops.shape[2:])
        else:
            bidx = torch.nonzero(valid)[:, 0].to(torch.float32)[:, None]
            rois = torch.cat([bidx, normed_boxes[valid][:, :4]], dim=1)
            crops = image_crops[valid]

        with torch.no_grad():
            teacher_crop_features = dist_model.encode_image(crops, normalize=False)
        student_roi_features = model.encode_pseudo_boxes(images, rois, normalize=False,
                                                         extract_type=getattr(args, "extract_type", "v2"))

        loss_cosine = cosine_distill_loss(student_roi_features, teacher_crop_features, model.visual.engine.ops,
                                          getattr(args, "cosine_weight", 1.0))
        losses = dict(loss_cosine=loss_cosine)
        return losses, len(images), model.logit_scale.exp()
