"""RegionCLIP method -- same call contract as the reference's src/training/region_clip.py:19-67:

    losses, batch_size, temp = RegionCLIP(args)(batch, model, dist_model, loss, device, cast_dtype, distributed, args)

batch = (images [B,3,S,S], boxes [B,max_boxes,6] = (x0,y0,x1,y1 in [0,1], label, valid)).  Student RoI features
(L2-normalised) are scored against a frozen bank of noun embeddings; the loss is binary cross-entropy over a
*federated* subset of <= max(100, #labels present) noun columns (get_fed_loss_inds, region_clip.py:7-16).

Here the column subset is drawn first (host-side torch: unique + multinomial, exactly the reference's calls), so only
[K, n_sampled] logits are ever formed: one MFMA GEMM against the gathered embeddings, one fused BCE kernel, and
the mirrored pair in backward.  Mathematically identical to slicing the full [K, 4764] logit matrix.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def get_fed_loss_inds(gt_classes, num_sample_cats, C):
    appeared = torch.unique(gt_classes)
    prob = appeared.new_ones(C).float()
    if len(appeared) < num_sample_cats:
        prob[appeared] = 0
        more = torch.multinomial(prob, num_sample_cats - len(appeared), replacement=False)
        appeared = torch.cat([appeared, more])
    return appeared


class _FedBCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, ops, nouns_sel, tgt, temp, weight):
        K, E = feats.shape
        ns = nouns_sel.shape[0]
        nsp = (ns + 63) // 64 * 64                                   # contraction padding of the backward GEMM
        fb = ops.empty((K, E), torch.bfloat16)
        ops.cast_f32_bf16(feats.contiguous(), fb)
        nb = ops.zeros((nsp, E), torch.bfloat16)
        nb[:ns] = nouns_sel.to(torch.bfloat16)
        logits = ops.empty((K, nsp), torch.float32)
        ops.gemm_nt(fb, nb, logits, epi=1)
        rowloss, loss = ops.empty((K,), torch.float32), ops.empty((1,), torch.float32)
        ops.fed_bce_fwd(logits, tgt, rowloss, loss, ns, temp, weight)
        ctx.save_for_backward(logits, tgt, nb)
        ctx.ops, ctx.ns, ctx.temp, ctx.weight, ctx.E = ops, ns, temp, weight, E
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        logits, tgt, nb = ctx.saved_tensors
        ops, K, nsp = ctx.ops, logits.shape[0], logits.shape[1]
        dz = ops.empty((K, nsp), torch.bfloat16)
        ops.fed_bce_bwd(logits, tgt, dz, ctx.ns, ctx.temp, ctx.weight, grad_out.contiguous().float().reshape(1))
        nbt = ops.empty((ctx.E, nsp), torch.bfloat16)
        ops.transpose_bf16(nb, nbt)
        dfeats = ops.empty((K, ctx.E), torch.float32)
        ops.gemm_nt(dz, nbt, dfeats, epi=1)
        return dfeats, None, None, None, None, None


class RegionCLIP(nn.Module):
    def __init__(self, args, noun_embeddings=None):
        super().__init__()
        if noun_embeddings is None:
            noun_embeddings = torch.from_numpy(np.load(args.train_embed_path))
        self.register_buffer("noun_embeddings", F.normalize(noun_embeddings.float(), dim=-1))
        self.place_holder = nn.Parameter(torch.ones(1))

    def __call__(self, batch, model, dist_model, loss, device, cast_dtype, distributed, args):
        if distributed:
            model = model.module
        images, boxes = batch
        images = images.to(device=device, dtype=cast_dtype, non_blocking=True)
        boxes = boxes.to(device=device, non_blocking=True).float()
        valid = boxes[..., -1] > 0.5
        bidx = torch.nonzero(valid)[:, 0].to(torch.float32)[:, None]
        sel = boxes[valid]
        rois = torch.cat([bidx, sel[:, :4]], dim=1)
        labels = sel[:, 4].long()
        box_features = model.encode_pseudo_boxes(images, rois, normalize=True, extract_type=getattr(args, "extract_type", "v2"))
        temp = model.logit_scale.exp().detach()
        # the kernels take the temperature as a launch argument: read it back only when logit_scale was written in place (it receives no
        # gradient in this method -- `.detach()` above -- and train_step's per-step clamp skips a parameter without a gradient, so that is
        # once per run; load_state_dict / copy_ bump the version, writes through `.data` must be followed by `_temp_cache = None`)
        ver = (id(model.logit_scale), model.logit_scale.data_ptr(), model.logit_scale._version)
        if getattr(self, "_temp_cache", (None, None))[0] != ver:
            self._temp_cache = (ver, float(temp))
        temp_f = self._temp_cache[1]
        nouns = self.noun_embeddings.to(box_features.device)
        appeared = get_fed_loss_inds(labels, 100, nouns.shape[0])
        # position of every box's label inside the sampled column set
        pos = torch.full((nouns.shape[0],), -1, dtype=torch.int32, device=labels.device)
        pos[appeared] = torch.arange(len(appeared), dtype=torch.int32, device=labels.device)
        loss_cls = _FedBCEFn.apply(box_features, model.visual.engine.ops, nouns[appeared], pos[labels].contiguous(), temp_f,
                                   float(getattr(args, "contrast_weight", 1.0)))
        return dict(loss_contrast=loss_cls), len(images), temp
