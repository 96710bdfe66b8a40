"""CLIPSelf distillation method -- same call contract as the reference's src/training/clipself.py:6-49:

    losses, batch_size, logit_scale = CLIPSelf()(batch, model, dist_model, loss, device, cast_dtype, distributed, args)

batch = (images [B,3,S,S], normed_boxes [B,max_boxes,5] = (x0,y0,x1,y1 in [0,1], valid), image_crops [B,max_boxes,3,Sc,Sc]).
The teacher forward, the student dense forward, RoIAlign, both L2 normalisations, the cosine loss and the whole
backward run as HIP kernels; the tensors handed back (`loss_cosine`) are ordinary autograd leaves of that path.
"""
import os
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


def mark_all_valid(normed_boxes, flag=None):
    """Producers of device-resident batches record whether every box slot is valid (they build the boxes on the host and know), so the
    step does not have to read the validity column back from the GPU.  Returns the tensor."""
    normed_boxes._cs_all_valid = bool((normed_boxes[..., -1] > 0.5).all()) if flag is None else bool(flag)
    return normed_boxes


def _known_all_valid(normed_boxes):
    if normed_boxes.device.type == "cpu":
        return bool((normed_boxes[..., -1] > 0.5).all())                      # host tensor: free to look at
    return getattr(normed_boxes, "_cs_all_valid", None)


# compute units the prefetched teacher pass leaves to the student's step on one GPU (0 = shared pool); profiles/r05_a_partition_sweep.md
DEFAULT_PARTITION_CUS = "0"


class CLIPSelf:
    """Besides the reference's call contract, the method can run the frozen teacher one batch ahead: `prefetch_teacher(next_batch,
    ...)` launches the teacher forward of the NEXT batch on a high-priority side stream, where it overlaps the student's backward,
    gradient all-reduce and AdamW of the current batch and the student forward of the next one (the teacher never changes, so its
    features do not depend on the student's update).  `__call__` picks the prefetched features up when it is handed that batch and
    otherwise computes them inline, exactly like the reference."""

    def __init__(self, partition_cus=None, partition_mask=None, partition_cap=None):
        self._pending = None           # (image_crops tensor of the prefetched batch, features, side stream)
        self._side = None
        # Static CU partition between the towers while a prefetched teacher pass is in flight: the teacher's persistent GEMMs leave
        # `partition_cus` compute units free and the student's are capped at that many workgroups, so the student's kernels stop queueing
        # behind 256-workgroup teacher launches (and the teacher's behind the student's).  0 = one shared pool, as before.
        # CLIPSELF_PARTITION_CUS / CLIPSELF_PARTITION_MASK=1 (queues with hardware CU masks: cs_stream_create_cu_mask).
        self.partition_cus = int(os.environ.get("CLIPSELF_PARTITION_CUS", DEFAULT_PARTITION_CUS)) if partition_cus is None else int(partition_cus)
        self.partition_mask = (os.environ.get("CLIPSELF_PARTITION_MASK", "0") == "1") if partition_mask is None else bool(partition_mask)
        # workgroups of the student's persistent GEMMs meanwhile (default: its share; CLIPSELF_PARTITION_CAP)
        cap = os.environ.get("CLIPSELF_PARTITION_CAP") if partition_cap is None else partition_cap
        self.partition_cap = int(cap) if cap not in (None, "") else self.partition_cus
        # what can be checked without a device is checked here, the rest (against the device's CU count) when the towers' ops are first seen
        if self.partition_cus < 0 or (self.partition_cus and (self.partition_cus < 8 or self.partition_cap < 8)):
            raise ValueError(f"CLIPSELF_PARTITION_CUS / _CAP = {self.partition_cus} / {self.partition_cap}: a tower's share is 0 (off) or at least 8 compute units")
        if self.partition_mask and self.partition_cus % 8:
            raise ValueError(f"CLIPSELF_PARTITION_MASK=1 needs a share that is a multiple of 8 compute units (one slice per XCD), got {self.partition_cus}")
        self._partition_checked = self.partition_cus == 0
        self._student_ops = None
        self._student_stream = None

    def _check_partition(self, ops):
        """The device-dependent half of the constructor's checks, once, before the first reservation is made (not an assert in
        the middle of a step): both towers keep >= 8 workgroups and the reservation fits the 8-bit field of the GEMM flags."""
        if self._partition_checked or ops is None or not hasattr(ops, "num_compute_units"):
            return
        n = ops.num_compute_units()
        if not (8 <= self.partition_cus <= min(n - 8, 255)) or not (8 <= self.partition_cap <= n) or n - self.partition_cap > 255:
            raise ValueError(f"tower partition {self.partition_cus} CUs (cap {self.partition_cap}) does not fit a device of {n} compute units: "
                             f"share in [8, {min(n - 8, 255)}], cap in [{max(8, n - 255)}, {n}]")
        self._partition_checked = True

    def _cap_student(self, on: bool):
        ops = self._student_ops
        if ops is not None and hasattr(ops, "cap_compute_units"):
            if on:
                self._check_partition(ops)
            ops.cap_compute_units(self.partition_cap if on else 0)

    def student_stream(self, ops):
        """With partition_mask: the queue the student's step should run on (CU mask = the student's share); None otherwise."""
        if not (self.partition_mask and self.partition_cus):
            return None
        if self._student_stream is None:
            n = ops.num_compute_units()
            self._student_stream = ops.stream_create_cu_mask(n - self.partition_cus, self.partition_cus)
        return self._student_stream

    @staticmethod
    def _valid_crops(normed_boxes, image_crops, all_valid=None):
        """all_valid: what the producer of the batch already knows about the validity column (`mark_all_valid`); None = look at the
        tensor, which costs a host sync when it lives on the GPU (the reference's boolean indexing pays the same sync)."""
        valid = normed_boxes[..., -1] > 0.5                                   # [B, max_boxes]
        if all_valid is None:
            all_valid = bool(valid.all())
        if all_valid:                                                         # dense batch: no gather needed
            return valid, True, image_crops.reshape(-1, *image_crops.shape[2:])
        return valid, False, image_crops[valid]

    def prefetch_teacher(self, batch, dist_model, device, cast_dtype, distributed):
        device = torch.device(device)
        if device.type != "cuda":
            return
        window = (0, 0)
        if distributed:
            # the first blocks of a prefetched pass run beside the student's gradient buckets: their GEMMs leave RCCL's CUs free
            window = getattr(dist_model, "prefetch_window", (0, 0))
            dist_model = dist_model.module
        _, normed_boxes, image_crops = batch
        boxes_d = normed_boxes.to(device=device, dtype=torch.float32, non_blocking=True)
        crops_d = image_crops.to(device=device, dtype=cast_dtype, non_blocking=True)
        _, _, crops = self._valid_crops(boxes_d, crops_d, _known_all_valid(normed_boxes))
        main = torch.cuda.current_stream(device)
        eng = getattr(getattr(dist_model, "visual", None), "engine", None)
        tops = getattr(eng, "ops", None)
        share = self.partition_cus if hasattr(tops, "share_compute_units") else 0
        if share:
            self._check_partition(tops)
        if self._side is None:
            if share and self.partition_mask:
                self._side = tops.stream_create_cu_mask(0, tops.num_compute_units() - share)      # the teacher's CUs, enforced by the dispatcher
            else:
                # single GPU: the teacher is the long pole, let it win CUs.  Data parallel: normal priority, so that RCCL's kernels
                # (launched at default priority) are not starved behind 3 ms persistent GEMMs
                prio = os.environ.get("CLIPSELF_TEACHER_STREAM_PRIORITY")
                self._side = torch.cuda.Stream(device=device, priority=int(prio) if prio is not None else (0 if distributed else -1))
        side = self._side
        side.wait_stream(main)
        with torch.cuda.stream(side), torch.no_grad():
            if eng is not None:
                eng.rccl_window = window
            if share:
                tops.share_compute_units(share)        # every persistent GEMM of this pass: grid = CUs - share (- RCCL's inside the window)
            try:
                feats = dist_model.encode_image(crops, normalize=False)
            finally:
                if eng is not None:
                    eng.rccl_window = (0, 0)
                if share:
                    tops.share_compute_units(0)
        crops.record_stream(side)
        self._pending = (image_crops, feats, side)
        self._cap_student(bool(share))                 # what the student queues from here on runs beside this pass

    def _teacher_features(self, image_crops, crops, dist_model):
        pend, self._pending = self._pending, None
        if pend is not None and pend[0] is image_crops:
            _, feats, side = pend
            main = torch.cuda.current_stream(feats.device)
            main.wait_stream(side)
            feats.record_stream(main)
            self._cap_student(False)                   # that pass is over once the stream gets here; the next prefetch caps again
            return feats
        self._cap_student(False)
        with torch.no_grad():
            return dist_model.encode_image(crops, normalize=False)

    def __call__(self, batch, model, dist_model, loss, device, cast_dtype, distributed, args):
        if distributed:
            model = model.module
            dist_model = dist_model.module
        images, normed_boxes, image_crops_in = batch    # note texts are not paired with images
        self._student_ops = getattr(getattr(getattr(model, "visual", None), "engine", None), "ops", None)
        self._cap_student(self._pending is not None and self.partition_cus > 0)      # a prefetched teacher pass is running beside this forward

        images = images.to(device=device, dtype=cast_dtype, non_blocking=True)
        normed_boxes = normed_boxes.to(device=device, dtype=torch.float32, non_blocking=True)
        image_crops = image_crops_in.to(device=device, dtype=cast_dtype, non_blocking=True)

        if getattr(args, "multiscale", False):
            side = images.shape[2]
            assert side == images.shape[3]
            choices = {1024: [320, 640, 896, 1024], 896: [336, 448, 672, 896]}.get(side)
            if choices is None:
                raise NotImplementedError
            tar = random.choice(choices)
            images = model.visual.engine.ops.resize_bilinear(images.float(), tar)      # F.interpolate(..., mode="bilinear") as one kernel

        valid, dense, crops = self._valid_crops(normed_boxes, image_crops, _known_all_valid(batch[1]))
        if dense:
            B, k = valid.shape
            idx = torch.arange(B, device=normed_boxes.device, dtype=torch.float32).repeat_interleave(k)[:, None]
            rois = torch.cat([idx, normed_boxes[..., :4].reshape(B * k, 4)], dim=1)
        else:
            bidx = torch.nonzero(valid)[:, 0].to(torch.float32)[:, None]
            rois = torch.cat([bidx, normed_boxes[valid][:, :4]], dim=1)

        # student first: when the teacher features were prefetched they are still being produced on the side stream
        student_roi_features = model.encode_pseudo_boxes(images, rois, normalize=False,
                                                         extract_type=getattr(args, "extract_type", "v2"))
        teacher_crop_features = self._teacher_features(image_crops_in, crops, dist_model)

        loss_cosine = cosine_distill_loss(student_roi_features, teacher_crop_features, model.visual.engine.ops,
                                          getattr(args, "cosine_weight", 1.0))
        losses = dict(loss_cosine=loss_cosine)
        return losses, len(images), model.logit_scale.exp()
