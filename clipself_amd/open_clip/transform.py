"""Image preprocessing with the reference's transform API (src/open_clip/transform.py), executed by the HIP crop kernel.

    det_image_transform(size, is_train=False, mean, std)                    -> ResizeLongest  (longest side -> size, bicubic, zero padding
                                                                               right/bottom, transform.py:136-191) + RGB + ToTensor + Normalize
    image_transform(size, is_train=False, mean, std, resize_longest_max=True) -> ResizeMaxSize (longest side -> size, bicubic, centred zero
                                                                               padding, transform.py:26-49) + RGB + ToTensor + Normalize

are what `create_model_and_transforms` hands to the distillation datasets as `[det transform, crop transform]` (factory.py:312-350).
Both are callables `PIL.Image | HxWx3 uint8 array/tensor -> float32 tensor [3, size, size]`; the arithmetic is cs_crop_resize_u8
(csrc/preprocess.hip), which restates Pillow's fixed-point bicubic resampling bit for bit, so outputs equal the reference's
Pillow/torchvision pipeline exactly (tests: CPU through the Pillow-backed reference op, `-m gpu` through the kernel against
oracle/pil_crops_ref.py).  The training loaders (training/data.py) call the same kernel on all boxes of an image at once; these
per-image callables exist so that code written against the reference's dataset contract runs unchanged.
Train-time augmentation (RandomResizedCrop, `is_train=True`) is not part of the distillation path (the reference itself raises for
the det transform) and is not built."""
from typing import Optional, Sequence

import numpy as np
import torch
import torch.utils.data

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)


def _triple(v, default):
    v = v or default
    return tuple(v) if isinstance(v, (list, tuple)) else (v,) * 3


class ResizeLongestNormalize:
    """One image -> [3, size, size]: resize the longest side to `size` (bicubic, Pillow arithmetic), zero-pad to a square (centred for the
    crop transform, right/bottom for the det transform), /255, normalise.  `ops` = the kernel face (HipOps by default, created on first
    use: needs a ROCm device); the result stays on the device the kernel ran on unless `output_device` says otherwise.
    Input contract: a PIL image, or an HxWx3 uint8 array / tensor (channels LAST -- the reference's ResizeLongest takes CHW tensors).
    The kernel runs in the calling process: not from DataLoader workers (it raises there).  Down-sampling is limited by the kernel's
    filter-tap budget, max(H, W) * 4 + 1 <= 64 * size (about 16x at size 224); larger images raise instead of being resized."""

    def __init__(self, size: int, pad_center: bool, mean=None, std=None, ops=None, output_device=None):
        if isinstance(size, (list, tuple)):
            if size[0] != size[1]:
                raise NotImplementedError("non-square transform sizes are not used by the CLIPSelf recipes")
            size = size[0]
        self.size, self.pad_center = int(size), bool(pad_center)
        self.mean, self.std = _triple(mean, OPENAI_DATASET_MEAN), _triple(std, OPENAI_DATASET_STD)
        self._ops, self.output_device = ops, output_device

    @property
    def ops(self):
        if self._ops is None:
            from ..hip import HipOps
            self._ops = HipOps()
        return self._ops

    def _as_u8(self, img):
        if isinstance(img, torch.Tensor):
            t = img
        elif isinstance(img, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(img))
        else:                                        # PIL.Image: the reference's `_convert_to_rgb`
            t = torch.from_numpy(np.asarray(img.convert("RGB")).copy())
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
            raise TypeError(f"expected an RGB image as HxWx3 uint8, got {tuple(t.shape)} {t.dtype}")
        return t.contiguous()

    def __call__(self, img):
        ops = self.ops
        u8 = self._as_u8(img)
        if getattr(ops, "name", "") == "hip":
            if torch.utils.data.get_worker_info() is not None:
                raise RuntimeError(
                    "this transform runs the crop / resize kernel on the GPU and cannot be called from a DataLoader worker process (a forked "
                    "worker cannot re-initialise the device): use num_workers=0 with it, or the batch loaders of clipself_amd.training.data "
                    "(GpuGridDistillLoader / GpuProposalDistillLoader), which decode on host threads and crop on the GPU")
            u8 = u8.cuda(non_blocking=True)
        H, W = u8.shape[0], u8.shape[1]
        box = torch.tensor([[0.0, 0.0, float(W), float(H)]], device=u8.device)
        out = ops.crop_resize(u8, box, self.size, pad_center=self.pad_center, mean=self.mean, std=self.std)[0]
        return out if self.output_device is None else out.to(self.output_device)

    def __repr__(self):
        pad = "centre" if self.pad_center else "right/bottom"
        return f"{type(self).__name__}(size={self.size}, pad={pad}, mean={self.mean}, std={self.std})"


class _TrainAugmentationNotBuilt:
    def __init__(self, what):
        self.what = what

    def __call__(self, *a, **k):
        raise NotImplementedError(f"{self.what}: train-time augmentation (RandomResizedCrop) is not on the distillation path and not built")


def det_image_transform(image_size, is_train: bool, mean: Optional[Sequence[float]] = None, std: Optional[Sequence[float]] = None,
                        fill_color: int = 0, aug_cfg=None, ops=None):
    if is_train:
        raise NotImplementedError              # as the reference (transform.py:157-158)
    if fill_color != 0:
        raise NotImplementedError("only zero padding exists")
    return ResizeLongestNormalize(image_size, pad_center=False, mean=mean, std=std, ops=ops)


def image_transform(image_size, is_train: bool, mean: Optional[Sequence[float]] = None, std: Optional[Sequence[float]] = None,
                    resize_longest_max: bool = False, fill_color: int = 0, aug_cfg=None, ops=None):
    if is_train:
        return _TrainAugmentationNotBuilt("image_transform(is_train=True)")
    if not resize_longest_max:
        raise NotImplementedError("Resize(shorter side) + CenterCrop is not used by the CLIPSelf recipes (resize_longest_max=True is)")
    if fill_color != 0:
        raise NotImplementedError("only zero padding exists")
    return ResizeLongestNormalize(image_size, pad_center=True, mean=mean, std=std, ops=ops)
