"""Batch contract of the distillation datasets + a synthetic, device-resident producer.

The reference's COCO/LVIS pipeline (src/training/data.py: GridDistillDataset :135-281, ProposalDistillDataset
:30-132; PIL + pycocotools in DataLoader workers) is outside the MI355X hot path; what the step
consumes is its *output contract* (data.py:247-281):
    images [B,3,S,S], normed_boxes [B,max_boxes,5] = (x0,y0,x1,y1 in [0,1], valid in {0,1}), image_crops [B,max_boxes,3,Sc,Sc]
`SyntheticDistillData` emits exactly that (SURVEY.md §8 M2 recipe) straight into HBM, shaped like the reference's
DataInfo/DataLoader pair (`.dataloader.num_batches`, `.num_samples`, `.set_epoch`)."""
from dataclasses import dataclass

import torch

from ..init import synthetic_batch


class _SyntheticLoader:
    def __init__(self, steps, batch_size, boxes, image_size, crop_size, device, rank, world, seed=1234, valid_prob=1.0, resident=True):
        self.num_batches, self.num_samples = steps, steps * batch_size * world
        self.args = (batch_size, boxes, image_size, crop_size)
        self.device, self.rank, self.seed, self.valid_prob, self.resident = device, rank, seed, valid_prob, resident
        self.epoch = 0
        self._cache = None
        self.region_clip = False

    def _make(self, i):
        b = synthetic_batch(*self.args, seed=self.seed + 1000 * self.epoch + i, rank=self.rank * 7919, valid_prob=self.valid_prob)
        if self.region_clip:                      # COCORegionCLIPDataset contract (data.py:390-459): (images, boxes[B,k,6] = xyxy,label,valid)
            g = torch.Generator().manual_seed(self.seed + i)
            labels = torch.randint(0, 4764, (*b[1].shape[:2], 1), generator=g).float()
            return b[0].to(self.device), torch.cat([b[1][..., :4], labels, b[1][..., 4:5]], dim=-1).to(self.device)
        return tuple(t.to(self.device) for t in b)

    def __len__(self):
        return self.num_batches

    def __iter__(self):
        for i in range(self.num_batches):
            if self.resident:                      # throughput mode: one batch generated once, re-used (already in HBM)
                if self._cache is None:
                    self._cache = self._make(0)
                yield self._cache
            else:
                yield self._make(i)


@dataclass
class DataInfo:
    dataloader: _SyntheticLoader

    def set_epoch(self, epoch):
        self.dataloader.epoch = epoch


class SyntheticPanopticVal:
    """Batches shaped like the panoptic validation set (src/training/data.py:331-387) plus `.embeddings`: boxes from the synthetic
    recipe, a rectangular mask per box at feature-map resolution, a class per box and a things/stuff flag per class."""

    def __init__(self, steps, batch_size, boxes, image_size, crop_size, grid, embed_dim, num_classes=32, seed=4321):
        g = torch.Generator().manual_seed(seed)
        self.embeddings = torch.randn(num_classes, embed_dim, generator=g).numpy()
        self.batches = []
        for i in range(steps):
            images, nb, crops = synthetic_batch(batch_size, boxes, image_size, crop_size, seed=seed + 1 + i, valid_prob=0.8)
            labels = torch.randint(0, num_classes, nb.shape[:2], generator=g).float()
            area = (nb[..., 2] - nb[..., 0]) * (nb[..., 3] - nb[..., 1]) * image_size * image_size
            info = torch.stack([labels, nb[..., 4], area, (labels % 3 != 0).float()], dim=-1)
            bboxes = torch.cat([nb[..., :4], info], dim=-1)                                        # [B,K,8]
            ys = (torch.arange(grid).float() + 0.5) / grid
            inside_y = (ys[None, None, :] >= nb[..., 1:2]) & (ys[None, None, :] <= nb[..., 3:4])    # [B,K,g]
            inside_x = (ys[None, None, :] >= nb[..., 0:1]) & (ys[None, None, :] <= nb[..., 2:3])
            masks = (inside_y[..., :, None] & inside_x[..., None, :]).float()                     # [B,K,g,g]
            masks[..., grid // 2, grid // 2] = 1.0                                                  # never empty
            self.batches.append((images, bboxes, crops, masks, crops.clone()))

    def __len__(self):
        return len(self.batches)


class _ValLoader:
    def __init__(self, dataset):
        self.dataset = dataset
        self.num_batches, self.num_samples = len(dataset), sum(len(b[0]) for b in dataset.batches)

    def __iter__(self):
        return iter(self.dataset.batches)


def get_data(args, preprocess_fns=None, epoch=0, tokenizer=None):
    if args.train_data != "synthetic":
        raise NotImplementedError(
            "only --train-data synthetic is wired in this build: the COCO/LVIS PIL pipeline is host-side and out of "
            "scope of the MI355X hot path (SURVEY.md §8 N3); any iterable yielding the batch contract can be plugged in")
    size = args.synthetic_image_size or args.det_image_size
    loader = _SyntheticLoader(args.synthetic_steps, args.batch_size, args.max_boxes, size, args.input_size,
                              args.device, args.rank, args.world_size, seed=1234 + args.seed,
                              valid_prob=0.7 if args.dataset_type == "proposals_distill" else 1.0, resident=False)
    loader.region_clip = args.dataset_type == "region_clip"
    data = {"train": DataInfo(loader)}
    if getattr(args, "val_data", None) == "synthetic":
        cfg = args.tower_cfg
        val = SyntheticPanopticVal(2, min(args.batch_size, 4), min(args.max_boxes, 6), size, args.input_size, size // cfg.patch_size,
                                   cfg.embed_dim, seed=4321 + args.seed)
        data["val"] = DataInfo(_ValLoader(val))
    return data
