"""Batch contract of the distillation datasets + a synthetic, device-resident producer.

The reference's COCO/LVIS pipeline (src/training/data.py: GridDistillDataset :135-281, ProposalDistillDataset
:30-132; PIL + pycocotools in DataLoader workers) is outside the MI355X hot path; what the step
consumes is its *output contract* (data.py:247-281):
    images [B,3,S,S], normed_boxes [B,max_boxes,5] = (x0,y0,x1,y1 in [0,1], valid in {0,1}), image_crops [B,max_boxes,3,Sc,Sc]
`SyntheticDistillData` emits exactly that (SURVEY.md §8 M2 recipe) straight into HBM, shaped like the reference's
DataInfo/DataLoader pair (`.dataloader.num_batches`, `.num_samples`, `.set_epoch`)."""
import os
import random
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
        from .clipself import mark_all_valid
        flag = bool((b[1][..., -1] > 0.5).all())          # decided on the host copy: the step never reads the validity column back
        out = tuple(t.to(self.device) for t in b)
        mark_all_valid(out[1], flag)
        return out

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
    dataloader: object

    def set_epoch(self, epoch):
        """Epoch-dependent shuffles AND per-sample draws (grid choice, box order): a run resumed at epoch e continues with epoch e's
        draws instead of replaying epoch 0's (epoch 0 keeps the loader's construction-time stream)."""
        dl = self.dataloader
        dl.epoch = epoch
        if epoch and hasattr(dl, "rng") and hasattr(dl, "seed"):
            dl.rng = random.Random(dl.seed + 7919 * epoch)


def _read_ahead(images, order, num_batches, batch_size):
    """Tell a lazily decoding image source (training/coco_source.py:DecodedImages) the order this epoch will ask in; plain lists ignore it."""
    hint = getattr(images, "hint", None)
    if hint is not None:
        hint([order[(b * batch_size + j) % len(order)] for b in range(num_batches) for j in range(batch_size)])


def grid_choices(max_split):
    """(M, N) grids the reference samples from (GridDistillDataset._init_choices, data.py:200-206)."""
    return [(m, n) for m in range(1, max_split + 1) for n in range((m + 1) // 2, min(m * 2 + 1, max_split + 1))]


def grid_boxes(M, N):
    """Cells of an M x N grid as (x0, y0, x1, y1) in [0,1], row-major (data.py:211-224)."""
    xs, ys = torch.linspace(0, 1, N + 1), torch.linspace(0, 1, M + 1)
    gx, gy = torch.meshgrid(xs, ys, indexing="xy")
    return torch.cat([torch.stack([gx[:M, :N], gy[:M, :N]], -1), torch.stack([gx[1:, 1:], gy[1:, 1:]], -1)], -1).view(-1, 4)


class GpuGridDistillLoader:
    """GridDistillDataset (src/training/data.py:135-281) with the pixel work on the GPU: decoded RGB images
    (uint8 HWC, resident in HBM) -> the batch contract.  Per image, as the reference: one (M, N) grid drawn from `grid_choices`, its
    cells shuffled and cut to max_boxes, optional enlargement by crop_scale clipped to the image, every cell cropped and resized
    (ResizeMaxSize, centred padding) to crop_size, the image itself resized (ResizeLongest, right/bottom padding) to det_size, boxes
    rescaled to the padded square.  Crops and the det image come from cs_crop_resize_u8, bit-identical to the Pillow path; the
    random draws use Python's `random` exactly where the reference does."""

    def __init__(self, images_u8, ops, batch_size, max_boxes, det_size, crop_size, max_split=6, crop_scale=1.0, steps=None, seed=0):
        self.images, self.ops = images_u8, ops
        self.batch_size, self.max_boxes, self.det_size, self.crop_size = batch_size, max_boxes, det_size, crop_size
        self.crop_scale = crop_scale
        self.choices = grid_choices(max_split)
        self.templates = {c: grid_boxes(*c) for c in self.choices}
        self.num_batches = steps if steps is not None else len(images_u8) // batch_size
        self.num_samples = self.num_batches * batch_size
        self.seed, self.rng = seed, random.Random(seed)
        self.epoch = 0

    def __len__(self):
        return self.num_batches

    def sample(self, img):
        """-> (det image [3,S,S], boxes [max_boxes,5], crops [max_boxes,3,Sc,Sc]) for one decoded image; also returns the pixel
        boxes actually cropped (for the tests)."""
        H, W = img.shape[0], img.shape[1]
        dev = img.device
        tmpl = self.templates[self.rng.choice(self.choices)]
        idx = list(range(len(tmpl)))
        self.rng.shuffle(idx)
        idx = idx[:self.max_boxes]
        px = tmpl[idx] * torch.tensor([W, H, W, H], dtype=torch.float32)
        crop_px = px.clone()
        if self.crop_scale > 1.0:                                    # data.py:236-241
            bw, bh = px[:, 2] - px[:, 0], px[:, 3] - px[:, 1]
            cx, cy = (px[:, 2] + px[:, 0]) / 2, (px[:, 3] + px[:, 1]) / 2
            d = 0.5 * self.crop_scale
            crop_px = torch.stack([(cx - bw * d).clamp(min=0), (cy - bh * d).clamp(min=0), (cx + bw * d).clamp(max=W),
                                   (cy + bh * d).clamp(max=H)], -1)
        k = len(idx)
        crops = torch.zeros(self.max_boxes, 3, self.crop_size, self.crop_size, device=dev)
        self.ops.crop_resize(img, crop_px.to(dev), self.crop_size, pad_center=True, out=crops[:k])
        whole = torch.tensor([[0.0, 0.0, float(W), float(H)]], device=dev)
        det = self.ops.crop_resize(img, whole, self.det_size, pad_center=False)[0]
        scale = min(self.det_size / H, self.det_size / W)            # get_scale (transform.py:194-207) of ResizeLongest's square output
        boxes = torch.zeros(self.max_boxes, 5)
        boxes[:k, :4] = px * scale / self.det_size
        boxes[:k, 4] = 1.0
        return det, boxes.to(dev), crops, crop_px

    def __iter__(self):
        order = list(range(len(self.images)))
        random.Random(1000 + self.epoch).shuffle(order)
        _read_ahead(self.images, order, self.num_batches, self.batch_size)
        for b in range(self.num_batches):
            parts = [self.sample(self.images[order[(b * self.batch_size + j) % len(order)]])[:3] for j in range(self.batch_size)]
            yield tuple(torch.stack([p[i] for p in parts]) for i in range(3))


class GpuProposalDistillLoader:
    """ProposalDistillDataset (src/training/data.py:30-132) with the pixel work on the GPU: per decoded image and its annotation boxes
    (x, y, w, h in pixels): annotations shuffled, the first max_anns considered, those outside [min_size^2, max_size^2] in area left as
    empty slots, student box = the annotation, teacher crop = the box enlarged 1.5x about its centre and clipped to the image
    (data.py:111-118), fallback to the top-left quarter image when nothing is valid (:122-124), boxes rescaled to the padded square."""

    def __init__(self, images_u8, annotations, ops, batch_size, det_size, crop_size, min_size=8, max_size=1024, max_anns=20, steps=None, seed=0):
        self.images, self.anns, self.ops = images_u8, annotations, ops
        self.batch_size, self.det_size, self.crop_size = batch_size, det_size, crop_size
        self.min_size, self.max_size, self.max_anns = min_size, max_size, max_anns
        self.num_batches = steps if steps is not None else len(images_u8) // batch_size
        self.num_samples = self.num_batches * batch_size
        self.seed, self.rng = seed, random.Random(seed)
        self.epoch = 0

    def __len__(self):
        return self.num_batches

    def sample(self, img, anns):
        H, W = img.shape[0], img.shape[1]
        dev = img.device
        order = list(range(len(anns)))
        self.rng.shuffle(order)
        boxes = torch.zeros(self.max_anns, 5)
        slots, crop_px = [], []
        for i, a in enumerate(order[:self.max_anns]):
            x, y, w, h = anns[a]
            if w * h < self.min_size ** 2 or w * h > self.max_size ** 2:
                continue
            cx, cy = x + w * 0.5, y + h * 0.5
            crop_px.append([max(cx - w * 0.75, 0), max(cy - h * 0.75, 0), min(cx + w * 0.75, W), min(cy + h * 0.75, H)])
            boxes[i] = torch.tensor([x, y, x + w, y + h, 1.0])
            slots.append(i)
        if not slots:                                                   # avoid an empty image
            boxes[0] = torch.tensor([0, 0, W / 4, H / 4, 1.0])
            crop_px, slots = [[0, 0, W // 4, H // 4]], [0]
        crop_px = torch.tensor(crop_px, dtype=torch.float32)
        crops = torch.zeros(self.max_anns, 3, self.crop_size, self.crop_size, device=dev)
        crops[slots] = self.ops.crop_resize(img, crop_px.to(dev), self.crop_size, pad_center=True)
        det = self.ops.crop_resize(img, torch.tensor([[0.0, 0.0, float(W), float(H)]], device=dev), self.det_size, pad_center=False)[0]
        boxes[:, :4] *= min(self.det_size / H, self.det_size / W) / self.det_size
        return det, boxes.to(dev), crops, crop_px, slots

    def __iter__(self):
        order = list(range(len(self.images)))
        random.Random(1000 + self.epoch).shuffle(order)
        _read_ahead(self.images, order, self.num_batches, self.batch_size)
        for b in range(self.num_batches):
            ids = [order[(b * self.batch_size + j) % len(order)] for j in range(self.batch_size)]
            parts = [self.sample(self.images[i], self.anns[i])[:3] for i in ids]
            yield tuple(torch.stack([p[i] for p in parts]) for i in range(3))


class GpuRegionClipLoader:
    """COCORegionCLIPDataset (src/training/data.py:390-459) with the pixel work on the GPU: only images that have annotations, per image the
    det transform (ResizeLongest, right/bottom padding) and up to max_anns = min(max annotations per image, 20) boxes as
    (x0, y0, x1, y1 in [0,1] of the padded square, class label = rank of the category id, valid) -- no crops, no teacher."""

    def __init__(self, images_u8, annotations, cat_id2label, ops, batch_size, det_size, max_anns=20, steps=None, seed=0):
        self.images, self.anns, self.cat_id2label, self.ops = images_u8, annotations, cat_id2label, ops
        self.batch_size, self.det_size, self.max_anns = batch_size, det_size, max_anns
        self.num_batches = steps if steps is not None else len(images_u8) // batch_size
        self.num_samples = self.num_batches * batch_size
        self.epoch = 0

    def __len__(self):
        return self.num_batches

    def sample(self, img, anns):
        H, W = img.shape[0], img.shape[1]
        dev = img.device
        boxes = torch.zeros(self.max_anns, 6)
        for i, a in enumerate(anns[:self.max_anns]):
            x, y, w, h = a["bbox"]
            boxes[i] = torch.tensor([x, y, x + w, y + h, float(self.cat_id2label[a["category_id"]]), 1.0])
        det = self.ops.crop_resize(img, torch.tensor([[0.0, 0.0, float(W), float(H)]], device=dev), self.det_size, pad_center=False)[0]
        boxes[:, :4] *= min(self.det_size / H, self.det_size / W) / self.det_size
        return det, boxes.to(dev)

    def __iter__(self):
        order = list(range(len(self.images)))
        random.Random(1000 + self.epoch).shuffle(order)
        _read_ahead(self.images, order, self.num_batches, self.batch_size)
        for b in range(self.num_batches):
            ids = [order[(b * self.batch_size + j) % len(order)] for j in range(self.batch_size)]
            parts = [self.sample(self.images[i], self.anns[i]) for i in ids]
            yield torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts])


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
        streaming = dataset.batches is dataset                  # CocoPanopticVal: one image per batch, produced on the fly
        self.num_batches, self.num_samples = len(dataset), len(dataset) if streaming else sum(len(b[0]) for b in dataset.batches)

    def __iter__(self):
        return iter(self.dataset.batches)


def coco_train_loader(args, ops=None):
    """`--train-data <instances|proposals>.json --train-image-root <dir>` as in the reference's scripts: the annotation file is indexed and the
    image files decoded on the host (training/coco_source.py), crops / det image / boxes are produced on the GPU by the loaders above."""
    from .coco_source import AnnotationBoxes, CocoIndex, DecodedImages, subset_ids
    if ops is None:
        from ..hip import HipOps
        ops = HipOps()
    index = CocoIndex(args.train_data)
    rank, world = getattr(args, "rank", 0), getattr(args, "world_size", 1)
    ratio = getattr(args, "train_ratio", 1.0) if args.dataset_type in ("grid_distill", "region_clip") else 1.0
    ids = subset_ids(index, ratio, rank, world, seed=args.seed, annotated_only=args.dataset_type == "region_clip")
    images = DecodedImages(index, args.train_image_root, args.device, image_ids=ids, workers=max(getattr(args, "workers", 1), 1) * 4,
                           seed=args.seed + rank)
    size, seed = args.det_image_size, 1234 + args.seed + 7919 * rank
    if args.dataset_type == "proposals_distill":
        return GpuProposalDistillLoader(images, AnnotationBoxes(images), ops, args.batch_size, size, args.input_size, min_size=args.min_size,
                                        max_size=args.max_size, seed=seed)
    if args.dataset_type == "grid_distill":
        return GpuGridDistillLoader(images, ops, args.batch_size, args.max_boxes, size, args.input_size, max_split=args.max_split,
                                    crop_scale=args.crop_scale, seed=seed)
    if args.dataset_type == "region_clip":
        return GpuRegionClipLoader(images, images.annotation_view(), index.cat_id2label, ops, args.batch_size, size,
                                   max_anns=min(max((len(v) for v in index.imgToAnns.values()), default=1), 20))
    raise NotImplementedError(f"--dataset-type {args.dataset_type}")


def coco_panoptic_val(args, ops=None):
    """`--val-data <panoptic json> --val-image-root <dir> --val-segm-root <dir> --embed-path <npy>` (get_coco_panoptic_dataset, data.py:457-497)."""
    from .coco_source import CocoPanopticVal
    if ops is None:
        from ..hip import HipOps
        ops = HipOps()
    return CocoPanopticVal(args.val_data, args.val_image_root, args.val_segm_root, args.embed_path, ops, args.device, args.det_image_size,
                           args.input_size, downsample_factor=args.downsample_factor, rank=getattr(args, "rank", 0) if args.distributed else 0,
                           world=getattr(args, "world_size", 1) if args.distributed else 1)


def _with_val(data, args):
    """Adds data['val']: `--val-data synthetic` (seeded panoptic-style batches) or a COCO-panoptic annotation file; anything else (e.g. the
    reference's default path when that file does not exist) means no evaluation."""
    val = getattr(args, "val_data", None)
    if val == "synthetic":
        cfg = args.tower_cfg
        size = getattr(args, "synthetic_image_size", None) or args.det_image_size
        ds = SyntheticPanopticVal(2, min(args.batch_size, 4), min(args.max_boxes, 6), size, args.input_size, size // cfg.patch_size,
                                  cfg.embed_dim, seed=4321 + args.seed)
        data["val"] = DataInfo(_ValLoader(ds))
    elif val and os.path.isfile(val):
        data["val"] = DataInfo(_ValLoader(coco_panoptic_val(args)))
    return data


def get_data(args, preprocess_fns=None, epoch=0, tokenizer=None):
    if args.train_data == "synthetic-raw":
        # decoded images of assorted sizes (uint8, HWC, in HBM) through the GPU grid-distill pipeline
        from ..hip import HipOps
        g = torch.Generator().manual_seed(77 + args.seed)
        sizes = [(427, 640), (640, 480), (500, 375), (333, 500), (480, 640), (612, 612)]
        n = max(args.batch_size, 8)
        images = [torch.randint(0, 256, (*sizes[i % len(sizes)], 3), generator=g, dtype=torch.uint8).to(args.device) for i in range(n)]
        size = args.synthetic_image_size or args.det_image_size
        seed = 1234 + args.seed + 7919 * args.rank
        if args.dataset_type == "proposals_distill":
            anns = []
            for im in images:                                           # random annotation boxes (x, y, w, h), a few of them tiny
                H, W = im.shape[:2]
                k = int(torch.randint(0, 12, (1,), generator=g))
                xy = torch.rand(k, 2, generator=g) * torch.tensor([W * 0.7, H * 0.7])
                wh = torch.rand(k, 2, generator=g) * torch.tensor([W * 0.3, H * 0.3]) + 2.0
                anns.append(torch.cat([xy, wh], 1).tolist())
            loader = GpuProposalDistillLoader(images, anns, HipOps(), args.batch_size, size, args.input_size, min_size=args.min_size,
                                              max_size=args.max_size, steps=args.synthetic_steps, seed=seed)
        else:
            loader = GpuGridDistillLoader(images, HipOps(), args.batch_size, args.max_boxes, size, args.input_size, max_split=args.max_split,
                                          crop_scale=args.crop_scale, steps=args.synthetic_steps, seed=seed)
        return _with_val({"train": DataInfo(loader)}, args)
    if args.train_data and args.train_data != "synthetic" and os.path.isfile(args.train_data):
        return _with_val({"train": DataInfo(coco_train_loader(args))}, args)
    if not args.train_data:                                             # evaluation-only runs (scripts/test_*.sh: --train-data "")
        data = _with_val({}, args)
        if data:
            return data
    if args.train_data != "synthetic":
        raise NotImplementedError(
            f"--train-data {args.train_data!r}: expected a COCO-style annotation file (decoded on the host, cropped / resized on the GPU: "
            "training/coco_source.py), 'synthetic' or 'synthetic-raw'")
    size = args.synthetic_image_size or args.det_image_size
    loader = _SyntheticLoader(args.synthetic_steps, args.batch_size, args.max_boxes, size, args.input_size,
                              args.device, args.rank, args.world_size, seed=1234 + args.seed,
                              valid_prob=0.7 if args.dataset_type == "proposals_distill" else 1.0, resident=False)
    loader.region_clip = args.dataset_type == "region_clip"
    return _with_val({"train": DataInfo(loader)}, args)
