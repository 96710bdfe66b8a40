"""Host side of the GPU input pipeline: COCO-style annotation files and image files -> decoded RGB images in HBM.

The reference reads its training sets through pycocotools + PIL inside DataLoader workers (src/training/data.py:30-83 and :135-197:
`COCO(input_filename)`, `coco.imgs`, `coco.imgToAnns`, `read_image` = `Image.open(os.path.join(image_root, file_name))`, images under
10 px or unreadable replaced by a random other sample).  Here only the *decode* stays on the host -- a small thread pool reads ahead
along the epoch's order -- and everything after it (crop, resize, pad, normalise) runs in `cs_crop_resize_u8` through
`GpuGridDistillLoader` / `GpuProposalDistillLoader` (clipself_amd/training/data.py), which index these objects like lists.
pycocotools is not needed: the two dictionaries it would build are built directly from the json.
"""
from __future__ import annotations

import json
import logging
import os
import random
from collections import OrderedDict, defaultdict
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


class CocoIndex:
    """`coco.imgs` / `coco.imgToAnns` of a COCO-style file (instances / captions / proposals json)."""

    def __init__(self, annotation_file: str):
        with open(annotation_file) as f:
            blob = json.load(f)
        self.imgs = OrderedDict((im["id"], im) for im in blob.get("images", []))
        self.imgToAnns = defaultdict(list)
        for ann in blob.get("annotations", []):
            self.imgToAnns[ann["image_id"]].append(ann)
        self.image_ids = list(self.imgs.keys())
        self.cats = {c["id"]: c for c in blob.get("categories", [])}
        self.cat_id2label = {cid: i for i, cid in enumerate(sorted(self.cats))}        # data.py:312-314,414-416

    @staticmethod
    def file_name(info: dict) -> str:
        """data.py:93-98: `file_name`, or the last two components of `coco_url` (LVIS)."""
        if "file_name" in info:
            return info["file_name"]
        parts = info["coco_url"].split("/")
        return os.path.join(parts[-2], parts[-1])


def decode_rgb(path: str):
    """uint8 [H, W, 3] numpy array of an image file, or None when it cannot be read / is under 10 px (data.py:52-83)."""
    from PIL import Image
    try:
        with Image.open(path) as im:
            if im.width < 10 or im.height < 10:
                print(f"Invalid image, size {im.size}", flush=True)
                return None
            return np.array(im.convert("RGB"))                     # own, writable buffer
    except Exception:                                        # the reference catches everything here as well
        print(f"Cannot load {path}", flush=True)
        return None


class DecodedImages:
    """List-like view of a dataset's images as uint8 HWC tensors on `device`, decoded lazily.  `hint(order)` (called by the loaders with
    the epoch's sample order) starts reading ahead on `workers` threads, `depth` images in front of the consumer."""

    def __init__(self, index: CocoIndex, image_root: str, device="cpu", image_ids=None, workers: int = 8, depth: int = 256, seed: int = 0):
        self.index, self.root, self.device = index, image_root, torch.device(device)
        self.image_ids = list(index.image_ids if image_ids is None else image_ids)
        self.pool = ThreadPoolExecutor(max_workers=max(workers, 1))
        self.depth = depth
        self._pending = OrderedDict()          # position -> Future
        self._order, self._cursor = [], 0
        self._rng = random.Random(seed)
        self.resolved = {}                     # requested position -> position actually served (differs after a fallback)

    def __len__(self):
        return len(self.image_ids)

    def path_of(self, i: int) -> str:
        return os.path.join(self.root, CocoIndex.file_name(self.index.imgs[self.image_ids[i]]))

    def annotations_of(self, i: int):
        return self.index.imgToAnns[self.image_ids[i]]

    def annotation_view(self):
        """List-like view of the raw annotation dicts aligned with this image list (follows fallbacks like AnnotationBoxes)."""
        images = self

        class _View:
            def __len__(self):
                return len(images)

            def __getitem__(self, i):
                return images.annotations_of(images.resolved.get(i, i))
        return _View()

    def hint(self, order):
        self._order, self._cursor = list(order), 0
        for fut in self._pending.values():
            fut.cancel()
        self._pending.clear()
        self._fill()

    def _fill(self):
        while self._cursor < len(self._order) and len(self._pending) < self.depth:
            i = self._order[self._cursor]
            self._cursor += 1
            if i not in self._pending:
                self._pending[i] = self.pool.submit(decode_rgb, self.path_of(i))

    def _decoded(self, i: int):
        fut = self._pending.pop(i, None)
        arr = fut.result() if fut is not None else decode_rgb(self.path_of(i))
        self._fill()
        return arr

    def resolve(self, i: int):
        """(index actually used, uint8 HWC tensor): an unreadable sample is replaced by a random other one (data.py:101-103,260-262)."""
        asked = i
        for _ in range(100):
            arr = self._decoded(i)
            if arr is not None:
                t = torch.from_numpy(arr)
                if self.device.type == "cuda":
                    t = t.pin_memory().to(self.device, non_blocking=True)
                self.resolved[asked] = i
                return i, t
            i = self._rng.randrange(len(self))
        raise RuntimeError(f"no readable image under {self.root}")

    def __getitem__(self, i: int):
        return self.resolve(i)[1]


class AnnotationBoxes:
    """List-like view of the `bbox` lists (x, y, w, h in pixels) aligned with a DecodedImages."""

    def __init__(self, images: DecodedImages):
        self.images = images

    def __len__(self):
        return len(self.images)

    def __getitem__(self, i: int):
        i = self.images.resolved.get(i, i)                    # follow the image that was actually served for this position
        return [list(a["bbox"]) for a in self.images.annotations_of(i)]


def subset_ids(index: CocoIndex, train_ratio: float, rank: int = 0, world: int = 1, seed: int = 0, annotated_only: bool = False):
    """image ids of this rank: the `train_ratio` random subset of GridDistillDataset (data.py:150-155), then every world-th id starting at
    `rank` (what DistributedSampler hands each process, data.py:548).  The subset is drawn from a generator seeded identically on every
    rank (the reference shuffles with each process's own global `random` state, so its ranks disagree about the subset)."""
    ids = [i for i in index.image_ids if index.imgToAnns.get(i)] if annotated_only else list(index.image_ids)   # data.py:397: RegionCLIP
    if train_ratio < 1.0:
        random.Random(seed).shuffle(ids)
        ids = ids[:int(len(ids) * train_ratio)]
    if world > 1:
        usable = len(ids) - len(ids) % world
        ids = ids[:usable][rank::world] if usable else ids
    logging.info(f"coco source: {len(ids)} images on rank {rank} of {world}")
    return ids


# ------------------------------------------------------------------------------------------------------------------ panoptic validation set
def rgb2id(color: np.ndarray) -> np.ndarray:
    """Segment ids of a COCO-panoptic PNG (panopticapi convention: R + 256 G + 256^2 B)."""
    c = color.astype(np.int64)
    return c[..., 0] + 256 * c[..., 1] + 256 * 256 * c[..., 2]


class CocoPanopticVal:
    """The reference's validation set (COCOPanopticDataset, src/training/data.py:283-387; index as COCOPanoptic, src/training/coco_api.py:52-112)
    with the pixel work on the GPU: per image one batch (the reference evaluates with batch size 1, data.py:484-487) of
        images [1,3,S,S], bboxes [1,K,8] = (x0,y0,x1,y1 in [0,1] of the padded square, class, valid, area, is_thing),
        image_crops [1,K,3,Sc,Sc], gt_masks [1,K,S/f,S/f], masked_image_crops [1,K,3,Sc,Sc]
    K = min(max segments per image, 100).  Things use their box enlarged 1.5x (clipped) as the crop, stuff the bounding box of its segment
    (`mask2box`: inclusive max coordinates, training/utils.py:25-30); areas outside [8^2, 1024^2] px are skipped; masks are the segment
    resized like `ResizeLongest(S / f)` does for tensors (bicubic, then > 0) -- torchvision's tensor resize is not pinned by the reference,
    the anti-aliased form of current releases is used.  `masked_image_crops` (crops of the image with everything outside the segment set to
    114) are produced only with masked_crops=True: zero_shot.run never reads them (zero_shot.py:30-75)."""

    def __init__(self, annotation_file, image_root, segm_root, embed_path, ops, device, det_size, crop_size, downsample_factor=16, rank=0,
                 world=1, masked_crops=False):
        with open(annotation_file) as f:
            blob = json.load(f)
        self.imgs = OrderedDict()
        for info in blob.get("images", []):
            info = dict(info, segm_file=info["file_name"].replace("jpg", "png"))
            self.imgs[info["id"]] = info
        self.img_to_anns = defaultdict(list)
        for ann in blob.get("annotations", []):
            for seg in ann["segments_info"]:
                self.img_to_anns[ann["image_id"]].append(dict(seg, image_id=ann["image_id"]))
        self.cats = {c["id"]: c for c in blob.get("categories", [])}
        self.cat_id2label = {cid: i for i, cid in enumerate(sorted(self.cats))}
        self.embeddings = np.load(embed_path)
        self.image_ids = list(self.imgs.keys())[rank::world] if world > 1 else list(self.imgs.keys())
        self.max_anns = min(max((len(v) for v in self.img_to_anns.values()), default=1), 100)
        self.image_root, self.segm_root, self.ops, self.device = image_root, segm_root, ops, torch.device(device)
        self.det_size, self.crop_size, self.mask_size = det_size, crop_size, det_size // downsample_factor
        self.min_size, self.max_size, self.masked_crops = 8, 1024, masked_crops
        self.batches = self                      # `_ValLoader`-style consumers iterate `dataset.batches`

    def __len__(self):
        return len(self.image_ids)

    def _mask(self, segment: np.ndarray) -> torch.Tensor:
        H, W = segment.shape
        scale = self.mask_size / float(max(H, W))
        nh, nw = round(H * scale), round(W * scale)
        m = torch.from_numpy(segment).float()[None, None]
        m = torch.nn.functional.interpolate(m, size=(nh, nw), mode="bicubic", align_corners=False, antialias=True)[0, 0]
        out = torch.zeros(self.mask_size, self.mask_size)
        out[:nh, :nw] = (m > 0.0).float()
        return out

    def item(self, idx: int):
        from PIL import Image
        info = self.imgs[self.image_ids[idx]]
        image = decode_rgb(os.path.join(self.image_root, info["file_name"]))
        if image is None:
            raise RuntimeError(f"validation image {info['file_name']} cannot be read")
        with Image.open(os.path.join(self.segm_root, info["segm_file"])) as im:
            segm = rgb2id(np.array(im.convert("RGB"), dtype=np.uint8))
        H, W = image.shape[:2]
        dev, K = self.device, self.max_anns
        img = torch.from_numpy(image).to(dev)
        boxes, masks = torch.zeros(K, 8), torch.zeros(K, self.mask_size, self.mask_size)
        slots, crop_px, segments = [], [], []
        for i, ann in enumerate(self.img_to_anns[info["id"]][:K]):
            is_thing = self.cats[ann["category_id"]]["isthing"]
            segment = segm == ann["id"]
            if is_thing > 0:
                x, y, w, h = ann["bbox"]
                cx, cy = x + w * 0.5, y + h * 0.5
                crop = [max(cx - w * 0.75, 0), max(cy - h * 0.75, 0), min(cx + w * 0.75, W), min(cy + h * 0.75, H)]
            else:
                if not segment.any():
                    continue
                ys, xs = np.where(segment)
                crop = [float(xs.min()), float(ys.min()), float(xs.max()), float(ys.max())]
                x, y, w, h = crop[0], crop[1], crop[2] - crop[0], crop[3] - crop[1]
            if w * h < self.min_size ** 2 or w * h > self.max_size ** 2:
                continue
            boxes[i] = torch.tensor([x, y, x + w, y + h, self.cat_id2label[ann["category_id"]], 1.0, w * h, is_thing], dtype=torch.float32)
            masks[i] = self._mask(segment)
            slots.append(i)
            crop_px.append(crop)
            segments.append(segment)
        crops = torch.zeros(K, 3, self.crop_size, self.crop_size, device=dev)
        masked = torch.zeros(K, 3, self.crop_size, self.crop_size, device=dev)
        if slots:
            px = torch.tensor(crop_px, dtype=torch.float32, device=dev)
            crops[slots] = self.ops.crop_resize(img, px, self.crop_size, pad_center=True)
            if self.masked_crops:
                for s, segment, box in zip(slots, segments, crop_px):
                    grey = image.copy()
                    grey[~segment] = 114
                    masked[s] = self.ops.crop_resize(torch.from_numpy(grey).to(dev), px.new_tensor([box]), self.crop_size, pad_center=True)[0]
        det = self.ops.crop_resize(img, torch.tensor([[0.0, 0.0, float(W), float(H)]], device=dev), self.det_size, pad_center=False)[0]
        boxes[:, :4] *= min(self.det_size / H, self.det_size / W) / self.det_size
        return det[None], boxes[None].to(dev), crops[None], masks[None].to(dev), masked[None]

    def __iter__(self):
        for i in range(len(self)):
            yield self.item(i)
