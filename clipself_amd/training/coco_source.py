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


def subset_ids(index: CocoIndex, train_ratio: float, rank: int = 0, world: int = 1, seed: int = 0):
    """image ids of this rank: the `train_ratio` random subset of GridDistillDataset (data.py:150-155), then every world-th id starting at
    `rank` (what DistributedSampler hands each process, data.py:548).  The subset is drawn from a generator seeded identically on every
    rank (the reference shuffles with each process's own global `random` state, so its ranks disagree about the subset)."""
    ids = list(index.image_ids)
    if train_ratio < 1.0:
        random.Random(seed).shuffle(ids)
        ids = ids[:int(len(ids) * train_ratio)]
    if world > 1:
        usable = len(ids) - len(ids) % world
        ids = ids[:usable][rank::world] if usable else ids
    logging.info(f"coco source: {len(ids)} images on rank {rank} of {world}")
    return ids
