"""CPU: the dataset-side logic of the GPU loaders (clipself_amd/training/data.py) with the pixel work done by Pillow through the
reference ops -- grid choices, shuffling and truncation, crop_scale enlargement, box rescaling, proposal filtering and fallback."""
import numpy as np
import pytest
import torch

from clipself_amd.training.data import GpuGridDistillLoader, GpuProposalDistillLoader, SyntheticPanopticVal, grid_boxes, grid_choices
from oracle.ops_ref import RefOps


def test_grid_templates_follow_the_reference_recipe():
    ch = grid_choices(6)
    assert len(ch) == 24 and ch[0] == (1, 1) and (6, 6) in ch and (1, 3) not in ch and (2, 5) not in ch      # n in [ceil(m/2), min(2m, 6)]
    b = grid_boxes(2, 3)
    assert b.shape == (6, 4)
    assert torch.allclose(b[0], torch.tensor([0, 0, 1 / 3, 0.5])) and torch.allclose(b[4], torch.tensor([1 / 3, 0.5, 2 / 3, 1.0]))
    assert torch.allclose((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]), torch.full((6,), 1 / 6))               # cells tile the image


def test_grid_loader_contract_on_cpu():
    rng = np.random.default_rng(0)
    imgs = [torch.from_numpy(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)) for h, w in ((120, 200), (150, 90))]
    loader = GpuGridDistillLoader(imgs, RefOps(), batch_size=2, max_boxes=5, det_size=64, crop_size=32, max_split=6, crop_scale=1.5, steps=2, seed=1)
    seen = 0
    for images, boxes, crops in loader:
        assert images.shape == (2, 3, 64, 64) and boxes.shape == (2, 5, 5) and crops.shape == (2, 5, 3, 32, 32)
        valid = boxes[..., 4] > 0.5
        assert valid.any(dim=1).all() and torch.all(crops[~valid] == 0) and torch.all(boxes[~valid] == 0)
        assert float(boxes[..., :4].min()) >= 0 and float(boxes[..., :4].max()) <= 1.0 + 1e-6
        seen += 1
    assert seen == 2 and len(loader) == 2
    # the box handed to the student is the grid cell; the crop handed to the teacher is that cell enlarged 1.5x and clipped
    det, bx, cr, crop_px = loader.sample(imgs[0])
    k = int(bx[:, 4].sum())
    cell = bx[:k, :4] * 64 / min(64 / 120, 64 / 200)
    assert torch.all(crop_px[:, :2] <= cell[:, :2] + 1e-3) and torch.all(crop_px[:, 2:] >= cell[:, 2:] - 1e-3)
    assert float(crop_px[:, 2].max()) <= 200 and float(crop_px[:, 3].max()) <= 120


def test_proposal_loader_filters_and_falls_back_on_cpu():
    rng = np.random.default_rng(1)
    img = torch.from_numpy(rng.integers(0, 256, (100, 160, 3), dtype=np.uint8))
    anns = [[10.0, 10.0, 50.0, 40.0], [0.0, 0.0, 2.0, 3.0], [60.0, 30.0, 90.0, 60.0]]                      # the second is below min_size^2
    loader = GpuProposalDistillLoader([img, img], [anns, [[1.0, 1.0, 1.0, 1.0]]], RefOps(), batch_size=2, det_size=64, crop_size=32, steps=1, seed=3)
    det, boxes, crops, crop_px, slots = loader.sample(img, anns)
    assert len(slots) == 2 and int(boxes[:, 4].sum()) == 2 and boxes.shape == (20, 5)
    assert torch.all(crops[[i for i in range(20) if i not in slots]] == 0)
    det, boxes, crops, crop_px, slots = loader.sample(img, [[1.0, 1.0, 1.0, 1.0]])                         # nothing valid -> quarter image
    assert slots == [0] and crop_px.tolist() == [[0, 0, 40, 25]]
    assert torch.allclose(boxes[0], torch.tensor([0, 0, 40 * 0.4 / 64, 25 * 0.4 / 64, 1.0]))
    b = next(iter(loader))
    assert b[0].shape == (2, 3, 64, 64) and b[2].shape == (2, 20, 3, 32, 32)


def test_synthetic_panoptic_val_contract():
    val = SyntheticPanopticVal(2, 3, 4, 32, 32, 4, 16, num_classes=5, seed=2)
    images, bboxes, crops, masks, masked = val.batches[0]
    assert images.shape == (3, 3, 32, 32) and bboxes.shape == (3, 4, 8) and crops.shape == (3, 4, 3, 32, 32) and masks.shape == (3, 4, 4, 4)
    assert val.embeddings.shape == (5, 16) and masks.sum(dim=(-1, -2)).min() >= 1 and set(bboxes[..., 7].unique().tolist()) <= {0.0, 1.0}


def test_numpy_restatement_of_the_resampling_is_pillow_exact():
    """oracle/resample_ref.py (the arithmetic cs_crop_resize_u8 implements: rounded crop box, 22-bit fixed-point bicubic taps, horizontal then
    vertical pass with uint8 rounding, pad, normalise) against Pillow on grid cells, free boxes (down- and up-sampling), boxes sticking out of
    the image, .5 edges and the whole-image (det transform) case."""
    from oracle.pil_crops_ref import OPENAI_MEAN, OPENAI_STD, pil_crops
    from oracle.resample_ref import crop_resize
    for (H, W, size, center) in ((213, 320, 96, True), (250, 167, 64, True), (96, 130, 224, True), (240, 320, 112, False)):
        rng = np.random.default_rng(H * 7 + W)
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        img[: H // 2, : W // 3] = (rng.integers(0, 256, (H // 2, W // 3, 1)) // 3 + 90).astype(np.uint8)
        boxes = []
        for M, N in ((1, 1), (2, 3), (5, 3)):
            xs, ys = np.linspace(0, 1, N + 1) * W, np.linspace(0, 1, M + 1) * H
            boxes += [(xs[j], ys[i], xs[j + 1], ys[i + 1]) for i in range(M) for j in range(N)]
        for _ in range(8):
            x0, y0 = rng.uniform(0, W * 0.6), rng.uniform(0, H * 0.6)
            boxes.append((x0, y0, min(x0 + rng.uniform(8, W * 0.4), W), min(y0 + rng.uniform(8, H * 0.4), H)))
        boxes += [(10.5, 20.5, 41.5, 37.5), (-7.0, -3.0, 30.0, 25.0), (W - 20.0, H - 12.0, W + 9.0, H + 5.0)]
        if not center:
            boxes = [(0.0, 0.0, float(W), float(H))]
        boxes = np.asarray(boxes, np.float32)
        want = pil_crops(img, boxes, size, center)
        got = crop_resize(img, boxes, size, center, OPENAI_MEAN, OPENAI_STD)
        assert got.shape == want.shape and int((got != want).sum()) == 0, (H, W, size, int((got != want).sum()))


def _write_coco(tmp_path, with_anns=True):
    """Six image files of assorted sizes and formats (one corrupt, one under 10 px) + a COCO-style json (one entry named by coco_url)."""
    import json
    from PIL import Image
    rng = np.random.default_rng(5)
    root = tmp_path / "images"
    (root / "val2017").mkdir(parents=True)
    specs = [("a.png", 120, 200), ("b.jpg", 150, 90), ("val2017/c.png", 64, 64), ("d.png", 97, 131), ("tiny.png", 5, 5), ("broken.jpg", 0, 0)]
    images, anns = [], []
    for k, (name, h, w) in enumerate(specs):
        if name == "broken.jpg":
            (root / name).write_bytes(b"not an image")
        else:
            mode = "L" if name == "d.png" else "RGB"                       # a grey-scale file: the reference converts to RGB too
            arr = rng.integers(0, 256, (h, w) if mode == "L" else (h, w, 3), dtype=np.uint8)
            Image.fromarray(arr, mode=mode).save(root / name, quality=92)
        info = {"id": 100 + k, "height": h, "width": w}
        if name.startswith("val2017/"):
            info["coco_url"] = "http://images.cocodataset.org/val2017/c.png"
        else:
            info["file_name"] = name
        images.append(info)
        if with_anns:
            for j in range(k + 1):
                anns.append({"id": 1000 + 10 * k + j, "image_id": 100 + k, "bbox": [3.0 * j, 2.0 * j, 20.0 + 5 * j, 12.0 + 7 * j], "category_id": 1})
    path = tmp_path / "ann.json"
    path.write_text(json.dumps({"images": images, "annotations": anns}))
    return path, root


def test_coco_files_feed_the_gpu_loaders_like_preloaded_images(tmp_path):
    """training/coco_source.py: annotation json + image files -> lazily decoded images (read-ahead threads) -> the same batches as the
    loaders produce from a preloaded list; unreadable and under-10-px files fall back to another sample, annotations follow the image."""
    from types import SimpleNamespace
    from PIL import Image
    from clipself_amd.training.coco_source import AnnotationBoxes, CocoIndex, DecodedImages, subset_ids
    from clipself_amd.training.data import coco_train_loader
    path, root = _write_coco(tmp_path)
    index = CocoIndex(str(path))
    assert index.image_ids == [100, 101, 102, 103, 104, 105] and len(index.imgToAnns[103]) == 4
    assert CocoIndex.file_name(index.imgs[102]) == "val2017/c.png"
    good = [0, 1, 2, 3]
    lazy = DecodedImages(index, str(root), "cpu", image_ids=[100, 101, 102, 103], workers=3, depth=2)
    pre = [torch.from_numpy(np.asarray(Image.open(lazy.path_of(i)).convert("RGB")).copy()) for i in good]
    assert all(torch.equal(lazy[i], pre[i]) for i in good) and lazy[3].shape == (97, 131, 3)
    anns = AnnotationBoxes(lazy)
    assert anns[2] == [[0.0, 0.0, 20.0, 12.0], [3.0, 2.0, 25.0, 19.0], [6.0, 4.0, 30.0, 26.0]]
    kw = dict(batch_size=2, det_size=64, crop_size=32, steps=3, seed=3)
    a = list(GpuGridDistillLoader(lazy, RefOps(), max_boxes=5, max_split=4, crop_scale=1.5, **kw))
    b = list(GpuGridDistillLoader(pre, RefOps(), max_boxes=5, max_split=4, crop_scale=1.5, **kw))
    assert len(a) == 3 and all(torch.equal(x, y) for ba, bb in zip(a, b) for x, y in zip(ba, bb))
    a = list(GpuProposalDistillLoader(lazy, anns, RefOps(), min_size=8, max_size=1024, **kw))
    b = list(GpuProposalDistillLoader(pre, [anns[i] for i in good], RefOps(), min_size=8, max_size=1024, **kw))
    assert all(torch.equal(x, y) for ba, bb in zip(a, b) for x, y in zip(ba, bb))

    # fallback: positions 4 (5x5 px) and 5 (corrupt) are served by another sample, and the annotation view follows
    every = DecodedImages(index, str(root), "cpu", workers=2, depth=4, seed=1)
    every.hint([5, 4, 0])
    j, img = every.resolve(5)
    assert j in good and torch.equal(img, pre[j]) and AnnotationBoxes(every)[5] == anns[j]
    j4, _ = every.resolve(4)
    assert j4 in good

    # sharding and the train_ratio subset are the same on every rank
    assert subset_ids(index, 1.0, 0, 2) == [100, 102, 104] and subset_ids(index, 1.0, 1, 2) == [101, 103, 105]
    half = [subset_ids(index, 0.5, r, 1, seed=9) for r in range(2)]
    assert half[0] == half[1] and len(half[0]) == 3

    # the entrypoint's data hook: --train-data <json> --train-image-root <dir>
    args = SimpleNamespace(train_data=str(path), train_image_root=str(root), dataset_type="grid_distill", device="cpu", rank=0, world_size=1, seed=0,
                           train_ratio=1.0, workers=1, det_image_size=64, input_size=32, batch_size=2, max_boxes=4, max_split=3, crop_scale=1.0,
                           min_size=8, max_size=1024)
    loader = coco_train_loader(args, ops=RefOps())
    assert loader.num_batches == 3
    images, boxes, crops = next(iter(loader))
    assert images.shape == (2, 3, 64, 64) and boxes.shape == (2, 4, 5) and crops.shape == (2, 4, 3, 32, 32) and float(boxes[..., 4].sum()) >= 2
    args.dataset_type = "proposals_distill"
    images, boxes, crops = next(iter(coco_train_loader(args, ops=RefOps())))
    assert boxes.shape == (2, 20, 5) and crops.shape == (2, 20, 3, 32, 32)


def _write_panoptic(tmp_path):
    """Two images + COCO-panoptic style files: <segm_root>/<name>.png with segment ids encoded as R + 256 G + 256^2 B, a json with
    segments_info (things carry a bbox, stuff is located through its mask) and categories with `isthing`, class embeddings as .npy."""
    import json
    from PIL import Image
    rng = np.random.default_rng(11)
    img_root, seg_root = tmp_path / "val", tmp_path / "panoptic"
    img_root.mkdir(), seg_root.mkdir()
    cats = [{"id": 7, "name": "thing-a", "isthing": 1}, {"id": 3, "name": "stuff-b", "isthing": 0}, {"id": 21, "name": "thing-c", "isthing": 1}] + \
           [{"id": 40 + i, "name": f"unused-{i}", "isthing": i % 2} for i in range(3)]          # top-5 needs more than five classes
    images, annotations = [], []
    layouts = {"p.jpg": (96, 128, [(300, 7, (10, 20, 50, 40)), (70000, 3, (64, 8, 60, 80)), (5, 21, (0, 70, 6, 5))]),      # last: 30 px^2 -> skipped
               "q.jpg": (80, 80, [(9, 3, (0, 0, 80, 30)), (1234567, 21, (20, 40, 40, 30))])}
    for k, (name, (H, W, segs)) in enumerate(layouts.items()):
        Image.fromarray(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).save(img_root / name, quality=95)
        ids = np.zeros((H, W), np.int64)
        info = []
        for sid, cat, (x, y, w, h) in segs:
            ids[y:y + h, x:x + w] = sid
            info.append({"id": sid, "category_id": cat, "bbox": [x, y, w, h], "area": w * h, "iscrowd": 0})
        png = np.stack([ids % 256, (ids // 256) % 256, ids // 65536], -1).astype(np.uint8)
        Image.fromarray(png).save(seg_root / name.replace("jpg", "png"))
        images.append({"id": 50 + k, "file_name": name, "height": H, "width": W})
        annotations.append({"image_id": 50 + k, "file_name": name.replace("jpg", "png"), "segments_info": info})
    ann = tmp_path / "panoptic.json"
    ann.write_text(json.dumps({"images": images, "annotations": annotations, "categories": cats}))
    emb = tmp_path / "emb.npy"
    np.save(emb, rng.standard_normal((6, 64)).astype(np.float32))
    return ann, img_root, seg_root, emb


def test_panoptic_validation_files_feed_the_zero_shot_evaluation(tmp_path):
    """training/coco_source.py:CocoPanopticVal -- the reference's COCOPanopticDataset contract (data.py:283-387) from real files -- and
    zero_shot.run / macc_with_is_thing on top of it with the tiny tower on the reference ops."""
    from types import SimpleNamespace
    from PIL import Image
    from clipself_amd.config import tiny_cfg
    from clipself_amd.init import seeded_visual_state
    from clipself_amd.open_clip.model import CustomCLIP
    from clipself_amd.training.coco_source import CocoPanopticVal, rgb2id
    from clipself_amd.training.zero_shot import zero_shot_eval
    from oracle.pil_crops_ref import pil_crops
    ann, img_root, seg_root, emb = _write_panoptic(tmp_path)
    ds = CocoPanopticVal(str(ann), str(img_root), str(seg_root), str(emb), RefOps(), "cpu", det_size=32, crop_size=32, downsample_factor=8)
    assert len(ds) == 2 and ds.max_anns == 3 and ds.mask_size == 4 and ds.cat_id2label == {3: 0, 7: 1, 21: 2, 40: 3, 41: 4, 42: 5} and ds.embeddings.shape == (6, 64)
    assert int(rgb2id(np.array([[[7, 1, 2]]], np.uint8))[0, 0]) == 7 + 256 + 2 * 65536
    images, boxes, crops, masks, masked = ds.item(0)
    assert images.shape == (1, 3, 32, 32) and boxes.shape == (1, 3, 8) and crops.shape == (1, 3, 3, 32, 32) and masks.shape == (1, 3, 4, 4)
    b = boxes[0]
    s = min(32 / 96, 32 / 128) / 32                                   # pixels -> fraction of the padded square
    assert torch.allclose(b[0], torch.tensor([10 * s, 20 * s, 60 * s, 60 * s, 1.0, 1.0, 2000.0, 1.0]))            # thing: its bbox
    assert torch.allclose(b[1], torch.tensor([64 * s, 8 * s, 123 * s, 87 * s, 0.0, 1.0, 59.0 * 79.0, 0.0]))       # stuff: mask2box (inclusive max)
    assert float(b[2].abs().sum()) == 0.0 and float(crops[0, 2].abs().sum()) == 0.0                                # 30 px^2 < 8^2: empty slot
    img = np.asarray(Image.open(img_root / "p.jpg").convert("RGB"))
    want = pil_crops(img, np.array([[0.0, 10.0, 72.5, 70.0], [64.0, 8.0, 123.0, 87.0]], np.float32), 32, True)      # thing: 1.5x box, clipped
    assert torch.equal(crops[0, :2], torch.from_numpy(want))
    assert torch.equal(images[0], torch.from_numpy(pil_crops(img, np.array([[0, 0, 128, 96]], np.float32), 32, False))[0])
    m = masks[0]
    # a 4x4 map of the 96x128 image (last row = padding): the bicubic resize + '> 0' dilates by the kernel's positive lobe, not further
    assert m[0].sum() > 0 and m[1].sum() > 0 and m[2].sum() == 0 and m[1][:, 0].sum() == 0 and m[0][:, 3].sum() == 0 and m[:, 3].sum() == 0
    assert float(masked.abs().sum()) == 0.0
    _, _, _, _, masked = CocoPanopticVal(str(ann), str(img_root), str(seg_root), str(emb), RefOps(), "cpu", 32, 32, 8, masked_crops=True).item(0)
    assert float(masked[0, 0].abs().sum()) > 0 and not torch.equal(masked[0, 0], crops[0, 0])

    cfg = tiny_cfg()
    model = CustomCLIP(cfg, ops=RefOps(), trainable=False)
    model.visual.engine.load_state(seeded_visual_state(cfg, 3))
    args = SimpleNamespace(device="cpu", precision="fp32", distributed=False, horovod=False, extract_type="v2", image_ave_pool=False, rank=0,
                           local_rank=0, world_size=1, zeroshot_frequency=1, epochs=1, val_data=str(ann), val_image_root=str(img_root),
                           val_segm_root=str(seg_root), embed_path=str(emb), det_image_size=32, input_size=32, downsample_factor=8,
                           train_data="", batch_size=1, max_boxes=4, seed=0)
    from clipself_amd.training.data import DataInfo, _ValLoader, coco_panoptic_val
    data = {"val": DataInfo(_ValLoader(coco_panoptic_val(args, ops=RefOps())))}       # what get_data() builds with the HIP ops
    assert data["val"].dataloader.num_batches == 2 and data["val"].dataloader.num_samples == 2
    metrics = zero_shot_eval(model, data, 1, args)
    assert {"rois.thing.macc1", "crops.thing.macc5", "maskpool.stuff.macc1"} <= set(metrics), sorted(metrics)
    assert all(0.0 <= v <= 1.0 for v in metrics.values())


def test_region_clip_batches_from_coco_files(tmp_path):
    """COCORegionCLIPDataset's contract (data.py:390-459) from files: annotated images only, boxes (xyxy in the padded square, label, valid)."""
    import json
    from types import SimpleNamespace
    from clipself_amd.training.data import coco_train_loader
    path, root = _write_coco(tmp_path)
    blob = json.loads(path.read_text())
    blob["categories"] = [{"id": 5, "name": "b"}, {"id": 1, "name": "a"}]
    blob["annotations"] = [a for a in blob["annotations"] if a["image_id"] != 101]             # image 101 loses its annotations -> dropped
    for k, a in enumerate(blob["annotations"]):
        a["category_id"] = 5 if k % 2 else 1
    path.write_text(json.dumps(blob))
    args = SimpleNamespace(train_data=str(path), train_image_root=str(root), dataset_type="region_clip", device="cpu", rank=0, world_size=1, seed=0,
                           train_ratio=1.0, workers=1, det_image_size=64, input_size=32, batch_size=2, max_boxes=4, max_split=3, crop_scale=1.0,
                           min_size=8, max_size=1024)
    loader = coco_train_loader(args, ops=RefOps())
    assert len(loader.images) == 5 and loader.max_anns == 6 and loader.num_batches == 2 and loader.cat_id2label == {1: 0, 5: 1}
    images, boxes = next(iter(loader))
    assert images.shape == (2, 3, 64, 64) and boxes.shape == (2, 6, 6)
    det, b = loader.sample(loader.images[3], loader.anns[3])                                    # d.png: 97 x 131, four annotations
    s = min(64 / 97, 64 / 131) / 64
    assert torch.allclose(b[1, :4], torch.tensor([3.0, 2.0, 28.0, 21.0]) * s) and b[:, 5].tolist() == [1, 1, 1, 1, 0, 0]
    assert set(b[:4, 4].tolist()) <= {0.0, 1.0} and det.shape == (3, 64, 64)


def test_create_model_and_transforms_returns_working_transforms():
    """open_clip.create_model_and_transforms hands the distillation datasets `[det transform, crop transform]` like the reference
    (src/open_clip/factory.py:303-350): callables PIL image -> normalised tensor.  det = ResizeLongest (right/bottom padding),
    crop = ResizeMaxSize (centred padding) -- checked against the Pillow statement of both (oracle/pil_crops_ref.py)."""
    from PIL import Image
    from clipself_amd.open_clip import create_model_and_transforms
    from oracle.ops_ref import RefOps
    from oracle.pil_crops_ref import pil_crops
    model, train_tf, val_tf = create_model_and_transforms("EVA02-CLIP-B-16", "eva", cache_dir=None, det_image_size=96,
                                                          dataset_type="grid_distill", ops=RefOps())
    assert isinstance(train_tf, list) and len(train_tf) == 2 and len(val_tf) == 2
    rng = np.random.default_rng(5)
    arr = rng.integers(0, 256, size=(75, 120, 3), dtype=np.uint8)
    img = Image.fromarray(arr, mode="RGB")
    whole = np.array([[0.0, 0.0, 120.0, 75.0]], np.float32)
    det = train_tf[0](img)
    assert tuple(det.shape) == (3, 96, 96) and det.dtype == torch.float32
    assert np.array_equal(det.numpy(), pil_crops(arr, whole, 96, pad_center=False)[0])
    crop = train_tf[1](img.crop((10, 5, 90, 70)))
    S = model.visual.image_size
    assert tuple(crop.shape) == (3, S, S)
    assert np.array_equal(crop.numpy(), pil_crops(arr, np.array([[10.0, 5.0, 90.0, 70.0]], np.float32), S, pad_center=True)[0])
    # zero padding sits right/bottom for the det transform (normalised zero = -mean/std), centred for the crop transform
    pad_val = (0.0 - 0.48145466) / 0.26862954
    assert abs(float(det[0, -1, -1]) - pad_val) < 1e-6 and abs(float(det[0, 95, 0]) - pad_val) < 1e-6        # 120x75 -> 96x60: rows 60.. are padding
    assert abs(float(crop[0, 0, 0]) - pad_val) < 1e-6 and abs(float(crop[0, -1, 0]) - pad_val) < 1e-6
    # arrays and grayscale PIL images are accepted like PIL RGB images
    assert torch.equal(train_tf[0](arr), det)
    assert tuple(train_tf[1](img.convert("L")).shape) == (3, S, S)
    # outside the distillation datasets the reference returns its train-time augmentation: not built, raises when called
    _, aug, _ = create_model_and_transforms("EVA02-CLIP-B-16", "eva", cache_dir=None, dataset_type=None, ops=RefOps())
    with pytest.raises(NotImplementedError):
        aug(img)
