"""TEST INFRASTRUCTURE -- CPU statement of the crop pipeline of the reference's distillation datasets, executed with the real
third-party library the reference calls (Pillow, through torchvision.transforms.functional in the original):

    GridDistillDataset._obtain_image_crops   src/training/data.py:226-245   image.crop(box) -> transforms[1]
    ResizeMaxSize (crop transform)           src/open_clip/transform.py:26-49  longest side -> S, bicubic, centred zero padding
    ResizeLongest (det transform)            src/open_clip/transform.py:169-191 longest side -> S, bicubic, right/bottom zero padding
    ToTensor + Normalize                     src/open_clip/transform.py:93-99,160-165

Used only by tests/ to pin clipself_amd/csrc/preprocess.hip (cs_crop_resize_u8), which restates Pillow's resampling bit-exactly.
"""
import numpy as np
from PIL import Image

OPENAI_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_STD = (0.26862954, 0.26130258, 0.27577711)


def pil_crops(image_hwc_u8: np.ndarray, boxes, size: int, pad_center: bool = True, mean=OPENAI_MEAN, std=OPENAI_STD) -> np.ndarray:
    img = Image.fromarray(image_hwc_u8, mode="RGB")
    mean = np.asarray(mean, np.float32)[:, None, None]
    std = np.asarray(std, np.float32)[:, None, None]
    out = np.empty((len(boxes), 3, size, size), np.float32)
    for i, box in enumerate(boxes):
        crop = img.crop(tuple(float(v) for v in box))
        w, h = crop.size
        scale = size / float(max(h, w))
        nh, nw = round(h * scale), round(w * scale)
        small = crop.resize((nw, nh), Image.BICUBIC)
        canvas = np.zeros((size, size, 3), np.uint8)
        ox, oy = ((size - nw) // 2, (size - nh) // 2) if pad_center else (0, 0)
        canvas[oy:oy + nh, ox:ox + nw] = np.asarray(small)
        t = canvas.transpose(2, 0, 1).astype(np.float32) / np.float32(255)
        out[i] = (t - mean) / std
    return out
