"""TEST INFRASTRUCTURE -- numpy restatement of the integer arithmetic behind `cs_crop_resize_u8` (clipself_amd/csrc/preprocess.hip), i.e. of
what the reference's crop pipeline makes Pillow compute (call sites: src/training/data.py:226-245, src/open_clip/transform.py:26-49,169-191):

    Image.crop(box)                 box rounded per coordinate (Python round = half to even), pixels outside the image are 0
    Image.resize((nw, nh), BICUBIC) Pillow's ImagingResample (third-party, Pillow 9+; not vendored by the reference): per output position the
                                    taps [xmin, xmax) of a bicubic kernel (a = -0.5) whose support is scaled by max(scale, 1), weights
                                    normalised in double precision and rounded to 22-bit fixed point; horizontal pass over every crop row,
                                    then vertical pass, each pass accumulating in int32 from 1 << 21, shifting by 22 and clipping to uint8
    pad to S x S (centred or right/bottom), /255, (x - mean) / std in float32

Pinned on CPU against Pillow itself (tests/test_data_cpu.py) -- the GPU test pins the HIP kernel against Pillow as well, so the three agree
bit for bit.  Nothing under clipself_amd/ imports this file."""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    x = np.abs(x)
    a = -0.5
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0, np.where(x < 2.0, (((x - 5.0) * x + 8.0) * x - 4.0) * a, 0.0))


def coefficients(in_size: int, out_size: int):
    """Per output position: (first tap, int32 weights) as Pillow's precompute_coeffs + normalize_coeffs_8bpc produce them."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size)
        w = _bicubic((np.arange(xmin, xmax, dtype=np.float64) - center + 0.5) * (1.0 / filterscale))
        total = w.sum()
        if total != 0.0:
            w = w / total
        k = np.where(w < 0, -0.5 + w * (1 << PRECISION_BITS), 0.5 + w * (1 << PRECISION_BITS)).astype(np.int64)   # C cast: toward zero
        out.append((xmin, k))
    return out


def _resample_axis0(img: np.ndarray, out_size: int) -> np.ndarray:
    """uint8 [n, ...] -> uint8 [out_size, ...] along axis 0."""
    res = np.empty((out_size,) + img.shape[1:], np.uint8)
    wide = img.astype(np.int64)
    for i, (lo, k) in enumerate(coefficients(img.shape[0], out_size)):
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(k, wide[lo:lo + len(k)], axes=(0, 0))
        res[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return res


def crop_resize(image_hwc_u8: np.ndarray, boxes, size: int, pad_center: bool, mean, std) -> np.ndarray:
    H, W, _ = image_hwc_u8.shape
    mean = np.asarray(mean, np.float32)[:, None, None]
    std = np.asarray(std, np.float32)[:, None, None]
    out = np.empty((len(boxes), 3, size, size), np.float32)
    for b, box in enumerate(boxes):
        x0, y0, x1, y1 = (int(np.rint(np.float64(v))) for v in box)          # Python round() on a float = rint (half to even)
        cw, ch = max(x1 - x0, 1), max(y1 - y0, 1)
        crop = np.zeros((ch, cw, 3), np.uint8)
        sx0, sy0, sx1, sy1 = max(x0, 0), max(y0, 0), min(x0 + cw, W), min(y0 + ch, H)
        if sx1 > sx0 and sy1 > sy0:
            crop[sy0 - y0:sy1 - y0, sx0 - x0:sx1 - x0] = image_hwc_u8[sy0:sy1, sx0:sx1]
        scale = size / float(max(ch, cw))
        nh, nw = max(int(np.rint(ch * scale)), 1), max(int(np.rint(cw * scale)), 1)
        horiz = _resample_axis0(crop.transpose(1, 0, 2), nw).transpose(1, 0, 2) if nw != cw else crop      # Pillow skips a pass that keeps the size
        small = _resample_axis0(horiz, nh) if nh != ch else horiz
        canvas = np.zeros((size, size, 3), np.uint8)
        ox, oy = ((size - nw) // 2, (size - nh) // 2) if pad_center else (0, 0)
        canvas[oy:oy + nh, ox:ox + nw] = small
        t = canvas.transpose(2, 0, 1).astype(np.float32) / np.float32(255)
        out[b] = (t - mean) / std
    return out
