"""ORACLE (test infrastructure only -- never imported by the product path): the image input stage that precedes the
model in the reference's inference jobs and training datasets.

The reference builds it with timm (multimodal_infer.py:86-91, cv_classifier_train.py:39-40):
``create_transform(input_size=(3,S,S), interpolation='bicubic', mean, std, crop_pct)`` = torchvision
``Resize(int(S / crop_pct), bicubic)`` on a PIL image -> ``CenterCrop(S)`` -> ``ToTensor`` -> ``Normalize(mean, std)``.
timm and torchvision are not installed here; their published pipeline is restated below, and the one
arithmetic-heavy piece -- Pillow's antialiased two-pass 8-bit bicubic resampling (Pillow ``src/libImaging/Resample.c``:
``precompute_coeffs``, ``normalize_coeffs_8bpc``, ``ImagingResampleHorizontal_8bpc`` / ``Vertical_8bpc``) -- is PINNED
bit-exactly against Pillow itself (installed: 12.2.0), see tests/test_oracle.py and tests/golden/gen_golden.py.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2      # Resample.c: fixed-point position of the 8-bit kernels


def _bicubic(x, a=-0.5):
    """Resample.c bicubic_filter (Keys, a = -0.5), support 2."""
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size, out_size, support=2.0):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the whole axis (box = [0, in_size)).
    Returns (ksize, bounds int32 [out,2] = (first input index, tap count), coefficients int32 [out, ksize])."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    sup = support * filterscale
    ksize = int(math.ceil(sup)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - sup + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + sup + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)                                   # same left-to-right double accumulation as the C loop
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bicubic_u8(img, out_w, out_h):
    """PIL ``Image.resize((out_w, out_h), BICUBIC)`` of an [H, W, C] uint8 image: horizontal pass rounded to 8 bits,
    then the vertical pass (ImagingResample's two passes)."""
    H, W, C = img.shape
    src = img.astype(np.int64)
    if out_w != W:
        _, bx, kx = resample_coeffs(W, out_w)
        tmp = np.empty((H, out_w, C), np.uint8)
        for xx in range(out_w):
            x0, n = bx[xx]
            acc = (src[:, x0:x0 + n, :] * kx[xx, :n].astype(np.int64)[None, :, None]).sum(1) + (1 << (PRECISION_BITS - 1))
            tmp[:, xx, :] = _clip8(acc)
        src = tmp.astype(np.int64)
    else:
        tmp = img
    if out_h != H:
        _, by, ky = resample_coeffs(H, out_h)
        out = np.empty((out_h, src.shape[1], C), np.uint8)
        for yy in range(out_h):
            y0, n = by[yy]
            acc = (src[y0:y0 + n] * ky[yy, :n].astype(np.int64)[:, None, None]).sum(0) + (1 << (PRECISION_BITS - 1))
            out[yy] = _clip8(acc)
        return out
    return np.ascontiguousarray(tmp)


def resize_target(H, W, size):
    """torchvision ``Resize(int)``: the shorter side becomes ``size``, the longer one int(size * long / short)."""
    if W <= H:
        return size, int(size * H / W)          # (out_w, out_h)
    return int(size * W / H), size


def crop_origin(H, W, S):
    """torchvision ``CenterCrop``: int(round((H - S) / 2.0)) -- Python's round (half to even)."""
    return int(round((H - S) / 2.0)), int(round((W - S) / 2.0))


IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def eval_transform(img, img_size=320, crop_pct=1.0, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """[H, W, 3] uint8 -> [3, S, S] float32, the reference's inference transform (multimodal_infer.py:86-91).
    ToTensor = uint8 -> float32 / 255; Normalize = (t - mean) / std, all in float32 as torch does."""
    H, W, _ = img.shape
    scale_size = int(math.floor(img_size / crop_pct))
    ow, oh = resize_target(H, W, scale_size)
    r = resize_bicubic_u8(img, ow, oh) if (ow, oh) != (W, H) else img
    if oh < img_size or ow < img_size:
        raise ValueError("image smaller than the crop after Resize (torchvision would pad)")
    top, left = crop_origin(oh, ow, img_size)
    c = r[top:top + img_size, left:left + img_size]
    t = c.astype(np.float32) / np.float32(255.0)
    m = np.asarray(mean, np.float32)
    s = np.asarray(std, np.float32)
    return np.ascontiguousarray(((t - m) / s).transpose(2, 0, 1))
