"""GPU image input stage: the transform the reference builds with timm ahead of the image tower,

    config = {'input_size': (3, 320, 320), 'interpolation': 'bicubic', 'mean': ..., 'std': ..., 'crop_pct': 1.0}
    transform_eff = create_transform(**config)          # multimodal_infer.py:86-91, cv_classifier_train.py:39-40
    img_tensor = transform_eff(Image.open(path).convert('RGB'))

i.e. torchvision ``Resize(int(S / crop_pct), bicubic)`` -> ``CenterCrop(S)`` -> ``ToTensor`` -> ``Normalize``, which the
reference runs on 16 CPU dataloader workers / one image at a time.  Here the decoded uint8 image is copied to the GPU and
resized, cropped and normalised there (``mmsim_preprocess_image``): bit-exact with Pillow's 8-bit two-pass bicubic
resampling, fp32 ToTensor / Normalize.  ``create_transform`` keeps timm's keyword names for the eval pipeline.

Host side = the fixed-point resampling kernels (Pillow Resample.c ``precompute_coeffs`` + ``normalize_coeffs_8bpc``),
built once per (input size, output size) in float64 exactly as Pillow builds them and cached on the device.
There is no CPU path: the result is a CUDA tensor and the HIP library is required.
"""
import math

import numpy as np
import torch

from . import ops
from ._lib import lib, MmsimError

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
_PRECISION_BITS = 22


def _bicubic(x):
    a = -0.5
    x = np.abs(x)
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


def bicubic_kernels(in_size, out_size):
    """Pillow's 8-bit bicubic kernels for resizing an axis of in_size samples to out_size (antialiased when shrinking).
    -> (ksize, bounds int32 [out, 2] = (first input index, taps), coefficients int32 [out, ksize], 22 fractional bits)."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 2.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # C (int) casts truncate
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size)
    n = xmax - xmin
    t = np.arange(ksize, dtype=np.int64)[None, :]
    w = _bicubic((t + xmin[:, None] - center[:, None] + 0.5) * (1.0 / fscale))
    w = np.where(t < n[:, None], w, 0.0)
    ww = np.zeros(out_size, np.float64)
    for j in range(ksize):                    # Pillow sums the taps left to right in double; keep that order
        ww = ww + w[:, j]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    k = np.where(w < 0, -0.5 + w * (1 << _PRECISION_BITS), 0.5 + w * (1 << _PRECISION_BITS)).astype(np.int64)   # truncation
    k = np.where(t < n[:, None], k, 0)
    return ksize, np.stack([xmin, n], 1).astype(np.int32), k.astype(np.int32)


def _resize_target(H, W, size):
    if W <= H:
        return size, int(size * H / W)              # (out_w, out_h), torchvision Resize(int)
    return int(size * W / H), size


class ImageTransform:
    """Callable mirroring the eval transform of timm's ``create_transform``; see ``create_transform`` below."""

    def __init__(self, input_size=224, interpolation="bicubic", mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD,
                 crop_pct=0.875, device="cuda"):
        if isinstance(input_size, (tuple, list)):
            if len(input_size) != 3 or input_size[0] != 3 or input_size[1] != input_size[2]:
                raise ValueError("input_size must be (3, S, S)")
            input_size = input_size[-1]
        if interpolation != "bicubic":
            raise ValueError("only interpolation='bicubic' (the reference's setting) is implemented")
        if len(mean) != 3 or len(std) != 3 or any(float(s) == 0.0 for s in std):
            raise ValueError("mean / std must have 3 entries, std non-zero")
        self.size = int(input_size)
        self.scale_size = int(math.floor(self.size / crop_pct))
        self.mean = tuple(float(np.float32(m)) for m in mean)
        self.std = tuple(float(np.float32(s)) for s in std)
        self.device = torch.device(device)
        self._tables = {}
        self._tmp = None

    def _axis(self, n_in, n_out):
        key = (n_in, n_out)
        t = self._tables.get(key)
        if t is None:
            ks, b, k = bicubic_kernels(n_in, n_out)
            t = (ks, b, torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device))
            self._tables[key] = t
        return t

    def _to_device_u8(self, img):
        if isinstance(img, torch.Tensor):
            t = img
        else:
            if hasattr(img, "convert"):                     # PIL image, as the reference passes
                img = np.asarray(img.convert("RGB"))
            a = np.ascontiguousarray(img)
            t = torch.from_numpy(a if a.flags.writeable else a.copy())     # np.asarray(PIL image) is read-only
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
            raise TypeError("expected an RGB image: PIL.Image, or uint8 [H, W, 3] array / tensor")
        return t.to(self.device, non_blocking=True).contiguous()

    def into(self, img, out):
        """Transform one image into ``out`` ([3, S, S] fp32 CUDA, e.g. a row of a batch tensor)."""
        if self.device.type != "cuda":
            raise MmsimError("ImageTransform: the input stage runs on the GPU only; there is no CPU path")
        t = self._to_device_u8(img)
        H, W, S = t.shape[0], t.shape[1], self.size
        ow, oh = _resize_target(H, W, self.scale_size)
        if ow < S or oh < S:
            raise ValueError(f"image {W}x{H} is smaller than the {S}x{S} crop after Resize({self.scale_size})")
        ksx, _, bx, kx = self._axis(W, ow)
        ksy, by_h, by, ky = self._axis(H, oh)
        top, left = int(round((oh - S) / 2.0)), int(round((ow - S) / 2.0))      # torchvision CenterCrop
        row0 = int(by_h[top, 0])
        nrows = int(by_h[top + S - 1, 0] + by_h[top + S - 1, 1]) - row0
        need = nrows * S * 4
        if self._tmp is None or self._tmp.numel() < need:
            self._tmp = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=self.device)
        if out.dtype != torch.float32 or not out.is_cuda or tuple(out.shape) != (3, S, S) or not out.is_contiguous():
            raise TypeError(f"out must be a contiguous fp32 CUDA tensor [3, {S}, {S}]")
        with torch.cuda.device(self.device):
            lib.preprocess_image(t.data_ptr(), H, W, 3 * W, bx.data_ptr(), kx.data_ptr(), ksx, ow, by.data_ptr(), ky.data_ptr(),
                                 ksy, oh, left, top, S, row0, nrows, self._tmp.data_ptr(), self._tmp.numel(), out.data_ptr(),
                                 *self.mean, *self.std, ops._stream())
        t.record_stream(torch.cuda.current_stream(self.device))
        return out

    def __call__(self, img):
        """PIL image / uint8 [H, W, 3] -> fp32 [3, S, S] on the GPU (what ``transform_eff(img)`` returns in the reference)."""
        out = torch.empty(3, self.size, self.size, dtype=torch.float32, device=self.device)
        return self.into(img, out)

    def batch(self, images):
        """List of images (any sizes) -> fp32 [B, 3, S, S]: the collate the reference does with torch.stack."""
        out = torch.empty(len(images), 3, self.size, self.size, dtype=torch.float32, device=self.device)
        for i, im in enumerate(images):
            self.into(im, out[i])
        return out


def create_transform(input_size=224, is_training=False, interpolation="bicubic", mean=IMAGENET_DEFAULT_MEAN,
                     std=IMAGENET_DEFAULT_STD, crop_pct=0.875, device="cuda", **unused):
    """timm-compatible keywords (``create_transform(**config)`` as in multimodal_infer.py:91).  The reference only ever uses the
    eval pipeline at inference; the random-crop / flip training augmentations of timm are not part of this stage."""
    if is_training:
        raise NotImplementedError("training-time augmentation (timm RandomResizedCropAndInterpolation etc.) is not implemented")
    return ImageTransform(input_size, interpolation, mean, std, crop_pct, device)
