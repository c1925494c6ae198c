"""TEST INFRASTRUCTURE (oracle): CPU restatement of the image half of the vision input pipeline.

Reference: ``V/data_utils/dataset.py:68-73,86-99`` -- ``Resize((R, R))`` -> ``ToTensor`` -> ``Normalize(0.5, 0.5)`` applied to
``Image.fromarray(LMDB_Image.get_image())``.  On a PIL image torchvision's ``Resize`` is ``Image.resize((R, R), BILINEAR)``
(torchvision/transforms/_functional_pil.py ``resize``; torchvision itself is not installed here), i.e. Pillow's two-pass
convolution resampler (Pillow ``src/libImaging/Resample.c``: ``precompute_coeffs`` / ``normalize_coeffs_8bpc`` /
``ImagingResampleHorizontal_8bpc`` / ``ImagingResampleVertical_8bpc``), third-party code that is not under /root/reference; its
published algorithm is restated below and pinned against Pillow 12.2.0 itself through ``tests/golden/g16_image_resize.npz``
(tests/golden/make_golden_images.py).  Integer arithmetic: results are compared bit for bit."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bilinear(x: float) -> float:
    x = -x if x < 0.0 else x
    return 1.0 - x if x < 1.0 else 0.0


def coeffs(in_size: int, out_size: int):
    """``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the full box [0, in_size): per output index (first input index,
    tap count) and the fixed-point taps."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale                      # bilinear: support 1
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bilinear((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255)


def pil_bilinear_resize(img: np.ndarray, R: int) -> np.ndarray:
    """uint8 [H, W, C] -> uint8 [R, R, C]: horizontal pass, then vertical pass on its uint8 result (``ImagingResample``)."""
    H, W, C = img.shape
    src = img.astype(np.int64)
    if W != R:
        b, k = coeffs(W, R)
        tmp = np.zeros((H, R, C), dtype=np.int64)
        for xx in range(R):
            x0, n = int(b[xx, 0]), int(b[xx, 1])
            acc = (1 << (PRECISION_BITS - 1)) + (src[:, x0:x0 + n, :] * k[xx, :n][None, :, None]).sum(1)
            tmp[:, xx, :] = _clip8(acc)
        src = tmp
    if H != R:
        b, k = coeffs(H, R)
        out = np.zeros((R, src.shape[1], C), dtype=np.int64)
        for yy in range(R):
            y0, n = int(b[yy, 0]), int(b[yy, 1])
            acc = (1 << (PRECISION_BITS - 1)) + (src[y0:y0 + n, :, :] * k[yy, :n][:, None, None]).sum(0)
            out[yy] = _clip8(acc)
        src = out
    return src.astype(np.uint8)


def to_tensor_normalize(img_u8: np.ndarray) -> np.ndarray:
    """``ToTensor`` + ``Normalize((0.5,)*3, (0.5,)*3)``: uint8 HWC -> float32 CHW."""
    x = img_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    return (x - np.float32(0.5)) / np.float32(0.5)
