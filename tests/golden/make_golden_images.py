#!/usr/bin/env python
"""Golden vectors for the image half of the vision input pipeline.  Run once in the build container:

    python tests/golden/make_golden_images.py

The reference resizes every item image with ``tv.transforms.Resize((R, R))`` on a PIL image (``V/data_utils/dataset.py:68-73``) =
``PIL.Image.resize((R, R), BILINEAR)``; torchvision is not installed here, Pillow (12.2.0) is, so the fixtures are produced by the
very routine torchvision would call.  Stored: small synthetic uint8 inputs of assorted sizes and Pillow's outputs (arrays only)."""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(20260927)
    res = {}
    cases = [(37, 53, 24), (64, 48, 24), (24, 24, 24), (90, 24, 24), (24, 71, 24), (11, 13, 24), (130, 97, 32), (200, 320, 56), (56, 41, 56)]
    for i, (H, W, R) in enumerate(cases):
        # smooth + noisy content so that rounding boundaries are exercised
        yy, xx = np.mgrid[0:H, 0:W]
        base = (127 + 100 * np.sin(yy / 7.0)[..., None] * np.cos(xx / 5.0)[..., None] * np.array([1.0, 0.7, -0.8])).clip(0, 255)
        img = (0.6 * base + 0.4 * rng.integers(0, 256, (H, W, 3))).astype(np.uint8)
        out = np.asarray(Image.fromarray(img).convert("RGB").resize((R, R), Image.BILINEAR))
        res[f"in{i}"], res[f"out{i}"] = img, out
    res["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "g16_image_resize.npz"), **res)
    print("g16:", len(cases), "cases, Pillow", Image.__version__ if hasattr(Image, "__version__") else "")


if __name__ == "__main__":
    main()
