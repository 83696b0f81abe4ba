#!/usr/bin/env python3
"""Small JPEG files of every kind the texture decoder accepts, with the pixels Pillow (libjpeg-turbo: slow-integer
IDCT, fancy upsampling -- the defaults OpenCV's imread also runs with) decodes them to.

    python tests/golden/make_jpeg_fixtures.py        -> tests/golden/jpeg/*.jpg, tests/golden/jpeg/expected.npz
"""
import os

import numpy as np
from PIL import Image

here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jpeg")
os.makedirs(here, exist_ok=True)
rng = np.random.default_rng(2024)


def picture(h, w):
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 120 * np.sin(x / 5.0) * np.cos(y / 7.0), 255.0 * x / max(w - 1, 1), 255.0 * y / max(h - 1, 1)], axis=2)
    base += rng.normal(0, 12, base.shape)
    base[h // 3: h // 2, w // 4: w // 2] = (250, 10, 30)           # a hard-edged patch: large chroma steps
    return np.clip(base, 0, 255).astype(np.uint8)


cases = {
    "c420_odd": dict(size=(37, 29), subsampling="4:2:0", quality=90),
    "c420_mcu": dict(size=(64, 48), subsampling="4:2:0", quality=75),
    "c422": dict(size=(41, 23), subsampling="4:2:2", quality=85),
    "c444": dict(size=(33, 31), subsampling="4:4:4", quality=95),
    "c420_restart": dict(size=(50, 70), subsampling="4:2:0", quality=80, restart_marker_blocks=3),
    "grey": dict(size=(45, 27), grey=True, quality=90),
    "c420_q30": dict(size=(40, 40), subsampling="4:2:0", quality=30),
    "tiny": dict(size=(1, 1), subsampling="4:2:0", quality=90),
}
expected = {}
for name, c in cases.items():
    h, w = c.pop("size")
    grey = c.pop("grey", False)
    img = picture(h, w)
    im = Image.fromarray(img[:, :, 0], "L") if grey else Image.fromarray(img, "RGB")
    path = os.path.join(here, name + ".jpg")
    im.save(path, format="JPEG", **c)
    expected[name] = np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1].copy()     # BGR, as cv::imread stores it
np.savez_compressed(os.path.join(here, "expected.npz"), **expected)
print({k: v.shape for k, v in expected.items()})
