#!/usr/bin/env python3
"""Small JPEG and PNG files of every kind the texture decoder accepts, with the pixels Pillow decodes them to
(libjpeg-turbo: slow-integer IDCT, fancy upsampling -- the defaults OpenCV's imread also runs with; PNG is lossless,
alpha dropped as cv::imread(IMREAD_COLOR) does).

    python tests/golden/make_jpeg_fixtures.py        -> tests/golden/jpeg/*.jpg, *.png, tests/golden/jpeg/expected.npz
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
# PNG: every colour type the decoder takes, stored / fast / best compression (stored, fixed and dynamic Huffman blocks)
png_cases = {
    "png_rgba": ("RGBA", dict(compress_level=9)),
    "png_rgb": ("RGB", dict(compress_level=1)),
    "png_grey": ("L", dict(compress_level=6)),
    "png_greya": ("LA", dict(compress_level=9)),
    "png_palette": ("P", dict(compress_level=9)),
    "png_stored": ("RGB", dict(compress_level=0)),
    "png_flat": ("RGB", dict(compress_level=9)),            # one colour: long matches, tiny dynamic trees
}
for name, (mode, kw) in png_cases.items():
    img = picture(45, 37)
    if name == "png_flat":
        img[:] = (12, 200, 77)
    im = Image.fromarray(img, "RGB")
    if mode == "RGBA":
        im.putalpha(Image.fromarray((img[:, :, 0] // 2 + 64).astype(np.uint8), "L"))
    elif mode == "LA":
        im = im.convert("L"); im.putalpha(Image.fromarray(img[:, :, 1], "L"))
    elif mode == "P":
        im = im.quantize(colors=50)
    elif mode != "RGB":
        im = im.convert(mode)
    path = os.path.join(here, name + ".png")
    im.save(path, format="PNG", **kw)
    expected[name] = np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1].copy()
# A deflate stream in which stored blocks FOLLOW Huffman blocks at arbitrary bit positions: a sync flush ends the
# current block and emits an empty stored block, which starts at the next byte boundary.  Written by hand (filter
# type 0 on every row), because an encoder never produces this by itself.
import struct
import zlib
img = picture(45, 37)
co = zlib.compressobj(6)
stream = b""
for y in range(45):
    stream += co.compress(b"\0" + img[y].tobytes())
    if y % 5 == 4:
        stream += co.flush(zlib.Z_FULL_FLUSH if y == 24 else zlib.Z_SYNC_FLUSH)
stream += co.flush()
def chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data))
path = os.path.join(here, "png_flushed.png")
open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 37, 45, 8, 2, 0, 0, 0)) + chunk(b"IDAT", stream) +
                       chunk(b"IEND", b""))
expected["png_flushed"] = np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1].copy()
assert np.array_equal(expected["png_flushed"], img[:, :, ::-1])
np.savez_compressed(os.path.join(here, "expected.npz"), **expected)
print({k: v.shape for k, v in expected.items()})
