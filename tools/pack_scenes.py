#!/usr/bin/env python3
"""Pack reference scenes into self-contained PTSCENE1 files under scenes/.

Run in a checkout that has the reference's scene directory; the packed files
are what tests, smoke() and bench.py read (the GPU box has no reference tree).
JPEG textures are decoded with PIL into temporary binary PPMs, which the host
reader (pth_scene_load_dir) then reads through its texture_dir argument.

usage: tools/pack_scenes.py [--ref /root/reference/scenes] [scene ...]
"""
import argparse
import os
import shutil
import sys
import tempfile

from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracing_amd import host  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference/scenes")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes"))
    ap.add_argument("scenes", nargs="*", default=["cbox", "car"])
    a = ap.parse_args()
    os.makedirs(os.path.join(a.out, "paths"), exist_ok=True)
    for name in a.scenes:
        sdir = os.path.join(a.ref, name)
        with tempfile.TemporaryDirectory() as tmp:
            for fn in os.listdir(sdir):
                if fn.lower().endswith(".jpg"):
                    Image.open(os.path.join(sdir, fn)).convert("RGB").save(os.path.join(tmp, fn[:-4] + ".ppm"))
            sc = host.HostScene.from_dir(sdir, tmp)
            dst = os.path.join(a.out, name + ".ptscene")
            sc.save_packed(dst)
            st = sc.build_tree()
            print(name, os.path.getsize(dst), "bytes", st)
        p = os.path.join(a.ref, "paths", name + "path.txt")
        if os.path.exists(p):
            shutil.copyfile(p, os.path.join(a.out, "paths", name + "path.txt"))


if __name__ == "__main__":
    main()
