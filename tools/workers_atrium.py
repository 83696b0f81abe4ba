#!/usr/bin/env python3
"""On a GPU box: views/s of pt_render_views on the atrium at the reference's -rm settings (-gi -rb, mainOMP model, 16 spp) with
1 ... 8 view workers per GPU (PT_TUNE_WORKERS)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pathtracing_amd import device

hs, st, views = bench.load_host_scene(sys.argv[1] if len(sys.argv) > 1 else "atrium")
device.init(0)
ds = device.DeviceScene(hs.desc())
opt = dict(gi=True, rb=True, model=device.MODEL_OMP, spp=16, gi_precision=device.GI_F32_FAST, size=512)
n = 128
for w in (1, 2, 3, 4, 6, 8):
    os.environ["PT_TUNE_WORKERS"] = str(w)
    ds.render_views(views[:16], view0=0, view_stride=1, **opt)          # the workers and their workspaces
    t0 = time.perf_counter()
    ds.render_views(views[:n], view0=0, view_stride=1, **opt)
    dt = time.perf_counter() - t0
    print(f"{w} workers: {n / dt:7.1f} views/s  {dt / n * 1e3:.3f} ms per view", flush=True)
