#!/usr/bin/env python3
"""Measured agreement of the fp32 GI path with the fp64 path / brute force, the numbers the test thresholds are set from."""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
import ptoracle as po
from pathtracing_amd import device, host
device.init(0)
def rays(n, seed, box):
    r = np.random.default_rng(seed)
    lo, hi = np.array(box[:3]), np.array(box[3:])
    src = lo + (hi - lo) * r.random((n, 3))
    d = r.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1)[:, None]
    return src, d
for name in ("cbox", "car", "atrium20k"):
    if name == "atrium20k":
        hs = host.HostScene.atrium(20000, 7); hs.save_packed("/tmp/a20k.ptscene"); orc = po.OracleScene("/tmp/a20k.ptscene")
    else:
        hs = host.HostScene.from_packed(os.path.join(root, "scenes", name + ".ptscene")); orc = po.OracleScene(os.path.join(root, "scenes", name + ".ptscene"))
    hs.build_tree(); ds = device.DeviceScene(hs.desc()); c = hs.camera()
    src, d = rays(20000, 11, orc.root_box()); tm = np.full(len(src), device.FLT_MAX)
    g = ds.trace(src, d, tm, mode=1); o = orc.trace(src, d, tm, mode=1)
    both = (g["found"] == 1) & (o["found"] == 1)
    same = both & (g["mesh"] == o["mesh"]) & (g["tri"] == o["tri"])
    rel = np.abs(g["t"][same] - o["t"][same]) / o["t"][same]
    print(f"{name}: found mismatch {(g['found'] != o['found']).mean():.2e}  same triangle {same.sum() / both.sum():.6f}  rel t q999 {np.quantile(rel, .999):.2e} max {rel.max():.2e}")
    for model in (0, 1):
        for spp, size in ((8, 512), (2, 160)):
            a = ds.render(c["eye"], c["angle"], c["zoom"], model=model, gi=True, spp=spp, size=size, gi_precision=0)
            b = ds.render(c["eye"], c["angle"], c["zoom"], model=model, gi=True, spp=spp, size=size, gi_precision=1)
            dd = a["rgb"] - b["rgb"]
            print(f"   model {model} spp {spp} size {size}: rmse {np.sqrt(np.mean(dd**2)):.2e}  |mean diff| {abs(a['rgb'].mean() - b['rgb'].mean()):.2e}  "
                  f"bytes differing {(a['bgr'] != b['bgr']).mean():.2e}  pixels |d|>0.02 {(np.abs(dd).max(axis=2) > 0.02).sum()}  shadow {a['stats']['shadow']} {b['stats']['shadow']}")
