#!/usr/bin/env python3
"""fp32 GI path against the fp64 path at equal spp (development aid)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracing_amd import device, host
name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tris = int(sys.argv[3]) if len(sys.argv) > 3 else 262267
hs = host.HostScene.atrium(tris, 12345) if name == "atrium" else host.HostScene.from_packed(f"scenes/{name}.ptscene")
hs.build_tree(); device.init(0); ds = device.DeviceScene(hs.desc()); c = hs.camera()
for model in (0, 1):
    a = ds.render(c["eye"], c["angle"], c["zoom"], gi=True, spp=spp, model=model, gi_precision=0, size=256)
    b = ds.render(c["eye"], c["angle"], c["zoom"], gi=True, spp=spp, model=model, gi_precision=1, size=256)
    d = b["rgb"] - a["rgb"]
    ok = np.isfinite(d)
    print(f"{name} model {model} spp {spp}: mean exact {np.nanmean(a['rgb']):.6f} fast {np.nanmean(b['rgb']):.6f} "
          f"rmse {np.sqrt(np.mean(d[ok]**2)):.3e} max|d| {np.abs(d[ok]).max():.3f} pixels |d|>0.02: {(np.abs(d).max(axis=2) > 0.02).sum()} "
          f"bgr differing bytes {(a['bgr'] != b['bgr']).mean():.4f}")
    print("   non-finite pixels exact", (~np.isfinite(a['rgb'])).any(axis=2).sum(), "fast", (~np.isfinite(b['rgb'])).any(axis=2).sum(),
          " mean over common finite: exact", a['rgb'][ok].mean(), "fast", b['rgb'][ok].mean(), " stats", a['stats'], b['stats'])
    top = np.argsort(-np.abs(np.nan_to_num(d)).max(axis=2).ravel())[:14]
    for p in top:
        i, j = divmod(int(p), 256)
        print(f"   px ({i},{j}) tri {a['tri'][p]} t {a['t'][p]:.2f} hit {a['ghit'][p]} nrm {a['gnrm'][p]} exact {a['rgb'][i, j]} fast {b['rgb'][i, j]}")
    big = np.abs(np.nan_to_num(d)).max(axis=2).ravel() > 0.02
    ut, cnt = np.unique(a['tri'][big], return_counts=True)
    o = np.argsort(-cnt)[:12]
    print("   tris with most differing pixels:", list(zip(ut[o].tolist(), cnt[o].tolist())))
