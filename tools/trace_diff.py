#!/usr/bin/env python3
"""GI-like rays from the G-buffer traced by the fp64 and the fp32 search; classify the differences (development aid)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracing_amd import device, host
tris = int(sys.argv[1]) if len(sys.argv) > 1 else 262267
hs = host.HostScene.atrium(tris, 12345); hs.build_tree(); device.init(0); ds = device.DeviceScene(hs.desc()); c = hs.camera()
a = ds.render(c["eye"], c["angle"], c["zoom"], size=256)
ok = (a["ghit"][:, 0] != -99999) & np.isfinite(a["gnrm"]).all(axis=1)
o = a["ghit"][ok]; n = a["gnrm"][ok]; tri0 = a["tri"][ok]; mesh0 = a["mesh"][ok]
rng = np.random.default_rng(1)
reps = 8
o = np.repeat(o, reps, 0); n = np.repeat(n, reps, 0); tri0 = np.repeat(tri0, reps); mesh0 = np.repeat(mesh0, reps)
d = rng.normal(size=o.shape); d /= np.linalg.norm(d, axis=1)[:, None]
s = np.sign((d * n).sum(1)); d *= s[:, None]
tm = np.full(len(o), 1e18)
e = ds.trace(o, d, tm, 0); f = ds.trace(o, d, tm, 1)
print("rays", len(o), "found exact", e["found"].mean(), "fast", f["found"].mean())
same = (e["found"] == f["found"]) & ((e["found"] == 0) | ((e["mesh"] == f["mesh"]) & (e["tri"] == f["tri"])))
print("mismatch fraction", 1 - same.mean())
bad = np.where(~same)[0]
both = bad[(e["found"][bad] == 1) & (f["found"][bad] == 1)]
print(" exact hit/fast miss", ((e["found"][bad] == 1) & (f["found"][bad] == 0)).sum(),
      " exact miss/fast hit", ((e["found"][bad] == 0) & (f["found"][bad] == 1)).sum(), " both hit, different", len(both))
dt = f["t"][both] - e["t"][both]
print(" both: fast nearer", (dt < 0).sum(), "fast farther", (dt > 0).sum(), " |dt| quantiles", np.quantile(np.abs(dt), [.1, .5, .9, .99]))
print(" both: exact t quantiles", np.quantile(e["t"][both], [.1, .5, .9]), " fast t quantiles", np.quantile(f["t"][both], [.1, .5, .9]))
print(" fast t < 1e-2:", (f["t"][both] < 1e-2).sum(), " exact t < 1e-2:", (e["t"][both] < 1e-2).sum())
print(" self hits (fast hit == origin tri):", ((f["mesh"][bad] == mesh0[bad]) & (f["tri"][bad] == tri0[bad])).sum(),
      " exact self hits:", ((e["mesh"][bad] == mesh0[bad]) & (e["tri"][bad] == tri0[bad])).sum())
um, cnt = np.unique(mesh0[bad], return_counts=True)
print(" origin meshes of mismatches", dict(zip(um.tolist(), cnt.tolist())), " of all", dict(zip(*[x.tolist() for x in np.unique(mesh0, return_counts=True)])))
for k in bad[:12]:
    print("  o", o[k], "d", d[k], "exact", e["found"][k], e["mesh"][k], e["tri"][k], e["t"][k], "fast", f["found"][k], f["mesh"][k], f["tri"][k], f["t"][k], "origin", mesh0[k], tri0[k])
