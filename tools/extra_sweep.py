#!/usr/bin/env python3
"""On a GPU box: the random option sweep of tests/test_gpu_parity.py (fp64 path against the oracle, every output bit for
bit, ray counts equal) with another seed and larger images, as a one-off check after changes to the work distribution.
usage: tools/extra_sweep.py [seed] [cases] [max size]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ptoracle as po
from pathtracing_amd import device, host
from conftest import scene_path
from test_gpu_parity import assert_same_frame

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 777
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 48
top = int(sys.argv[3]) if len(sys.argv) > 3 else 220
rng = np.random.default_rng(seed)
names = ["cbox", "car", "pine", "terrace"]
device.init(0)
hs, ds, orc = {}, {}, {}
for n in names:
    p = scene_path(n)
    hs[n] = host.HostScene.from_packed(p); hs[n].build_tree()
    ds[n] = device.DeviceScene(hs[n].desc())
    orc[n] = po.OracleScene(p)
t0 = time.time()
for case in range(cases):
    name = names[case % 4]
    c = hs[name].camera()
    box = orc[name].root_box()
    span = float(np.max(box[3:] - box[:3]))
    eye = (np.array(c["eye"]) + rng.normal(size=3) * 0.08 * span).tolist()
    cam = dict(eye=eye, angle=float(c["angle"] + rng.normal() * 0.8), zoom=float(c["zoom"] * rng.uniform(0.6, 1.6)))
    kw = dict(model=int(rng.integers(0, 2)), gi=bool(rng.integers(0, 2)), rb=bool(rng.integers(0, 2)), spp=int(rng.integers(1, 5)),
              size=int(rng.integers(41, top + 1)), view=int(rng.integers(0, 1000)), seed=int(rng.integers(0, 2 ** 32)))
    g = ds[name].render(**cam, gi_precision=device.GI_F64_EXACT, **kw)
    o = orc[name].render(**cam, rng=po.RNG_PHILOX, threads=16, **kw)
    what = f"case {case}: {name} {cam} {kw}"
    assert_same_frame(g, o, what)
    for k in ("primary", "shadow", "mirror", "gi", "omp_retrace"):
        assert g["stats"][k] == o["counts"][k]["searches"], (what, k)
    print(f"case {case} ok: {name} size {kw['size']} rb {kw['rb']} gi {kw['gi']} model {kw['model']}  ({time.time() - t0:.0f} s)", flush=True)
print(f"{cases} cases, seed {seed}: every output identical to the oracle's")
