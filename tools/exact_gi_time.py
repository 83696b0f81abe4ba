#!/usr/bin/env python3
"""Time the fp64 GI pass (development aid): tools/exact_gi_time.py [scene] [spp]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracing_amd import device, host
name = sys.argv[1] if len(sys.argv) > 1 else "car"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
hs = host.HostScene.atrium() if name == "atrium" else host.HostScene.from_packed(f"scenes/{name}.ptscene")
hs.build_tree(); device.init(0); ds = device.DeviceScene(hs.desc()); c = hs.camera()
for model in (0, 1):
    for rb in (False, True):
        ds.render(c["eye"], c["angle"], c["zoom"], gi=True, rb=rb, spp=spp, model=model, gi_precision=0, aov=False)
        t = time.time(); r = ds.render(c["eye"], c["angle"], c["zoom"], gi=True, rb=rb, spp=spp, model=model, gi_precision=0, aov=False); dt = time.time() - t
        print(f"{name} model {model} rb {rb} spp {spp}: {dt*1e3:.1f} ms total, gi kernels {r['stats']['ms_gi_trace']:.1f} ms")
