import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracing_amd import device, host
device.init(0)
for name in ("car", "atrium"):
    hs = host.HostScene.atrium(262267, 12345) if name == "atrium" else host.HostScene.from_packed(f"scenes/{name}.ptscene")
    hs.build_tree(); ds = device.DeviceScene(hs.desc()); c = hs.camera()
    offs = [-0.1, 0.0, 0.1]
    idx = np.arange(512 * 512).reshape(64, 8, 64, 8).transpose(0, 2, 1, 3).reshape(-1)      # tile order
    subs = [device.camera_rays(c["eye"], c["angle"], c["zoom"], x=offs[s % 3], y=offs[s // 3])[idx] for s in range(9)]
    d = np.stack(subs, axis=1).reshape(-1, 3)                                                # [pixel][sub]
    src = np.repeat(np.array(c["eye"])[None], len(d), 0)
    big = np.full(len(d), device.FLT_MAX)
    for rep in range(2):
        ds.trace(src, d, big, 0)
        ds.trace(src, d, big, 2)
        ds.render(c["eye"], c["angle"], c["zoom"], model=1, rb=True, aov=False)
    ds.close()
