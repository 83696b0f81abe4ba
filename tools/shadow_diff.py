#!/usr/bin/env python3
"""Which GI samples of the MAIN model light differently in the fp32 path and the fp64 path (development aid).

For the pixels with the largest difference: re-shoot their GI sample rays with pt_trace in both arithmetics, build the
shading point of each hit in numpy (flat or interpolated normal as util/polygonMesh.cpp:81-88 does), and ask both
searches whether each light is visible from it."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracing_amd import device, host

name = sys.argv[1] if len(sys.argv) > 1 else "pine"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
size = 256
hs = host.HostScene.from_packed(f"scenes/{name}.ptscene")
hs.build_tree(); device.init(0); ds = device.DeviceScene(hs.desc()); c = hs.camera()
d = hs.desc()
def arr(ptr, n, t): return np.ctypeslib.as_array(ptr, shape=(n,)).astype(t).copy()
vert = arr(d.vert, 3 * d.n_vert, np.float64).reshape(-1, 3)
norm = arr(d.norm, 3 * d.n_norm, np.float64).reshape(-1, 3)
tvi = arr(d.tri_vi, 3 * d.n_tri, np.int64).reshape(-1, 3)
tni = arr(d.tri_ni, 3 * d.n_tri, np.int64).reshape(-1, 3)
tmesh = arr(d.tri_mesh, d.n_tri, np.int64); tlocal = arr(d.tri_local, d.n_tri, np.int64)
gid = {(int(m), int(l)): k for k, (m, l) in enumerate(zip(tmesh, tlocal))}
lights = [(d.lights[k].type, np.array(d.lights[k].pos[:])) for k in range(d.n_light)]

kw = dict(gi=True, spp=spp, model=0, size=size, view=0)
a = ds.render(c["eye"], c["angle"], c["zoom"], gi_precision=0, **kw)
b = ds.render(c["eye"], c["angle"], c["zoom"], gi_precision=1, **kw)
diff = np.abs(b["rgb"] - a["rgb"]).max(axis=2).ravel()
top = np.argsort(-diff)[:12]

def unit(v): return v / np.sqrt((v * v).sum())
def coord(N):
    if abs(N[0]) > abs(N[1]): Nt = np.array([N[2], 0, -N[0]]) / np.sqrt(N[0] * N[0] + N[2] * N[2])
    else: Nt = np.array([0, -N[2], N[1]]) / np.sqrt(N[1] * N[1] + N[2] * N[2])
    return Nt, np.cross(N, Nt)

for p in top:
    N, hit = a["gnrm"][p], a["ghit"][p]
    Nt, Nb = coord(N)
    sm = device.hemisphere_samples(0x5EED, 0, int(p), spp)
    dirs = np.array([s[0] * Nb + s[1] * N + s[2] * Nt for s in sm])
    src = np.repeat(hit[None], spp, 0)
    big = np.full(spp, device.FLT_MAX)
    e, f = ds.trace(src, dirs, big, 0), ds.trace(src, dirs, big, 1)
    print(f"pixel {divmod(int(p), size)} diff {diff[p]:.3f} primary tri {a['tri'][p]}")
    for s in range(spp):
        line = f"  sample {s}: exact ({e['mesh'][s]},{e['tri'][s]}) t {e['t'][s]:.4f}  fast ({f['mesh'][s]},{f['tri'][s]}) t {f['t'][s]:.4f}"
        if e["found"][s]:
            g = gid[(int(e["mesh"][s]), int(e["tri"][s]))]
            u, v = e["u"][s], e["v"][s]
            n0, n1, n2 = norm[tni[g]]
            nn = unit((1 - u - v) * n0 + u * n1 + v * n2)
            P = hit + dirs[s] * e["t"][s]
            v0, v1, v2 = vert[tvi[g]]
            ng = np.cross(v1 - v0, v2 - v0)
            line += f"  hit {np.round(P, 3)} n {np.round(nn, 3)} n.ng {np.dot(nn, unit(ng)):.3f} d.n {np.dot(dirs[s], nn):.3f}"
            for kind, pos in lights:
                o = P + 1e-4 * nn
                l = pos - o; dis = np.sqrt((l * l).sum()); l = l / dis
                se = ds.trace(o[None], l[None], np.array([dis]), 0)
                sf = ds.trace(o[None], l[None], np.array([dis]), 1)
                line += f" | light c {np.dot(l, nn):+.3f} exact {'S' if se['found'][0] else 'v'}({se['mesh'][0]},{se['tri'][0]},{se['t'][0]:.4g}) fast {'S' if sf['found'][0] else 'v'}({sf['mesh'][0]},{sf['tri'][0]},{sf['t'][0]:.4g})"
        print(line)
