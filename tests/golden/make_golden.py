#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the CPU oracle.

The oracle is pinned to the reference's known answers by tests/test_oracle_pins.py;
these files freeze its outputs so that the GPU tests can also be checked against
data that was committed, not only against code that runs next to them.

  primary_<scene>.npz   default view, main.cpp model, no -gi/-rb:
                        mesh, tri      (mesh, triangle) of the reference search per pixel
                        degenerate     pixels where a brute-force nearest hit differs from it
                        t_grid         hit distance on the stride-4 pixel grid (fp64)
                        bgr_sha256     digest of the BGR8 frame
                        rgb_mean       mean linear RGB
  gi_<scene>_<model>.npz  -gi with the counter RNG (seed 0x5EED, view 3), reference sample count:
                        rgb_grid       linear RGB on the stride-8 pixel grid (fp64), bgr_sha256
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ptoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def main():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import scene_path      # scenes/<name>.ptscene, or tests/golden/<name>/ packed by the host reader
    for name in ("cbox", "car", "pine", "terrace"):
        s = po.OracleScene(scene_path(name))
        r = s.render(model=po.MODEL_MAIN, threads=8, brute=True)
        deg = (r["mesh"] != r["bmesh"]) | (r["tri"] != r["btri"])
        np.savez_compressed(os.path.join(OUT, f"primary_{name}.npz"), mesh=r["mesh"].astype(np.int8), tri=r["tri"],
                            degenerate=np.flatnonzero(deg).astype(np.int32), t_grid=r["t"].reshape(512, 512)[::4, ::4].copy(),
                            bgr_sha256=sha(r["bgr"]), rgb_mean=np.float64(r["rgb"].mean()))
        print(name, "primary: degenerate", int(deg.sum()), "mean", r["rgb"].mean())
        for model, tag in ((po.MODEL_MAIN, "main"), (po.MODEL_OMP, "omp")):
            g = s.render(model=model, gi=True, rng=po.RNG_PHILOX, seed=0x5EED, view=3, threads=8, aov=False)
            np.savez_compressed(os.path.join(OUT, f"gi_{name}_{tag}.npz"), rgb_grid=g["rgb"][::8, ::8].copy(),
                                bgr_sha256=sha(g["bgr"]), rgb_mean=np.float64(g["rgb"].mean()))
            print(name, tag, "gi mean", g["rgb"].mean())


if __name__ == "__main__":
    main()
