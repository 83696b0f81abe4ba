#!/usr/bin/env python3
"""Converged -gi fixtures from the CPU oracle: tests/golden/converged_<scene>_<model>.npz.

128 x 128, 512 samples per pixel, counter RNG, the scene's own camera:
  rgb         linear image of seed 0x5EED / view 3 (float32)
  noise_rmse  RMSE between that image and the oracle's image of an independent sample set (seed 0xBEEF) at the
              same sample count: the yardstick for "converged images match at equal spp" -- a renderer with the
              right expectation and its own samples lands within about this distance; the bar used is 3 x.
  mean        mean of the fp64 image
Usage: python tests/golden/make_converged.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ptoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SIZE, SPP = 128, 512


def main():
    from conftest import scene_path
    for name in ("cbox", "car"):
        s = po.OracleScene(scene_path(name))
        for model, tag in ((po.MODEL_MAIN, "main"), (po.MODEL_OMP, "omp")):
            kw = dict(model=model, gi=True, spp=SPP, size=SIZE, rng=po.RNG_PHILOX, view=3, threads=8, aov=False)
            a = s.render(seed=0x5EED, **kw)["rgb"]
            b = s.render(seed=0xBEEF, **kw)["rgb"]
            noise = float(np.sqrt(np.mean((a - b) ** 2)))
            np.savez_compressed(os.path.join(OUT, f"converged_{name}_{tag}.npz"), rgb=a.astype(np.float32),
                                noise_rmse=np.float64(noise), mean=np.float64(a.mean()), spp=np.int32(SPP))
            print(name, tag, "noise rmse", noise, "mean", a.mean(), flush=True)


if __name__ == "__main__":
    main()
