"""Committed golden vectors (tests/golden/, made by make_golden.py): the oracle must
still reproduce them (CPU), and the HIP path must reproduce them too (GPU)."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import ptoracle as po


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def check_primary(name, r):
    g = np.load(os.path.join(GOLDEN, f"primary_{name}.npz"))
    assert np.array_equal(r["mesh"], g["mesh"].astype(np.int32))
    assert np.array_equal(r["tri"], g["tri"])
    assert np.array_equal(r["t"].reshape(512, 512)[::4, ::4], g["t_grid"])      # bit for bit
    assert np.array_equal(sha(r["bgr"]), g["bgr_sha256"])
    assert r["rgb"].mean() == float(g["rgb_mean"])
    return g


def check_gi(name, tag, r):
    g = np.load(os.path.join(GOLDEN, f"gi_{name}_{tag}.npz"))
    assert np.array_equal(r["rgb"][::8, ::8], g["rgb_grid"])
    assert np.array_equal(sha(r["bgr"]), g["bgr_sha256"])


@pytest.mark.parametrize("name", ["cbox", "car", "pine", "terrace"])
def test_oracle_reproduces_golden_primary(oracle_scenes, name):
    r = oracle_scenes(name).render(model=po.MODEL_MAIN, threads=8)
    g = check_primary(name, r)
    assert len(g["degenerate"]) == {"cbox": 0, "car": 29, "pine": 0, "terrace": 0}[name]


def test_oracle_reproduces_golden_gi(oracle_scenes):
    r = oracle_scenes("cbox").render(model=po.MODEL_MAIN, gi=True, rng=po.RNG_PHILOX, seed=0x5EED, view=3, threads=8, aov=False)
    check_gi("cbox", "main", r)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cbox", "car", "pine", "terrace"])
def test_gpu_reproduces_golden_primary(gpu_scenes, host_scenes, name):
    """Primary-ray hit ids: identical to the reference search on EVERY pixel, hence also on the
    non-degenerate ones, where they equal the brute-force nearest hit."""
    c = host_scenes(name).camera()
    r = gpu_scenes(name).render(c["eye"], c["angle"], c["zoom"], model=0)
    check_primary(name, r)


@pytest.mark.gpu
@pytest.mark.parametrize("name,model,tag", [("cbox", 0, "main"), ("cbox", 1, "omp"), ("car", 0, "main"), ("car", 1, "omp"),
                                            ("pine", 0, "main"), ("pine", 1, "omp"), ("terrace", 0, "main"), ("terrace", 1, "omp")])
def test_gpu_reproduces_golden_gi(gpu_scenes, host_scenes, name, model, tag):
    c = host_scenes(name).camera()
    r = gpu_scenes(name).render(c["eye"], c["angle"], c["zoom"], model=model, gi=True, seed=0x5EED, view=3, gi_precision=0)
    check_gi(name, tag, r)
    # the fp32 GI path against the same golden image: RMSE on the stored grid
    f = gpu_scenes(name).render(c["eye"], c["angle"], c["zoom"], model=model, gi=True, seed=0x5EED, view=3, gi_precision=1)
    g = np.load(os.path.join(GOLDEN, f"gi_{name}_{tag}.npz"))
    rmse = float(np.sqrt(np.mean((f["rgb"][::8, ::8] - g["rgb_grid"]) ** 2)))
    assert rmse < 3e-3, rmse
