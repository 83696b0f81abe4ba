"""Known answers of the CPU oracle for the kd build and the default views.

Reference-held (what pins the oracle, together with the rendered frames of tests/test_refimage_pins.py):
* kd reference counts "Naive" / "Optimized": project.pdf Table 1 of the reference
  (cbox 12440 / 9073, car 97248 / 67293).

Regression values, NOT reference-held pins: everything else in this file (tree shape, root box, sum of primary hit
distances, ray census, mean linear RGB with the rand() stream, number of rays on which the search differs from a
brute-force nearest hit).  They were recorded during the survey of the reference from a build of its util/*.cpp
behind stand-in OpenCV / Boost headers, which does not count as running the reference; they are kept because
they catch any change of the oracle's behaviour.
"""
import numpy as np
import pytest

from oracle import ptoracle as po

TREE = {
    "cbox": dict(triangles=2190, naive=12440, optimized=9073, nodes=3001, leaves=1501, max_depth=16, root_entries=0),
    "car": dict(triangles=10264, naive=97248, optimized=67293, nodes=15839, leaves=7920, max_depth=25, root_entries=0),
}
ROOT_BOX = {
    "cbox": [-102.01, -0.01, -104.01, 100.01, 159.01, 99.01],
    "car": [-500.01, -0.01, -500.01, 500.01, 1000.01, 500.01],
}
MAX_LEAF_AT_SEARCH = {"cbox": 36, "car": 121}
SUM_T = {"cbox": 46739906.816942871, "car": 187641621.687513083}
DEGENERATE = {"cbox": 0, "car": 29}


@pytest.mark.parametrize("name", ["cbox", "car"])
def test_tree_known_answers(oracle_scenes, name):
    s = oracle_scenes(name)
    st = s.tree_stats()
    for k, v in TREE[name].items():
        assert st[k] == v, (k, st[k], v)
    np.testing.assert_allclose(s.root_box(), ROOT_BOX[name], rtol=0, atol=1e-9)
    ni, _, ent = s.tree_dump()
    assert ni[ni[:, 0] == 1, 2].max() == MAX_LEAF_AT_SEARCH[name]
    assert len(ent) == TREE[name]["optimized"]


@pytest.mark.parametrize("name", ["cbox", "car"])
def test_primary_hits_known_answers(oracle_scenes, name):
    s = oracle_scenes(name)
    r = s.render(model=po.MODEL_MAIN, gi=False, threads=8, brute=True)
    hit = r["mesh"] >= 0
    assert hit.sum() == 512 * 512
    total = 0.0
    for t in r["t"]:            # the reference value was summed sequentially in pixel order
        total += t
    assert abs(total - SUM_T[name]) < 1e-6 * 1e-3, total
    deg = (r["mesh"] != r["bmesh"]) | (r["tri"] != r["btri"])
    assert deg.sum() == DEGENERATE[name]


def test_census_and_mean_rgb_direct_cbox(oracle_scenes):
    r = oracle_scenes("cbox").render(model=po.MODEL_MAIN, gi=False, threads=8, aov=False)
    c = r["counts"]
    assert (c["primary"]["searches"], c["shadow"]["searches"], c["mirror"]["searches"]) == (262144, 479574, 217430)
    assert round(r["rgb"].mean(), 6) == 0.279564


# (scene, model, rb) -> (mean linear RGB, total searches incl. the OMP re-traces); reference rand() stream, seed 1
GI_PINS = [
    ("cbox", po.MODEL_MAIN, False, 0.416105, 3056300),
    ("cbox", po.MODEL_MAIN, True, 0.416109, None),
    ("car", po.MODEL_MAIN, False, 0.201160, 3932160),
    ("cbox", po.MODEL_OMP, False, 0.554872, None),
    ("car", po.MODEL_OMP, False, 0.255335, None),
]


@pytest.mark.parametrize("name,model,rb,mean,searches", GI_PINS)
def test_gi_mean_rgb_with_reference_rand_stream(oracle_scenes, name, model, rb, mean, searches):
    r = oracle_scenes(name).render(model=model, gi=True, rb=rb, rng=po.RNG_LIBC, aov=False)
    assert round(r["rgb"].mean(), 6) == mean
    if searches is not None:
        assert sum(v["searches"] for v in r["counts"].values()) == searches
    # the visibility re-trace of the OMP model never reports an occluder
    assert r["dummy_true"] == 0


def test_counter_rng_statistics_match_rand_stream(oracle_scenes):
    """The counter RNG replaces rand(): the image mean must agree statistically."""
    s = oracle_scenes("cbox")
    a = s.render(model=po.MODEL_MAIN, gi=True, rng=po.RNG_PHILOX, threads=8, aov=False)["rgb"].mean()
    assert abs(a - 0.416105) < 2e-3


def test_sincos_2pi_accuracy():
    rs = np.random.default_rng(1).random(2000)
    for r in list(rs) + [0.0, 0.125, 0.25, 0.5, 0.75, 0.999999999]:
        c, s = po.sincos_2pi(float(r))
        # numpy rounds the argument 2*pi*r itself (half an ulp at 6.28 is 4.4e-16)
        assert abs(c - np.cos(2 * np.pi * r)) < 1.5e-15 and abs(s - np.sin(2 * np.pi * r)) < 1.5e-15


def test_hemisphere_samples_are_unit_and_upper():
    v = po.hemisphere_samples(0x5EED, 3, 12345, 4096)
    np.testing.assert_allclose(np.linalg.norm(v, axis=1), 1.0, atol=1e-15)
    assert (v[:, 1] >= 0).all() and (v[:, 1] < 1).all()
    # uniform hemisphere: E[cos theta] = 1/2
    assert abs(v[:, 1].mean() - 0.5) < 0.02
