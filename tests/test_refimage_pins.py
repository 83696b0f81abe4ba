"""The oracle -- and the HIP path -- against frames the REFERENCE ITSELF rendered.

tests/golden/refimages_cbox.npz holds every 16th pixel of every 16th row of 2 x 304 of the 512 x 512 frames the
reference ships under ml/data/CBOX (made by tests/golden/make_refimage_pins.py, which also documents how the
frames were matched to cameras):

  dif[n]     CBOXdif/cbox<k>.jpg      main.cpp shading model, no -gi, view angle 22.5 deg * (k % 16)
  gi4096[n]  CBOX4096ssp/cbox<k>.jpg  mainOMP.cpp model with -gi, camera of mainOMP.cpp:164-165

with k = k[n], path position k // 16 of scenes/paths/cboxpath.txt.  A 32 x 32 render samples exactly those pixels.
The frames are JPEGs: a pixel is off by about 0.8 / 255 RMS from what the renderer wrote.  Bars (values in 1 / 255):

  * no -gi: RMSE <= 1.5 on every frame (measured: 0.5 .. 1.16, i.e. JPEG noise), for the oracle and the HIP path;
  * -gi: the reference used its libc rand() stream at 4096 samples, the checkers the counter RNG at 1024, so the
    comparison is statistical: RMSE <= 4.5 per frame (measured 1.4 .. 3.6: sampling noise of 1024 samples plus JPEG),
    frame means within 0.6 (measured <= 0.31), and the mean over all frames of the frame-mean difference within 0.1
    (measured -0.03: no bias in the GI estimate).

  * -gi at FOUR samples (gi4[n], CBOX4ssp/cbox<k>.jpg, same model and camera): a 4-sample frame is mostly noise, so
    what is compared is how far it lies from the reference's converged frame -- RMSE(oracle at 4 samples, gi4096[n]) /
    RMSE(gi4[n], gi4096[n]) is 0.89 .. 1.37 per frame, 1.08 on average (the JPEG takes a little of the reference's noise
    away); 2 samples give 1.40 on average, 8 samples 0.83 -- and the frame means (within 3.7 per frame, 0.07 on average).
    Bars: ratio 0.8 .. 1.5 per frame and 0.98 .. 1.18 on average, means within 4.5 per frame and 0.3 on average.  This
    pins the sample count N and the weight 2 * mclr * GI / N (mainOMP.cpp:132-134) at a second N;
  * four CBOXdif frames kept whole (tests/golden/refimages_cbox_full.npz, 512 x 512, position 5, one view per wall):
    RMSE <= 1.5 as above, and the same render moved by one pixel in any direction is at least twice as far from
    the reference's frame on the three views that show edges -- a one-pixel error cannot hide.

These are the reference-held pins of search, intersection, shading (both models: shadows, mirror bounce, textures),
GI sampling and weights, and the 8-bit quantisation.

What no output the reference holds can pin (the oracle's reading of the source is all there is): the GI pass of
main.cpp (mpnt += 2 * mclr * GI / N with the 0.6 weight of a depth -1 return, main.cpp:77,129-134), the -rb bundles of
either front end (main.cpp:116-123, mainOMP.cpp:117-127), and every scene but cbox (the reference's CAR frames were
rendered from another arrangement of the scene, and it holds no frame of pine).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from oracle import ptoracle as po

GI_SPP = 1024


@pytest.fixture(scope="module")
def pins():
    z = np.load(os.path.join(GOLDEN, "refimages_cbox.npz"))
    lines = open(os.path.join(ROOT, "scenes", "paths", "cboxpath.txt")).read().split("\n")
    pos = [tuple(float(x) for x in lines[1 + i].split()) for i in range(int(lines[0]))]
    full = np.load(os.path.join(GOLDEN, "refimages_cbox_full.npz"))
    return {"k": z["k"], "dif": z["dif"], "gi": z["gi4096"], "gi4": z["gi4"], "pos": pos, "full_k": full["k"], "full_dif": full["dif"]}


def absolute_angle(j):
    return np.pi / 180.0 * (22.5 * j)


def shipped_angle(p, j):
    a, c = p[0], p[2]                                  # mainOMP.cpp:164-165
    return -np.arccos(-(a / np.sqrt(a * a + c * c))) + np.pi / 180.0 * (22.5 * j)


def rmse(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def test_fixture_is_what_the_generator_describes(pins):
    assert pins["dif"].shape == pins["gi"].shape == (304, 32, 32, 3) and pins["dif"].dtype == np.uint8
    assert list(pins["k"]) == list(range(16, 320)) and len(pins["pos"]) == 20
    assert pins["gi4"].shape == (304, 32, 32, 3) and pins["full_dif"].shape == (4, 512, 512, 3) and list(pins["full_k"]) == [80, 84, 88, 92]
    for n, k in enumerate(pins["full_k"]):                      # the whole frames contain the subsampled ones
        assert np.array_equal(pins["full_dif"][n][::16, ::16], pins["dif"][int(k) - 16])


def test_oracle_direct_frames_equal_the_reference_frames(pins, oracle_scenes):
    orc = oracle_scenes("cbox")
    worst = 0.0
    for n, k in enumerate(pins["k"]):
        p, j = divmod(int(k), 16)
        o = orc.render(eye=list(pins["pos"][p]), angle=absolute_angle(j), zoom=1.0, model=po.MODEL_MAIN, gi=False, size=32,
                       threads=4, aov=False)["bgr"]
        e = rmse(o, pins["dif"][n])
        worst = max(worst, e)
        assert e <= 1.5, (int(k), e)
    assert worst > 0.3          # JPEG frames: an exact match would mean the fixture was made from the oracle


def test_the_pin_tells_the_models_and_cameras_apart(pins, oracle_scenes):
    """Frame 21 (position 1, angle index 5) looks at mirror walls: the mainOMP.cpp model (mirror bounce weighted 1, no
    shadows) is 15 / 255 away from the reference's frame, the shipped camera formula looks somewhere else entirely."""
    orc = oracle_scenes("cbox")
    n, p, j = 21 - 16, 1, 5
    kw = dict(eye=list(pins["pos"][p]), zoom=1.0, gi=False, size=32, threads=4, aov=False)
    assert rmse(orc.render(angle=absolute_angle(j), model=po.MODEL_MAIN, **kw)["bgr"], pins["dif"][n]) <= 1.5
    assert rmse(orc.render(angle=absolute_angle(j), model=po.MODEL_OMP, **kw)["bgr"], pins["dif"][n]) > 8.0
    assert rmse(orc.render(angle=shipped_angle(pins["pos"][p], j), model=po.MODEL_MAIN, **kw)["bgr"], pins["dif"][n]) > 8.0
    assert rmse(orc.render(angle=absolute_angle(j), model=po.MODEL_MAIN, **kw)["bgr"], pins["gi"][n]) > 8.0


def test_oracle_gi_frames_match_the_reference_converged_frames(pins, oracle_scenes):
    """One frame per path position (19 of the 304; the GPU test below takes them all)."""
    orc = oracle_scenes("cbox")
    diffs = []
    for p in range(1, 20):
        j = (5 * p) % 16
        k = 16 * p + j
        o = orc.render(eye=list(pins["pos"][p]), angle=shipped_angle(pins["pos"][p], j), zoom=1.0, model=po.MODEL_OMP, gi=True,
                       spp=GI_SPP, size=32, threads=8, aov=False, view=k, rng=po.RNG_PHILOX)["bgr"]
        r = pins["gi"][k - 16]
        assert rmse(o, r) <= 4.5, (k, rmse(o, r))
        diffs.append(float(o.mean()) - float(r.mean()))
        assert abs(diffs[-1]) <= 0.6, (k, diffs[-1])
    assert abs(np.mean(diffs)) <= 0.15, np.mean(diffs)           # 19 frames; the bound for all 304 is 0.1


def four_sample_check(render4, pins, frames):
    """render4(k, p, j) -> 32 x 32 BGR frame at 4 samples.  Returns (noise ratios, mean differences) over `frames`."""
    ratios, diffs = [], []
    for k in frames:
        p, j = divmod(int(k), 16)
        o = render4(int(k), p, j)
        conv, r4 = pins["gi"][k - 16], pins["gi4"][k - 16]
        ratios.append(rmse(o, conv) / rmse(r4, conv))
        diffs.append(float(o.mean()) - float(r4.mean()))
        assert 0.8 <= ratios[-1] <= 1.5, (k, ratios[-1])
        assert abs(diffs[-1]) <= 4.5, (k, diffs[-1])
    return ratios, diffs


def test_oracle_four_sample_frames_scatter_like_the_reference_four_sample_frames(pins, oracle_scenes):
    """Every frame of the set (a 32 x 32 frame at 4 samples is 8 000 searches)."""
    orc = oracle_scenes("cbox")

    def render(n):
        return lambda k, p, j: orc.render(eye=list(pins["pos"][p]), angle=shipped_angle(pins["pos"][p], j), zoom=1.0, model=po.MODEL_OMP,
                                          gi=True, spp=n, size=32, threads=8, aov=False, view=k, rng=po.RNG_PHILOX)["bgr"]
    ratios, diffs = four_sample_check(render(4), pins, range(16, 320))
    assert 0.98 <= np.mean(ratios) <= 1.18, np.mean(ratios)
    assert abs(np.mean(diffs)) <= 0.3, np.mean(diffs)
    # the figure tells sample counts apart: half or twice the samples fall outside the bar for the average
    for n, lo, hi in ((2, 1.25, 1.6), (8, 0.7, 0.93)):
        r = []
        for k in range(16, 320, 4):
            p, j = divmod(k, 16)
            r.append(rmse(render(n)(k, p, j), pins["gi"][k - 16]) / rmse(pins["gi4"][k - 16], pins["gi"][k - 16]))
        assert lo <= np.mean(r) <= hi, (n, np.mean(r))


def test_oracle_whole_frames_and_the_one_pixel_shift(pins, oracle_scenes):
    orc = oracle_scenes("cbox")
    for n, k in enumerate(pins["full_k"]):
        p, j = divmod(int(k), 16)
        o = orc.render(eye=list(pins["pos"][p]), angle=absolute_angle(j), zoom=1.0, model=po.MODEL_MAIN, gi=False, size=512,
                       threads=8, aov=False)["bgr"]
        e = rmse(o, pins["full_dif"][n])
        assert e <= 1.5, (int(k), e)
        if k != 80:                                    # view 80 faces a bare wall: nothing in it moves with a pixel
            for sh, ax in ((1, 0), (-1, 0), (1, 1), (-1, 1)):
                assert rmse(np.roll(o, sh, axis=ax), pins["full_dif"][n]) >= 2.0 * e, (int(k), sh, ax)


@pytest.mark.gpu
def test_hip_frames_against_the_reference_frames(pins, gpu_scenes, oracle_scenes):
    """All 304 + 304 frames through the C ABI: the fp64 path without -gi, and -gi at 1024 samples in both GI precisions."""
    from pathtracing_amd import device
    sc = gpu_scenes("cbox")
    diffs = {device.GI_F64_EXACT: [], device.GI_F32_FAST: []}
    for n, k in enumerate(pins["k"]):
        k = int(k)
        p, j = divmod(k, 16)
        eye = list(pins["pos"][p])
        g = sc.render(eye, absolute_angle(j), 1.0, model=device.MODEL_MAIN, gi=False, size=32, aov=False)["bgr"]
        assert rmse(g, pins["dif"][n]) <= 1.5, (k, rmse(g, pins["dif"][n]))
        for prec in (device.GI_F64_EXACT, device.GI_F32_FAST):
            if prec == device.GI_F64_EXACT and k % 4:
                continue                                       # the fp64 GI path on every fourth frame
            g = sc.render(eye, shipped_angle(pins["pos"][p], j), 1.0, model=device.MODEL_OMP, gi=True, spp=GI_SPP, size=32,
                          gi_precision=prec, view=k, aov=False)["bgr"]
            r = pins["gi"][n]
            assert rmse(g, r) <= 4.5, (k, prec, rmse(g, r))
            d = float(g.mean()) - float(r.mean())
            assert abs(d) <= 0.6, (k, prec, d)
            diffs[prec].append(d)
    assert len(diffs[device.GI_F32_FAST]) == 304 and abs(np.mean(diffs[device.GI_F32_FAST])) <= 0.1
    assert abs(np.mean(diffs[device.GI_F64_EXACT])) <= 0.12
    # and the direct frames are the oracle's, bit for bit, at this size too
    k = 100
    p, j = divmod(k, 16)
    o = oracle_scenes("cbox").render(eye=list(pins["pos"][p]), angle=absolute_angle(j), zoom=1.0, model=po.MODEL_MAIN, gi=False,
                                     size=32, threads=4, aov=False)["bgr"]
    g = sc.render(list(pins["pos"][p]), absolute_angle(j), 1.0, model=device.MODEL_MAIN, gi=False, size=32, aov=False)["bgr"]
    assert np.array_equal(o, g)


@pytest.mark.gpu
def test_hip_four_sample_frames_and_whole_frames(pins, gpu_scenes, oracle_scenes):
    """The 4-sample set through the C ABI in both GI precisions, and the four whole frames: within the bar of the
    reference's frames, and the oracle's frames bit for bit."""
    from pathtracing_amd import device
    sc = gpu_scenes("cbox")
    for prec in (device.GI_F32_FAST, device.GI_F64_EXACT):
        render = lambda k, p, j: sc.render(list(pins["pos"][p]), shipped_angle(pins["pos"][p], j), 1.0, model=device.MODEL_OMP, gi=True,
                                           spp=4, size=32, gi_precision=prec, view=k, aov=False)["bgr"]
        ratios, diffs = four_sample_check(render, pins, range(16, 320))
        assert 0.98 <= np.mean(ratios) <= 1.18, (prec, np.mean(ratios))
        assert abs(np.mean(diffs)) <= 0.3, (prec, np.mean(diffs))
    orc = oracle_scenes("cbox")
    for n, k in enumerate(pins["full_k"]):
        p, j = divmod(int(k), 16)
        kw = dict(model=device.MODEL_MAIN, gi=False, size=512, aov=False)
        g = sc.render(list(pins["pos"][p]), absolute_angle(j), 1.0, **kw)["bgr"]
        assert rmse(g, pins["full_dif"][n]) <= 1.5, int(k)
        o = orc.render(eye=list(pins["pos"][p]), angle=absolute_angle(j), zoom=1.0, model=po.MODEL_MAIN, gi=False, size=512, threads=8,
                       aov=False)["bgr"]
        assert np.array_equal(g, o), int(k)
