#!/usr/bin/env python3
"""Pins of the oracle against frames the reference itself rendered (build container only).

The reference ships 512 x 512 JPEG frames of its own renderer under ml/data/<SCENE>/ (training data of its
denoiser): <SCENE>dif (no -gi), <SCENE>4ssp and <SCENE>4096ssp (-gi, noisy and converged), 320 each = 20 path
positions x 16 angles.  This script finds which of them the shipped scene files and camera path reproduce and
stores those, subsampled, as a fixture; tests/test_refimage_pins.py then holds the oracle (CPU) and the HIP
path (GPU) against them.

What was found for cbox (scenes/cbox, scenes/paths/cboxpath.txt; frame k = position k // 16, angle index k % 16):

* CBOXdif/cbox<k>.jpg is the main.cpp shading model (shadow rays, mirror bounce weighted 0.6, main.cpp:41-78)
  without -gi, at the ABSOLUTE view angle 22.5 deg * (k % 16) from the path position -- an earlier form of
  renderParallel's camera than the shipped one (mainOMP.cpp:164-165 adds -acos(-unit(a, 0, c).x)).
* CBOX4096ssp/cbox<k>.jpg is the mainOMP.cpp model with -gi (mainOMP.cpp:37-136: no shadows, mirror bounce
  weighted 1, 2 * mclr * GI / N) at the shipped camera of mainOMP.cpp:164-165.
* CBOX4ssp/cbox<k>.jpg is that same model and camera at FOUR samples per pixel (the data set's noisy input): its
  frames scatter around the 4096-sample frames exactly as much as 4-sample renders of the oracle do (RMSE against
  the converged frame 24-33 / 255; 2 samples would give 31-43, 8 samples 18-27), so the set pins the sample count
  N and the weight 2 * mclr * GI / N of mainOMP.cpp:132-134 at a second N.
* All three hold for the 304 frames of positions 1..19; the 16 frames of position 0 were rendered from some other
  place and are left out.
* Four CBOXdif frames are also kept whole, 512 x 512 (position 5, angle indices 0, 4, 8, 12: one view per wall):
  a render that is off by one pixel anywhere cannot hide in them as it could in every 16th pixel.
* ml/data/CAR/* was rendered from an arrangement of the car scene that differs from the shipped scenes/car
  (the room is smaller and the car fills more of the frame at every position; a grid + simplex search over the
  eye position leaves an RMSE above 20 / 255), so no car frame can serve as a pin.  Blind matching (every 8th
  path view rendered without -gi in both shading models, each compared with all 320 CARdif frames) does find
  pairs at 2.8-5 / 255, but they are the data set's darkest frames (mean brightness 14-20 of 255; frames 135, 165,
  167 and 183 are the "best match" of two dozen different views each) with no relation between the indices, while
  the frame of the same index is 12-70 / 255 away: agreement of mostly black pictures, not of renders.

A size-S render samples the image plane at j / S, so pixel (i, j) of a 32 x 32 render is exactly pixel
(16 i, 16 j) of the 512 x 512 frame: the fixture keeps every 16th pixel of every 16th row.

Usage: python tests/golden/make_refimage_pins.py   (needs /root/reference and Pillow; writes
tests/golden/refimages_cbox.npz, refimages_cbox_full.npz and refimages_cbox.json)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/ml/data/CBOX"
STRIDE, SIDE = 16, 32
GI_SPP = 1024
FULL_K = (80, 84, 88, 92)


def path_positions(path_file):
    lines = open(path_file).read().split("\n")
    n = int(lines[0])
    return [tuple(float(x) for x in lines[1 + i].split()) for i in range(n)]


def shipped_angle(pos, j):
    """mainOMP.cpp:164-165."""
    a, c = pos[0], pos[2]
    return -np.arccos(-(a / np.sqrt(a * a + c * c))) + np.pi / 180.0 * (22.5 * j)


def absolute_angle(j):
    return np.pi / 180.0 * (22.5 * j)


def rmse(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def main():
    from PIL import Image

    from oracle import ptoracle as po
    orc = po.OracleScene(os.path.join(ROOT, "scenes", "cbox.ptscene"))
    zoom = orc.camera()["zoom"]
    pos = path_positions(os.path.join(ROOT, "scenes", "paths", "cboxpath.txt"))

    def ref(sub, k):
        im = np.asarray(Image.open(os.path.join(REF, sub, f"cbox{k}.jpg")).convert("RGB"))
        assert im.shape == (512, 512, 3)
        return im[::STRIDE, ::STRIDE, ::-1].copy()          # BGR, as the frames of the render path

    rows, keep, dif, gi, gi4 = [], [], [], [], []
    for k in range(320):
        p, j = divmod(k, 16)
        d_ref, g_ref, g4_ref = ref("CBOXdif", k), ref("CBOX4096ssp", k), ref("CBOX4ssp", k)
        d = orc.render(eye=list(pos[p]), angle=absolute_angle(j), zoom=zoom, model=po.MODEL_MAIN, gi=False, size=SIDE,
                       threads=8, aov=False)["bgr"]
        g = orc.render(eye=list(pos[p]), angle=shipped_angle(pos[p], j), zoom=zoom, model=po.MODEL_OMP, gi=True, spp=GI_SPP,
                       size=SIDE, threads=8, aov=False, view=k)["bgr"]
        row = {"k": k, "position": p, "angle_index": j, "dif_rmse": round(rmse(d, d_ref), 3),
               "gi_rmse": round(rmse(g, g_ref), 3), "gi_mean_diff": round(float(g.mean()) - float(g_ref.mean()), 3)}
        # the 4-sample set: how far a 4-sample frame lies from the converged one, reference against oracle (and the
        # oracle at 2 and 8 samples, to show that the figure tells sample counts apart)
        row["gi4_ref_noise"] = round(rmse(g4_ref, g_ref), 3)
        for n in (2, 4, 8):
            o = orc.render(eye=list(pos[p]), angle=shipped_angle(pos[p], j), zoom=zoom, model=po.MODEL_OMP, gi=True, spp=n,
                           size=SIDE, threads=8, aov=False, view=k)["bgr"]
            row[f"gi{n}_oracle_noise"] = round(rmse(o, g_ref), 3)
            if n == 4:
                row["gi4_mean_diff"] = round(float(o.mean()) - float(g4_ref.mean()), 3)
        rows.append(row)
        if p >= 1:
            keep.append(k); dif.append(d_ref); gi.append(g_ref); gi4.append(g4_ref)
        print(row, flush=True)
    out = os.path.join(ROOT, "tests", "golden", "refimages_cbox.npz")
    np.savez_compressed(out, k=np.array(keep, np.int32), dif=np.array(dif, np.uint8), gi4096=np.array(gi, np.uint8),
                        gi4=np.array(gi4, np.uint8))
    full, full_rows = [], []
    for k in FULL_K:
        p, j = divmod(k, 16)
        im = np.asarray(Image.open(os.path.join(REF, "CBOXdif", f"cbox{k}.jpg")).convert("RGB"))[:, :, ::-1].copy()
        d = orc.render(eye=list(pos[p]), angle=absolute_angle(j), zoom=zoom, model=po.MODEL_MAIN, gi=False, size=512, threads=8,
                       aov=False)["bgr"]
        shifted = {name: round(rmse(np.roll(d, sh, axis=ax), im), 3) for name, sh, ax in (("down", 1, 0), ("up", -1, 0), ("right", 1, 1), ("left", -1, 1))}
        full.append(im)
        full_rows.append({"k": k, "rmse": round(rmse(d, im), 3), "rmse_oracle_shifted_by_one_pixel": shifted})
        print(full_rows[-1], flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "refimages_cbox_full.npz"), k=np.array(FULL_K, np.int32),
                        dif=np.array(full, np.uint8))
    kept = [r for r in rows if r["position"] >= 1]
    ratio = lambda n: [r[f"gi{n}_oracle_noise"] / r["gi4_ref_noise"] for r in kept]
    summary = {"source": "ml/data/CBOX/{CBOXdif,CBOX4096ssp}/cbox<k>.jpg of the reference, every 16th pixel of every 16th row, BGR",
               "oracle_gi_spp": GI_SPP, "frames_kept": len(kept),
               "dif_rmse_max_kept": max(r["dif_rmse"] for r in kept), "gi_rmse_max_kept": max(r["gi_rmse"] for r in kept),
               "gi_mean_diff_max_abs_kept": max(abs(r["gi_mean_diff"]) for r in kept),
               "dif_rmse_min_position0": min(r["dif_rmse"] for r in rows if r["position"] == 0),
               "gi_rmse_min_position0": min(r["gi_rmse"] for r in rows if r["position"] == 0),
               "gi4": {"what": "RMSE of a 4-sample frame against the reference's 4096-sample frame: oracle at n samples / reference's CBOX4ssp frame",
                       **{f"ratio_n{n}": {"min": round(min(ratio(n)), 3), "mean": round(float(np.mean(ratio(n))), 3), "max": round(max(ratio(n)), 3)}
                          for n in (2, 4, 8)},
                       "mean_diff_max_abs": max(abs(r["gi4_mean_diff"]) for r in kept),
                       "mean_diff_mean": round(float(np.mean([r["gi4_mean_diff"] for r in kept])), 3)},
               "full_frames": full_rows,
               "frames": rows}
    json.dump(summary, open(os.path.join(ROOT, "tests", "golden", "refimages_cbox.json"), "w"), indent=0)
    print({k: v for k, v in summary.items() if k != "frames"})


if __name__ == "__main__":
    main()
