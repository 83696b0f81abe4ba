"""Host library: reader / packed scenes / kd build / view list / C-ABI surface."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, scene_path
from pathtracing_amd import host

REF_SCENES = "/root/reference/scenes"


def preorder(nodes, refs):
    """Walk the flattened tree in pre-order -> the oracle's dump layout."""
    ni, nd, ent = [], [], []
    stack = [0]
    while stack:
        k = stack.pop()
        n = nodes[k]
        leaf = n["axis"] < 0
        ni.append((1 if leaf else 0, -1 if leaf else n["axis"], n["count"], len(ent)))
        nd.append(list(n["lo"]) + list(n["hi"]) + [0.0 if leaf else n["split"]])
        ent.extend(refs[n["first"]:n["first"] + n["count"]].tolist())
        if not leaf:
            stack.append(n["left"] + 1)
            stack.append(n["left"])
    return np.array(ni, np.int32), np.array(nd), np.array(ent, np.int64)


@pytest.mark.parametrize("name", ["cbox", "car", "pine", "terrace"])
def test_host_tree_equals_oracle_tree(host_scenes, oracle_scenes, name):
    hs, os_ = host_scenes(name), oracle_scenes(name)
    st, ot = hs.stats, os_.tree_stats()
    for k in ("triangles", "naive", "optimized", "nodes", "leaves", "max_depth", "max_leaf", "root_entries"):
        assert st[k] == ot[k], k
    nodes, refs, tri_mesh, tri_local = hs.tree_arrays()
    ni, nd, ent = preorder(nodes, refs)
    oni, ond, oent = os_.tree_dump()
    assert np.array_equal(ni, oni)
    assert np.array_equal(nd, ond)            # boxes and split planes bit for bit
    assert np.array_equal(tri_mesh[ent], oent[:, 0]) and np.array_equal(tri_local[ent], oent[:, 1])


@pytest.mark.parametrize("name", ["cbox", "car", "pine", "terrace"])
def test_children_adjacent_and_slices_cover_refs(host_scenes, name):
    nodes, refs, _, _ = host_scenes(name).tree_arrays()
    inner = nodes["axis"] >= 0
    assert (nodes["left"][inner] > 0).all() and (nodes["left"][inner] + 1 < len(nodes)).all()
    assert nodes["count"].sum() == len(refs)
    order = np.argsort(nodes["first"], kind="stable")
    ends = nodes["first"][order] + nodes["count"][order]
    assert (ends[:-1] <= nodes["first"][order][1:]).all()


def test_packed_round_trip(tmp_path, host_scenes):
    src = scene_path("cbox")
    sc = host.HostScene.from_packed(src)
    out = tmp_path / "again.ptscene"
    sc.save_packed(str(out))
    assert open(src, "rb").read() == open(out, "rb").read()


def test_bad_inputs_fail_loudly(tmp_path):
    with pytest.raises(host.HostError):
        host.HostScene.from_packed(str(tmp_path / "missing.ptscene"))
    bad = tmp_path / "bad.ptscene"
    bad.write_bytes(b"PTSCENE1" + b"\0" * 10)
    with pytest.raises(host.HostError):
        host.HostScene.from_packed(str(bad))
    with pytest.raises(host.HostError):
        host.HostScene.from_dir(str(tmp_path))
    sc = host.HostScene.from_packed(scene_path("cbox"))
    with pytest.raises(host.HostError):
        sc.desc()                     # tree not built yet


@pytest.mark.skipif(not os.path.isdir(REF_SCENES), reason="reference scene directory not present")
@pytest.mark.parametrize("name", ["cbox", "car", "pine"])
def test_text_reader_reproduces_packed_scene(tmp_path, name):
    """The committed packed scenes are what the text reader makes of the reference files."""
    from PIL import Image
    sdir = os.path.join(REF_SCENES, name)
    for fn in os.listdir(sdir):
        if fn.lower().endswith(".jpg"):
            Image.open(os.path.join(sdir, fn)).convert("RGB").save(str(tmp_path / (fn[:-4] + ".ppm")))
    sc = host.HostScene.from_dir(sdir, str(tmp_path))
    out = tmp_path / "x.ptscene"
    sc.save_packed(str(out))
    assert open(scene_path(name), "rb").read() == open(out, "rb").read()


@pytest.mark.skipif(not os.path.isdir(REF_SCENES), reason="reference scene directory not present")
@pytest.mark.parametrize("name", ["cbox", "car", "pine"])
def test_text_reader_with_own_jpeg_decoder_reproduces_packed_scene(tmp_path, name):
    """Same, with the reader decoding the .jpg textures itself (no decoded .ppm anywhere): the built-in decoder
    yields the texels libjpeg-turbo yields, so the packed file is byte-identical again."""
    sc = host.HostScene.from_dir(os.path.join(REF_SCENES, name))
    out = tmp_path / "x.ptscene"
    sc.save_packed(str(out))
    assert open(scene_path(name), "rb").read() == open(out, "rb").read()


def test_image_decoders_against_fixtures(tmp_path):
    """PNGs of every supported colour type and deflate block kind decode to exactly Pillow's pixels (alpha dropped).
    Baseline JPEGs of every supported kind (4:2:0 with odd sizes, 4:2:2, 4:4:4, grey, restart markers, coarse
    quantisation, 1 x 1) decode to exactly the pixels libjpeg-turbo produced (tests/golden/make_jpeg_fixtures.py);
    progressive files and non-JPEG data are refused, not mis-decoded."""
    d = os.path.join(ROOT, "tests", "golden", "jpeg")
    expected = np.load(os.path.join(d, "expected.npz"))
    assert len(expected.files) >= 8
    assert sum(n.startswith("png_") for n in expected.files) >= 8
    for name in expected.files:
        got = host.decode_image(os.path.join(d, name + (".png" if name.startswith("png_") else ".jpg")))
        assert got.shape == expected[name].shape, name
        assert np.array_equal(got, expected[name]), (name, int(np.abs(got.astype(int) - expected[name]).max()))
    bad = tmp_path / "bad.jpg"
    bad.write_bytes(b"\x89PNG\r\n\x1a\n" + bytes(64))
    with pytest.raises(host.HostError):
        host.decode_image(str(bad))
    data = open(os.path.join(d, "c420_mcu.jpg"), "rb").read()
    trunc = tmp_path / "trunc.jpg"
    trunc.write_bytes(data[: len(data) // 3])                     # header intact, scan cut short: still no crash
    try:
        host.decode_image(str(trunc))
    except host.HostError:
        pass
    prog = tmp_path / "prog.jpg"
    prog.write_bytes(data.replace(b"\xff\xc0", b"\xff\xc2", 1))  # frame marker of a progressive file
    with pytest.raises(host.HostError):
        host.decode_image(str(prog))
    with pytest.raises(host.HostError):
        host.decode_image(str(tmp_path / "missing.jpg"))
    png = open(os.path.join(d, "png_rgba.png"), "rb").read()
    for cut in (20, len(png) // 2, len(png) - 20):                  # header only, half the pixel data, no end chunk
        t = tmp_path / f"cut{cut}.png"
        t.write_bytes(png[:cut])
        with pytest.raises(host.HostError):
            host.decode_image(str(t))
    t = tmp_path / "sixteen.png"
    t.write_bytes(png[:24] + bytes([16]) + png[25:])                # bit depth 16: refused
    with pytest.raises(host.HostError):
        host.decode_image(str(t))


def test_reader_text_format_quirks(tmp_path):
    """Record type is the LAST character of the first token; uv clamps to [0, 0.99];
    polygons are fanned; `g name 1` marks a mirror group; '%' stops the file."""
    d = tmp_path / "s"
    d.mkdir()
    (d / "master.txt").write_text("# c\nC 1 2 3 4 90 180 0.5\nO m.txt\nL p 0 10 0 1 1 1 3000\n% stop\nO never.txt\n")
    (d / "m.txt").write_text(
        "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvt 1.5 -2\nvt 0.25 0.5\n"
        "g plain\nf 1/1/1 2/2/1 3/1/1 4/2/1\ng shiny 1\nf 1/1/1 2/2/1 3/2/1\n%\nf 1/1/1 2/2/1 3/2/1\n")
    sc = host.HostScene.from_dir(str(d))
    cam = sc.camera()
    assert cam["eye"] == [1, 2, 3] and cam["zoom"] == 0.5
    assert abs(cam["angle"] - 0.01745329251 * 180) < 1e-15 and abs(cam["astep"] - 0.01745329251 * 90) < 1e-15
    st = sc.build_tree()
    assert st["triangles"] == 3
    d_ = sc.desc()
    vi = np.ctypeslib.as_array(d_.tri_vi, shape=(9,))
    assert vi.tolist() == [0, 1, 2, 0, 2, 3, 0, 1, 2]
    assert np.ctypeslib.as_array(d_.tri_reft, shape=(3,)).tolist() == [0, 0, 1]
    assert np.ctypeslib.as_array(d_.tri_tm, shape=(3,)).tolist() == [-99999] * 3
    uv = np.ctypeslib.as_array(d_.uv, shape=(4,))
    assert uv.tolist() == [0.99, 0.0, 0.25, 0.5]
    assert d_.n_light == 1 and d_.lights[0].type == ord("p") and d_.lights[0].mag == 3000.0


def test_views_from_path(host_scenes):
    cam = host_scenes("car").camera()
    views = host.views_from_path(os.path.join(ROOT, "scenes", "paths", "carpath.txt"), cam["zoom"])
    assert len(views) == 20 * 16
    a, b, c = -350.0, 75.0, 350.0
    base = -np.arccos(-(a / np.sqrt(a * a + c * c)))
    for j in (0, 1, 15):
        v = views[j]
        assert list(v.eye) == [a, b, c] and v.zoom == cam["zoom"]
        assert abs(v.angle - (base + np.pi / 180.0 * 22.5 * j)) < 1e-15


def test_camera_step():
    eye = (C.c_double * 3)(1.0, 2.0, 3.0)
    ang = C.c_double(0.3)
    L = host.lib()
    assert L.pth_camera_step(eye, C.byref(ang), 10.0, 0.1, ord("w")) == 0
    assert abs(eye[0] - (1 + np.cos(0.3) * 10)) < 1e-15 and eye[1] == 2.0 and abs(eye[2] - (3 + np.sin(0.3) * 10)) < 1e-15
    assert L.pth_camera_step(eye, C.byref(ang), 10.0, 0.1, ord("r")) == 0 and abs(ang.value - 0.4) < 1e-15
    assert L.pth_camera_step(eye, C.byref(ang), 10.0, 0.1, ord("?")) != 0


def test_image_writers(tmp_path):
    img = (np.arange(4 * 6 * 3) % 251).astype(np.uint8).reshape(4, 6, 3)
    host.write_image(str(tmp_path / "a.ppm"), img)
    raw = (tmp_path / "a.ppm").read_bytes()
    assert raw.startswith(b"P6\n6 4\n255\n")
    rgb = np.frombuffer(raw[len(b"P6\n6 4\n255\n"):], np.uint8).reshape(4, 6, 3)
    assert np.array_equal(rgb[..., ::-1], img)
    host.write_image(str(tmp_path / "a.bmp"), img)
    assert (tmp_path / "a.bmp").read_bytes()[:2] == b"BM"


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pth?_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    """Both shared libraries load on a machine without a GPU and export what the headers declare."""
    from pathtracing_amd import device
    hl, dl = host.lib(), device.lib()
    names = _declared("pathtracing_host.h")
    assert "pth_scene_load_dir" in names and len(names) >= 12
    for n in names:
        assert hasattr(hl, n), n
    names = _declared("pathtracing_amd.h")
    assert set(names) == set(device.EXPORTS)
    for n in names:
        assert hasattr(dl, n), n
    assert b"gfx950" in dl.pt_version()


def test_ctypes_mirror_matches_the_c_header(tmp_path):
    """The ctypes structures of host.py / device.py have the sizes the C compiler gives the structs of the header
    (the header also compiles as plain C)."""
    import ctypes
    import subprocess
    from pathtracing_amd import device
    names = ["PtKdNode", "PtTexture", "PtLight", "PtSceneDesc", "PtCamera", "PtOptions", "PtAov", "PtStats", "PtSceneInfo"]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "pathtracing_amd.h"\n#include "pathtracing_host.h"\nint main(void) {\n' +
                   "".join(f'  printf("{n} %zu\\n", sizeof({n}));\n' for n in names) + "  return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for n in names:
        cls = getattr(device, n, None) or getattr(host, n)
        assert ctypes.sizeof(cls) == int(out[n]), n


def test_damaged_packed_scene_is_refused(tmp_path):
    """A .ptscene whose triangle records point outside their pools (or whose counts the file cannot hold) is refused by
    the reader itself: the kd build and the flattening index the pools unchecked."""
    import struct
    raw = bytearray(open(scene_path("cbox"), "rb").read())
    n_light = struct.unpack_from("<I", raw, 8 + 56 + 24)[0]
    at = 8 + 56 + 24 + 4 + n_light * 60
    assert struct.unpack_from("<I", raw, at)[0] == 1                          # one mesh
    at += 4 + 24
    nv, nn, nt, ntri, ntex = struct.unpack_from("<5I", raw, at)
    assert ntri == 2190 and ntex == 7
    tris = at + 20 + nv * 24 + nn * 24 + nt * 16                              # 44-byte records: vi[3] ni[3] ti[3] tm reft

    def load(edit):
        bad = bytearray(raw)
        edit(bad)
        p = tmp_path / "damaged.ptscene"
        p.write_bytes(bad)
        return host.HostScene.from_packed(str(p))

    load(lambda b: None)                                                      # the undamaged copy loads
    for what, edit in (
            ("face index", lambda b: struct.pack_into("<i", b, tris + 44 * 5, 10 ** 9)),                 # vi[0] of triangle 5
            ("face index", lambda b: struct.pack_into("<i", b, tris + 44 * 100 + 12, -3)),               # ni[0] negative
            ("face index", lambda b: struct.pack_into("<i", b, tris + 44 * 7 + 24, nt)),                 # ti[0] == pool size
            ("texture index", lambda b: struct.pack_into("<i", b, tris + 44 * 9 + 36, ntex)),            # tm one past the end
            ("texture index", lambda b: struct.pack_into("<i", b, tris + 44 * 9 + 36, -(ntex + 1) - 1)), # per-hit form of the same
            ("corrupt", lambda b: struct.pack_into("<I", b, at + 12, 0x7FFFFFFF))):                      # triangle count
        with pytest.raises(host.HostError, match=what):
            load(edit)


def test_jpeg_writer_round_trip(tmp_path):
    """Frames can be written the way the reference's data sets are (cv::imwrite of a .jpg): the file this build writes is
    read back by Pillow and by this build's own decoder to within a few grey levels of the frame (quality 95, 4:2:0),
    for sizes that are not multiples of the 16-pixel MCU, and lower quality gives a smaller file."""
    from PIL import Image
    rng = np.random.default_rng(3)
    for rows, cols in ((1, 1), (37, 53), (64, 64)):
        yy, xx = np.mgrid[0:rows, 0:cols]
        img = np.stack([127 + 120 * np.sin(xx / 9.0), 127 + 120 * np.cos(yy / 7.0), (xx * 3 + yy * 2) % 256], axis=2)
        img = np.clip(img + rng.normal(0, 2, img.shape), 0, 255).astype(np.uint8)             # BGR
        p = str(tmp_path / f"f{rows}x{cols}.jpg")
        host.write_image(p, img)
        pil = np.asarray(Image.open(p).convert("RGB"))[:, :, ::-1]
        own = host.decode_image(p)
        assert pil.shape == own.shape == img.shape
        if rows > 1:
            # luma is kept closely; the blue ramp wraps every 85 pixels, which 4:2:0 smears, hence the looser colour bound
            err = pil.astype(float) - img
            luma = err @ np.array([0.114, 0.587, 0.299])
            assert np.sqrt(np.mean(luma ** 2)) < 3.0 and np.sqrt(np.mean(err ** 2)) < 12.0
        assert np.abs(pil.astype(int) - own.astype(int)).max() <= 2          # the two decoders agree on the file
    big = str(tmp_path / "q95.jpg"); small = str(tmp_path / "q40.jpg")
    host.write_image(big, img, quality=95); host.write_image(small, img, quality=40)
    assert os.path.getsize(small) < os.path.getsize(big)
    with pytest.raises(host.HostError):
        host.write_image(str(tmp_path / "no_such_dir" / "x.jpg"), img)
