"""The fp32 tree build on the host, without a GPU: quad records against binary records, and split clipping against
points sampled on every triangle (tests/refine_check.cpp states the checks).  The build is host code inside a HIP
source file, so the checker is compiled with hipcc (which cross-compiles without a GPU)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT, scene_path

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc is not installed")
    exe = tmp_path_factory.mktemp("refine") / "refine_check"
    pkg = os.path.join(ROOT, "pathtracing_amd")
    subprocess.run([HIPCC, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-x", "hip",
                    os.path.join(ROOT, "tests", "refine_check.cpp"), os.path.join(pkg, "csrc", "device", "pt_refine.hip"),
                    "-o", str(exe), "-L" + pkg, "-lpthost", "-Wl,-rpath," + pkg, "-pthread"], check=True, capture_output=True)
    return str(exe)


@pytest.mark.parametrize("name", ["cbox", "car", "pine", "terrace", "atrium"])
def test_quads_match_the_binary_tree_and_clipping_loses_nothing(checker, tmp_path, name):
    path = scene_path(name) if name != "atrium" else str(tmp_path / "atrium.ptscene")
    if name == "atrium":                     # the procedural hall at 20 000 triangles: slanted debris, thin cells
        from pathtracing_amd import host
        host.HostScene.atrium(20000, 12345).save_packed(path)
    r = subprocess.run([checker, path], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 2 and lines[0].startswith("clip 0:") and lines[1].startswith("clip 1:"), r.stdout
