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
    libdir, san = pkg, []
    if os.environ.get("PT_SANITIZE") == "1":
        # host code only (each sanitizer option directly after -Xarch_host).  The compiler is clang, whose sanitizer runtime
        # does not mix with gcc's: the checker is instrumented together with the tree build it compiles in, linked with the
        # plain host library, and run without gcc's runtime preloaded (run_checker).
        san = ["-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined", "-Xarch_host", "-fno-sanitize-recover=all", "-g"]
    subprocess.run([HIPCC, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-x", "hip",
                    os.path.join(ROOT, "tests", "refine_check.cpp"), os.path.join(pkg, "csrc", "device", "pt_refine.hip"),
                    "-o", str(exe), "-L" + libdir, "-lpthost", "-Wl,-rpath," + libdir, "-pthread"] + san, check=True, capture_output=True)
    return str(exe)


@pytest.mark.parametrize("name", ["cbox", "car", "pine", "terrace", "atrium"])
def test_quads_match_the_binary_tree_and_clipping_loses_nothing(checker, tmp_path, name):
    path = scene_path(name) if name != "atrium" else str(tmp_path / "atrium.ptscene")
    if name == "atrium":                     # the procedural hall at 20 000 triangles: slanted debris, thin cells
        from pathtracing_amd import host
        host.HostScene.atrium(20000, 12345).save_packed(path)
    env = dict(os.environ)
    if env.get("PT_SANITIZE") == "1":
        env["LD_PRELOAD"] = ":".join(x for x in env.get("LD_PRELOAD", "").split(":") if x and "libasan" not in x)
    r = subprocess.run([checker, path], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 2 and lines[0].startswith("clip 0:") and lines[1].startswith("clip 1:"), r.stdout
