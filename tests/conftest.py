import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def sanitizer_flags():
    """Extra compiler flags for the C / C++ helpers that tests build: with PT_SANITIZE=1 (`make asan-test`) they are
    instrumented like the libraries they are linked with."""
    if os.environ.get("PT_SANITIZE") == "1":
        return ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
    return []


_packed_here = {}


def scene_path(name):
    """Packed scene file of `name`: scenes/<name>.ptscene, or -- for a scene kept as a directory of text files under
    tests/golden/ -- a temporary file that the host reader packs on first use."""
    p = os.path.join(SCENES, name + ".ptscene")
    if os.path.exists(p):
        return p
    if name not in _packed_here:
        import tempfile
        from pathtracing_amd import host
        out = os.path.join(tempfile.mkdtemp(prefix="ptscene_"), name + ".ptscene")
        host.HostScene.from_dir(os.path.join(GOLDEN, name)).save_packed(out)
        _packed_here[name] = out
    return _packed_here[name]


@pytest.fixture(scope="session")
def oracle_scenes():
    """Oracle scenes (loaded + kd-tree built) keyed by name, built once per session."""
    from oracle import ptoracle as po
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = po.OracleScene(scene_path(name))
        return cache[name]
    return get


@pytest.fixture(scope="session")
def host_scenes():
    from pathtracing_amd import host
    cache = {}

    def get(name):
        if name not in cache:
            sc = host.HostScene.from_packed(scene_path(name))
            sc.build_tree()
            cache[name] = sc
        return cache[name]
    return get


@pytest.fixture(scope="session")
def gpu_scenes(host_scenes):
    """Scenes uploaded to the GPU through the C ABI, keyed by name."""
    from pathtracing_amd import device
    device.init(0)
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = device.DeviceScene(host_scenes(name).desc())
        return cache[name]
    yield get
    for s in cache.values():
        s.close()
