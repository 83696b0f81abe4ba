"""pt_trace_device, the entry point that allocates nothing: its outputs equal pt_trace mode 1 (the same fp32
nearest-hit search with buffers allocated per call) on every ray, bit for bit -- found, (mesh, tri), t, u, v --
on two calls in a row that share the library's one scratch array, the second with a smaller input."""
import ctypes as C

import numpy as np
import pytest

from pathtracing_amd import device

pytestmark = pytest.mark.gpu


def rays(n, seed, box):
    rng = np.random.default_rng(seed)
    lo, hi = np.array(box[:3]), np.array(box[3:])
    src = lo + (hi - lo) * (0.1 + 0.8 * rng.random((n, 3)))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[: n // 50, 0] = 0.0
    return src, d


def hip_runtime():
    """The HIP runtime the library itself is linked to (already mapped once the library is loaded): device buffers for
    the test come from it, so that the test does not bring a second runtime into the process."""
    device.lib()
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            hip = C.CDLL(line.split()[-1])
            hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
            hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            hip.hipFree.argtypes = [C.c_void_p]
            return hip
    raise RuntimeError("libptamd.so is loaded but no HIP runtime is mapped")


def run_device(sc, src, d, tmin):
    hip = hip_runtime()
    n = len(src)
    H2D, D2H = 1, 2
    bufs = []

    def dev(a):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), a.nbytes) == 0
        assert hip.hipMemcpy(p, a.ctypes.data, a.nbytes, H2D) == 0          # synchronous: done when it returns
        bufs.append(p)
        return p

    ins = [dev(np.ascontiguousarray(a, np.float64)) for a in (src, d, tmin)]
    # outputs start with junk, so that a value the call did not write is seen
    outs_h = [np.full(n, -7, np.int32) for _ in range(3)] + [np.full(n, -7.0) for _ in range(3)]
    outs = [dev(a) for a in outs_h]
    try:
        sc.trace_device(n, *ins, *outs)
        for p, a in zip(outs, outs_h):
            assert hip.hipMemcpy(a.ctypes.data, p, a.nbytes, D2H) == 0
    finally:
        for p in bufs:
            hip.hipFree(p)
    return dict(zip(("found", "mesh", "tri", "t", "u", "v"), outs_h))


@pytest.mark.parametrize("name", ["car", "terrace"])
def test_no_alloc_trace_equals_pt_trace(gpu_scenes, oracle_scenes, name):
    sc = gpu_scenes(name)
    box = oracle_scenes(name).root_box()
    for n, seed in ((100003, 31), (4097, 32)):            # two calls in a row, the second smaller
        src, d = rays(n, seed, box)
        tmin = np.full(n, device.FLT_MAX)
        tmin[::7] = np.random.default_rng(seed).random(len(tmin[::7])) * 300.0      # bounded queries among them
        want = sc.trace(src, d, tmin, mode=1)
        got = run_device(sc, src, d, tmin)
        assert want["found"].sum() > n // 4
        for k in ("found", "mesh", "tri", "t", "u", "v"):
            assert np.array_equal(got[k], want[k]), (name, n, k)


def test_no_alloc_trace_limits(gpu_scenes):
    sc = gpu_scenes("cbox")
    with pytest.raises(device.DeviceError, match="2\\^24"):
        sc.trace_device((1 << 24) + 1, 8, 8, 8, 8, 8, 8, 8, 8, 8)
    with pytest.raises(device.DeviceError, match="null"):
        sc.trace_device(4, 0, 8, 8, 8, 8, 8, 8, 8, 8)
    sc.trace_device(0, 0, 0, 0, 0, 0, 0, 0, 0, 0)            # nothing to do: no pointer is looked at
