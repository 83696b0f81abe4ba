"""ctypes bindings of the MI355X render library (include/pathtracing_amd.h).

There is no CPU fallback: every call goes to libptamd.so, and loading fails
loudly when the library has not been built.
"""
import ctypes as C
import os

import numpy as np

from .host import PtCamera, PtSceneDesc

_DIR = os.path.dirname(os.path.abspath(__file__))

MODEL_MAIN, MODEL_OMP = 0, 1
GI_F64_EXACT, GI_F32_FAST = 0, 1
FLT_MAX = float(np.finfo(np.float32).max)


class PtOptions(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("gi", C.c_int32), ("rb", C.c_int32), ("model", C.c_int32),
                ("spp", C.c_int32), ("gi_precision", C.c_int32), ("seed", C.c_uint32), ("view", C.c_uint32),
                ("size", C.c_int32), ("flags", C.c_uint32), ("part", C.c_uint32), ("n_parts", C.c_uint32)]


class PtAov(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("mesh", "tri", "t", "u", "v", "rgb", "gclr", "gnrm", "ghit")]


class PtStats(C.Structure):
    _fields_ = [("primary", C.c_uint64), ("shadow", C.c_uint64), ("mirror", C.c_uint64), ("gi", C.c_uint64),
                ("omp_retrace", C.c_uint64), ("ms_total", C.c_double), ("ms_gi_trace", C.c_double),
                ("gi_trace_launches", C.c_uint64), ("gi_nodes_visited", C.c_uint64), ("gi_tris_tested", C.c_uint64),
                ("ms_primary", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class PtSceneInfo(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("host_nodes", "host_refs", "dev_nodes", "dev_refs", "dev_leaves", "dev_max_leaf",
                                          "dev_interior_refs", "dev_bytes")] + [("host_depth", C.c_int32), ("dev_depth", C.c_int32)]


EXPORTS = ("pt_scene_info", "pt_init", "pt_shutdown", "pt_device_count", "pt_scene_slab_size", "pt_scene_slab_export",
           "pt_scene_upload_slab", "pt_last_error", "pt_version", "pt_scene_upload", "pt_scene_free",
           "pt_scene_blob_size", "pt_scene_blob_write", "pt_scene_upload_blob", "pt_render_view",
           "pt_render_view_device", "pt_render_views", "pt_trace", "pt_trace_device", "pt_camera_rays", "pt_hemisphere_samples")

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_DIR, "libptamd.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make device` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.pt_last_error.restype = C.c_char_p
        L.pt_version.restype = C.c_char_p
        L.pt_init.argtypes = [C.c_int]
        L.pt_scene_upload.argtypes = [C.POINTER(PtSceneDesc), C.POINTER(C.c_void_p)]
        L.pt_scene_free.argtypes = [C.c_void_p]
        L.pt_scene_info.argtypes = [C.c_void_p, C.POINTER(PtSceneInfo)]
        L.pt_scene_blob_size.argtypes = [C.POINTER(PtSceneDesc), C.POINTER(C.c_size_t)]
        L.pt_scene_blob_write.argtypes = [C.POINTER(PtSceneDesc), C.c_void_p, C.c_size_t]
        L.pt_scene_upload_blob.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.pt_render_view.argtypes = [C.c_void_p, C.POINTER(PtCamera), C.POINTER(PtOptions), C.c_void_p,
                                     C.POINTER(PtAov), C.POINTER(PtStats)]
        L.pt_render_view_device.argtypes = [C.c_void_p, C.POINTER(PtCamera), C.POINTER(PtOptions), C.c_void_p,
                                            C.POINTER(PtStats)]
        L.pt_render_views.argtypes = [C.c_void_p, C.POINTER(PtCamera), C.c_size_t, C.c_uint32, C.c_uint32,
                                      C.POINTER(PtOptions), C.c_void_p, C.POINTER(PtStats)]
        L.pt_trace.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 6
        L.pt_scene_slab_size.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        L.pt_scene_slab_export.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.pt_scene_upload_slab.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
        L.pt_trace_device.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 9
        L.pt_camera_rays.argtypes = [C.POINTER(PtCamera), C.c_int, C.c_double, C.c_double, C.c_void_p]
        L.pt_hemisphere_samples.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        _lib = L
    return _lib


class DeviceError(RuntimeError):
    pass


def _check(rc, what):
    if rc != 0:
        raise DeviceError(f"{what} failed ({rc}): {lib().pt_last_error().decode()}")


def init(device=0):
    """device >= 0: that GPU; -1: every visible GPU (pt_render_views then shards its views over them)."""
    _check(lib().pt_init(device), "pt_init")


def device_count():
    return lib().pt_device_count()


def make_camera(eye, angle, zoom):
    return PtCamera((C.c_double * 3)(*eye), angle, zoom)


FLAG_COUNT_WORK = 1


def make_options(gi=False, rb=False, model=MODEL_MAIN, spp=0, gi_precision=GI_F64_EXACT, seed=0x5EED, view=0, size=512,
                 flags=0, part=0, n_parts=0):
    return PtOptions(C.sizeof(PtOptions), int(gi), int(rb), model, spp, gi_precision, seed, view, size, flags, part, n_parts)


def scene_blob(desc):
    """Serialise a PtSceneDesc into one byte blob (numpy uint8)."""
    n = C.c_size_t()
    _check(lib().pt_scene_blob_size(C.byref(desc), C.byref(n)), "pt_scene_blob_size")
    blob = np.zeros(n.value, np.uint8)
    _check(lib().pt_scene_blob_write(C.byref(desc), blob.ctypes.data, n.value), "pt_scene_blob_write")
    return blob


class DeviceScene:
    """A scene resident on the GPU (PtScene handle)."""

    def __init__(self, desc=None, blob=None, slab=None):
        """desc: a PtSceneDesc; blob: the scene description as bytes (pt_scene_blob_write); slab: (address, bytes,
        is_device) of a built scene's device data (slab_export of another DeviceScene, possibly in another process)."""
        h = C.c_void_p()
        if slab is not None:
            ptr, nbytes, is_device = slab
            _check(lib().pt_scene_upload_slab(C.c_void_p(ptr), nbytes, int(is_device), C.byref(h)), "pt_scene_upload_slab")
        elif blob is not None:
            blob = np.ascontiguousarray(blob, np.uint8)
            _check(lib().pt_scene_upload_blob(blob.ctypes.data, blob.size, C.byref(h)), "pt_scene_upload_blob")
        else:
            _check(lib().pt_scene_upload(C.byref(desc), C.byref(h)), "pt_scene_upload")
        self._h = h

    def slab_size(self):
        n = C.c_size_t()
        _check(lib().pt_scene_slab_size(self._h, C.byref(n)), "pt_scene_slab_size")
        return n.value

    def slab_export(self, ptr, nbytes, is_device):
        """Copy the scene's device data (one position-independent block) to host or device memory at `ptr`."""
        _check(lib().pt_scene_slab_export(self._h, C.c_void_p(ptr), nbytes, int(is_device)), "pt_scene_slab_export")

    def info(self):
        i = PtSceneInfo()
        _check(lib().pt_scene_info(self._h, C.byref(i)), "pt_scene_info")
        return {k: getattr(i, k) for k, _ in PtSceneInfo._fields_}

    def render(self, eye, angle, zoom, aov=True, **opt):
        o = make_options(**opt)
        size = o.size or 512
        n = size * size
        cam = make_camera(eye, angle, zoom)
        res = {"bgr": np.zeros((size, size, 3), np.uint8)}
        a = PtAov()
        if aov:
            res.update(mesh=np.full(n, -2, np.int32), tri=np.full(n, -2, np.int32), t=np.zeros(n), u=np.zeros(n),
                       v=np.zeros(n), rgb=np.zeros((size, size, 3)), gclr=np.zeros((n, 3)), gnrm=np.zeros((n, 3)),
                       ghit=np.zeros((n, 3)))
            for k, _ in PtAov._fields_:
                setattr(a, k, res[k].ctypes.data)
        st = PtStats()
        _check(lib().pt_render_view(self._h, C.byref(cam), C.byref(o), res["bgr"].ctypes.data,
                                    C.byref(a) if aov else None, C.byref(st)), "pt_render_view")
        res["stats"] = st.as_dict()
        return res

    def render_views(self, cams, view0=0, view_stride=1, **opt):
        o = make_options(**opt)
        size = o.size or 512
        n = len(cams)
        arr = (PtCamera * n)(*cams)
        frames = np.zeros((n, size, size, 3), np.uint8)
        st = PtStats()
        _check(lib().pt_render_views(self._h, arr, n, view0, view_stride, C.byref(o), frames.ctypes.data,
                                     C.byref(st)), "pt_render_views")
        return frames, st.as_dict()

    def render_to_device(self, cam, d_ptr, stats=None, **opt):
        """Render one view into device memory at d_ptr (size*size*3 bytes)."""
        o = make_options(**opt)
        st = stats if stats is not None else PtStats()
        _check(lib().pt_render_view_device(self._h, C.byref(cam), C.byref(o), C.c_void_p(d_ptr), C.byref(st)),
               "pt_render_view_device")
        return st

    def trace(self, src, dir_, tmin_in, mode=0):
        src = np.ascontiguousarray(src, np.float64)
        dir_ = np.ascontiguousarray(dir_, np.float64)
        tmin_in = np.ascontiguousarray(tmin_in, np.float64)
        n = src.shape[0]
        found, mesh, tri = (np.zeros(n, np.int32) for _ in range(3))
        t, u, v = (np.zeros(n) for _ in range(3))
        _check(lib().pt_trace(self._h, n, src.ctypes.data, dir_.ctypes.data, tmin_in.ctypes.data, mode,
                              found.ctypes.data, mesh.ctypes.data, tri.ctypes.data, t.ctypes.data, u.ctypes.data,
                              v.ctypes.data), "pt_trace")
        return {"found": found, "mesh": mesh, "tri": tri, "t": t, "u": u, "v": v}

    def trace_device(self, n, d_src, d_dir, d_tmin_in, d_found, d_mesh, d_tri, d_t, d_u, d_v):
        """pt_trace_device: the fp32 nearest-hit search on n rays in device memory (arguments are device addresses);
        allocates nothing."""
        _check(lib().pt_trace_device(self._h, n, d_src, d_dir, d_tmin_in, d_found, d_mesh, d_tri, d_t, d_u, d_v),
               "pt_trace_device")

    def close(self):
        if self._h:
            lib().pt_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def camera_rays(eye, angle, zoom, size=512, x=0.0, y=0.0):
    out = np.zeros((size * size, 3))
    cam = make_camera(eye, angle, zoom)
    _check(lib().pt_camera_rays(C.byref(cam), size, x, y, out.ctypes.data), "pt_camera_rays")
    return out


def hemisphere_samples(seed, view, pixel, n):
    out = np.zeros((n, 3))
    _check(lib().pt_hemisphere_samples(seed, view, pixel, n, out.ctypes.data), "pt_hemisphere_samples")
    return out
