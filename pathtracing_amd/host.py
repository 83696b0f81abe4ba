"""ctypes bindings of the host scene library (include/pathtracing_host.h)."""
import ctypes as C
import os

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))


class PtKdNode(C.Structure):
    _fields_ = [("lo", C.c_double * 3), ("hi", C.c_double * 3), ("split", C.c_double),
                ("axis", C.c_int32), ("left", C.c_int32), ("first", C.c_int32), ("count", C.c_int32)]


class PtTexture(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("bgr", C.POINTER(C.c_uint8))]


class PtLight(C.Structure):
    _fields_ = [("type", C.c_int32), ("reserved", C.c_int32), ("mag", C.c_double),
                ("pos", C.c_double * 3), ("color", C.c_double * 3)]


class PtSceneDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32),
                ("n_mesh", C.c_uint32), ("n_vert", C.c_uint32), ("n_norm", C.c_uint32), ("n_uv", C.c_uint32),
                ("n_tri", C.c_uint32), ("n_tex", C.c_uint32), ("n_light", C.c_uint32), ("n_node", C.c_uint32),
                ("n_ref", C.c_uint32),
                ("vert", C.POINTER(C.c_double)), ("norm", C.POINTER(C.c_double)), ("uv", C.POINTER(C.c_double)),
                ("tri_vi", C.POINTER(C.c_int32)), ("tri_ni", C.POINTER(C.c_int32)), ("tri_ti", C.POINTER(C.c_int32)),
                ("tri_tm", C.POINTER(C.c_int32)), ("tri_reft", C.POINTER(C.c_int32)),
                ("tri_mesh", C.POINTER(C.c_int32)), ("tri_local", C.POINTER(C.c_int32)),
                ("mesh_albedo", C.POINTER(C.c_double)),
                ("tex", C.POINTER(PtTexture)), ("lights", C.POINTER(PtLight)),
                ("background", C.c_double * 3),
                ("nodes", C.POINTER(PtKdNode)), ("refs", C.POINTER(C.c_int32))]


class PtCamera(C.Structure):
    _fields_ = [("eye", C.c_double * 3), ("angle", C.c_double), ("zoom", C.c_double)]


class PthTreeStats(C.Structure):
    _fields_ = [("triangles", C.c_int64), ("naive", C.c_int64), ("optimized", C.c_int64),
                ("nodes", C.c_int64), ("leaves", C.c_int64), ("max_depth", C.c_int64),
                ("max_leaf", C.c_int64), ("root_entries", C.c_int64), ("build_seconds", C.c_double)]


_lib = None


def lib():
    """Load libpthost.so (built in-tree by `make host`)."""
    global _lib
    if _lib is None:
        path = os.path.join(_DIR, "libpthost.so")
        if os.environ.get("PT_SANITIZE") == "1":         # the sanitizer build of `make asan` (CPU tests only)
            path = os.path.join(os.path.dirname(_DIR), "build", "asan", "libpthost.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make host` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.pth_last_error.restype = C.c_char_p
        L.pth_scene_load_dir.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        L.pth_scene_load_packed.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.pth_scene_save_packed.argtypes = [C.c_void_p, C.c_char_p]
        L.pth_scene_build_tree.argtypes = [C.c_void_p, C.POINTER(PthTreeStats)]
        L.pth_scene_desc.argtypes = [C.c_void_p, C.POINTER(PtSceneDesc)]
        L.pth_scene_camera.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.pth_scene_free.argtypes = [C.c_void_p]
        L.pth_scene_free.restype = None
        L.pth_views_from_path.argtypes = [C.c_char_p, C.c_double, C.POINTER(PtCamera), C.c_size_t,
                                          C.POINTER(C.c_size_t)]
        L.pth_camera_step.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double, C.c_int]
        L.pth_decode_image.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]
        L.pth_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        L.pth_write_bmp.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        L.pth_write_jpeg.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.pth_scene_make_atrium.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        _lib = L
    return _lib


class HostError(RuntimeError):
    pass


def _check(rc, what):
    if rc != 0:
        raise HostError(f"{what} failed ({rc}): {lib().pth_last_error().decode()}")


class HostScene:
    """A scene held by the host library: reader -> kd build -> PtSceneDesc."""

    def __init__(self, handle):
        self._h = handle
        self.stats = None
        self._desc = None

    @classmethod
    def from_dir(cls, scene_dir, texture_dir=None):
        h = C.c_void_p()
        _check(lib().pth_scene_load_dir(os.fsencode(scene_dir), os.fsencode(texture_dir) if texture_dir else None,
                                        C.byref(h)), "pth_scene_load_dir")
        return cls(h)

    @classmethod
    def from_packed(cls, path):
        h = C.c_void_p()
        _check(lib().pth_scene_load_packed(os.fsencode(path), C.byref(h)), "pth_scene_load_packed")
        return cls(h)

    @classmethod
    def atrium(cls, triangles=262267, seed=12345):
        h = C.c_void_p()
        _check(lib().pth_scene_make_atrium(triangles, seed, C.byref(h)), "pth_scene_make_atrium")
        return cls(h)

    def save_packed(self, path):
        _check(lib().pth_scene_save_packed(self._h, os.fsencode(path)), "pth_scene_save_packed")

    def build_tree(self):
        st = PthTreeStats()
        _check(lib().pth_scene_build_tree(self._h, C.byref(st)), "pth_scene_build_tree")
        self.stats = {k: getattr(st, k) for k, _ in PthTreeStats._fields_}
        self._desc = None
        return self.stats

    def desc(self):
        if self._desc is None:
            d = PtSceneDesc()
            _check(lib().pth_scene_desc(self._h, C.byref(d)), "pth_scene_desc")
            self._desc = d
        return self._desc

    def camera(self):
        """(eye[3], step, astep, angle, zoom) of the scene's C record."""
        out = (C.c_double * 7)()
        _check(lib().pth_scene_camera(self._h, out), "pth_scene_camera")
        v = list(out)
        return {"eye": v[0:3], "step": v[3], "astep": v[4], "angle": v[5], "zoom": v[6]}

    def tree_arrays(self):
        """The flattened tree as numpy arrays (copies), for tests."""
        d = self.desc()
        nodes = np.ctypeslib.as_array(C.cast(d.nodes, C.POINTER(C.c_uint8)), shape=(d.n_node * C.sizeof(PtKdNode),))
        dt = np.dtype([("lo", "<f8", 3), ("hi", "<f8", 3), ("split", "<f8"), ("axis", "<i4"), ("left", "<i4"),
                       ("first", "<i4"), ("count", "<i4")])
        nodes = nodes.view(dt).copy()
        refs = np.ctypeslib.as_array(d.refs, shape=(d.n_ref,)).copy() if d.n_ref else np.zeros(0, np.int32)
        tri_mesh = np.ctypeslib.as_array(d.tri_mesh, shape=(d.n_tri,)).copy()
        tri_local = np.ctypeslib.as_array(d.tri_local, shape=(d.n_tri,)).copy()
        return nodes, refs, tri_mesh, tri_local

    def close(self):
        if self._h:
            lib().pth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def views_from_path(path_file, zoom):
    """Cameras of the multi-view job: positions x 16 yaw angles (mainOMP.cpp:225-235,164-165)."""
    n = C.c_size_t()
    _check(lib().pth_views_from_path(os.fsencode(path_file), zoom, None, 0, C.byref(n)), "pth_views_from_path")
    arr = (PtCamera * n.value)()
    _check(lib().pth_views_from_path(os.fsencode(path_file), zoom, arr, n.value, C.byref(n)), "pth_views_from_path")
    return arr


def camera_step(eye, angle, step, astep, key):
    """One camera step of simulate() (main.cpp:160-187); key in "wasdlr".  Returns (eye, angle)."""
    e = (C.c_double * 3)(*eye)
    a = C.c_double(angle)
    _check(lib().pth_camera_step(e, C.byref(a), step, astep, ord(key)), "pth_camera_step")
    return list(e), a.value


def decode_image(path):
    """Decode a texture file the way the scene reader does (baseline JPEG or 8-bit PNG, by content):
    (rows, cols, 3) uint8, BGR."""
    r, c = C.c_int(), C.c_int()
    _check(lib().pth_decode_image(os.fsencode(path), C.byref(r), C.byref(c), None, 0), "pth_decode_image")
    out = np.zeros((r.value, c.value, 3), np.uint8)
    _check(lib().pth_decode_image(os.fsencode(path), C.byref(r), C.byref(c), out.ctypes.data, out.size), "pth_decode_image")
    return out


decode_jpeg = decode_image


def write_image(path, bgr, quality=95):
    """By the file name's ending: .bmp, .jpg / .jpeg (baseline JPEG, quality as cv::imwrite's default), else PPM."""
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    low = str(path).lower()
    if low.endswith(".jpg") or low.endswith(".jpeg"):
        _check(lib().pth_write_jpeg(os.fsencode(path), bgr.ctypes.data, bgr.shape[0], bgr.shape[1], quality), "pth_write_jpeg")
        return
    fn = lib().pth_write_bmp if low.endswith(".bmp") else lib().pth_write_ppm
    _check(fn(os.fsencode(path), bgr.ctypes.data, bgr.shape[0], bgr.shape[1]), "pth_write_image")
