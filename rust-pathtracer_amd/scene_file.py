"""ctypes mirror of include/pt_scene_file.h (libptscene.so): the reference's TOML config / scene front end in C++
(csrc/host/scene_file.cpp).  CPU-only; used by tests and tools, the product entry point is csrc/host/ptcli."""
import ctypes as C
import os

from . import api

HERE = os.path.dirname(os.path.abspath(__file__))
LIBRARY_PATH = os.path.join(HERE, "csrc", "libptscene.so")
DATA_ROOT = HERE  # file names in the TOML files are "data/..."; pt_scene_file_set_root(<package directory>)


class RenderSettings(C.Structure):
    _fields_ = [("filename", C.c_char_p), ("width", C.c_uint32), ("height", C.c_uint32), ("integrator", C.c_int32),
                ("light_samples", C.c_uint32), ("medium_aware", C.c_int32), ("camera_samples", C.c_uint32),
                ("min_bounces", C.c_int32), ("max_bounces", C.c_int32), ("hwss", C.c_int32), ("threads", C.c_int32),
                ("min_samples", C.c_uint32), ("max_samples", C.c_int32), ("camera_id", C.c_char_p),
                ("russian_roulette", C.c_int32), ("only_direct", C.c_int32), ("has_wavelength_bounds", C.c_int32),
                ("wavelength_lo", C.c_float), ("wavelength_hi", C.c_float), ("has_premultiply", C.c_int32), ("premultiply", C.c_float),
                ("colorspace", C.c_int32), ("tonemap", C.c_int32), ("has_exposure", C.c_int32), ("exposure", C.c_float),
                ("key_value", C.c_float), ("white_point", C.c_float), ("luminance_only", C.c_int32), ("silenced", C.c_int32)]


class SceneFileError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("pt_status %d: %s" % (status, message))
        self.status = status
        self.message = message


_lib = None


def library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIBRARY_PATH):
            raise RuntimeError("libptscene.so not built: run python -c 'import __graft_entry__ as g; g.build()'")
        L = C.CDLL(LIBRARY_PATH)
        vp = C.c_void_p
        L.pt_scene_file_last_error.restype = C.c_char_p
        L.pt_scene_file_set_root.argtypes = [C.c_char_p]
        L.pt_config_load.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.pt_config_free.argtypes = [vp]
        L.pt_config_scene_file.restype = C.c_char_p; L.pt_config_scene_file.argtypes = [vp]
        L.pt_config_renderer.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.pt_config_render_settings_count.restype = C.c_uint32; L.pt_config_render_settings_count.argtypes = [vp]
        L.pt_config_render_settings.argtypes = [vp, C.c_uint32, C.POINTER(RenderSettings)]
        L.pt_config_render_desc.argtypes = [vp, C.c_uint32, C.c_uint64, C.POINTER(api.RenderDesc)]
        L.pt_config_output_desc.argtypes = [vp, C.c_uint32, C.c_float, C.POINTER(api.OutputDesc)]
        L.pt_scene_file_load.argtypes = [C.c_char_p, vp, C.POINTER(vp)]
        L.pt_scene_file_free.argtypes = [vp]
        L.pt_scene_file_desc.restype = C.POINTER(api.SceneDesc); L.pt_scene_file_desc.argtypes = [vp]
        for n in ("material",):
            getattr(L, "pt_scene_file_" + n).restype = C.c_int64; getattr(L, "pt_scene_file_" + n).argtypes = [vp, C.c_char_p]
        for n in ("curve", "texture", "camera"):
            getattr(L, "pt_scene_file_" + n).restype = C.c_int32; getattr(L, "pt_scene_file_" + n).argtypes = [vp, C.c_char_p]
        L.pt_scene_file_warning_count.restype = C.c_uint32; L.pt_scene_file_warning_count.argtypes = [vp]
        L.pt_scene_file_warning.restype = C.c_char_p; L.pt_scene_file_warning.argtypes = [vp, C.c_uint32]
        L.pt_image_read.argtypes = [C.c_char_p, C.c_int32, C.c_float, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_float))]
        L.pt_image_free.argtypes = [C.POINTER(C.c_float)]
        L.pt_scene_file_set_root(DATA_ROOT.encode())
        _lib = L
    return _lib


def set_root(path=None):
    """The directory the "data/..." file names of config and scene files are relative to (default: this package)."""
    library().pt_scene_file_set_root((path or DATA_ROOT).encode())


def _check(status):
    if status != api.PT_OK:
        raise SceneFileError(status, library().pt_scene_file_last_error().decode())


IMAGE_GREY8, IMAGE_RGBA8, IMAGE_HDR, IMAGE_EXR = range(4)


def read_image(path, kind, alpha_fill=0.0):
    """The texture parser's image readers (src/parsing/texture.rs:48-153): float32 [H, W] or [H, W, 4]."""
    import numpy as np
    w, h, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
    data = C.POINTER(C.c_float)()
    _check(library().pt_image_read(os.fsencode(path), kind, alpha_fill, C.byref(w), C.byref(h), C.byref(c), C.byref(data)))
    try:
        a = np.ctypeslib.as_array(data, shape=(h.value * w.value * c.value,)).copy()
    finally:
        library().pt_image_free(data)
    return a.reshape((h.value, w.value) if c.value == 1 else (h.value, w.value, c.value))


class Config:
    """get_config (src/parsing/mod.rs:565-582)."""

    def __init__(self, path):
        self.handle = C.c_void_p()
        _check(library().pt_config_load(os.fsencode(path), C.byref(self.handle)))

    def __del__(self):
        if getattr(self, "handle", None):
            library().pt_config_free(self.handle)
            self.handle = None

    @property
    def scene_file(self):
        return library().pt_config_scene_file(self.handle).decode()

    @property
    def renderer(self):
        w, h = C.c_uint32(), C.c_uint32()
        kind = library().pt_config_renderer(self.handle, C.byref(w), C.byref(h))
        return kind, (w.value, h.value)

    def __len__(self):
        return library().pt_config_render_settings_count(self.handle)

    def render_settings(self, i):
        s = RenderSettings()
        _check(library().pt_config_render_settings(self.handle, i, C.byref(s)))
        return s

    def render_desc(self, i, seed=1):
        d = api.RenderDesc()
        _check(library().pt_config_render_desc(self.handle, i, seed, C.byref(d)))
        return d

    def output_desc(self, i, factor=1.0):
        d = api.OutputDesc()
        _check(library().pt_config_output_desc(self.handle, i, factor, C.byref(d)))
        return d


class SceneFile:
    """construct_world (src/parsing/mod.rs:145-563): `desc` is a pt_scene_desc owned by this object."""

    def __init__(self, path, config=None):
        self.handle = C.c_void_p()
        self.config = config
        _check(library().pt_scene_file_load(os.fsencode(path), config.handle if config else None, C.byref(self.handle)))

    def __del__(self):
        if getattr(self, "handle", None):
            library().pt_scene_file_free(self.handle)
            self.handle = None

    @property
    def desc(self):
        return library().pt_scene_file_desc(self.handle).contents

    def material(self, name):
        return library().pt_scene_file_material(self.handle, name.encode())

    def curve(self, name):
        return library().pt_scene_file_curve(self.handle, name.encode())

    def texture(self, name):
        return library().pt_scene_file_texture(self.handle, name.encode())

    def camera(self, name):
        return library().pt_scene_file_camera(self.handle, name.encode())

    @property
    def warnings(self):
        n = library().pt_scene_file_warning_count(self.handle)
        return [library().pt_scene_file_warning(self.handle, i).decode() for i in range(n)]

    # what api.Scene expects of a "builder"
    def desc_and_keepalive(self):
        return self.desc, self
