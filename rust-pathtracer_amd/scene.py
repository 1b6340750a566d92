"""Host-side scene assembly: builds the flat pt_scene_desc (include/pt_api.h) the way the
reference's `construct_world` (src/parsing/mod.rs:145-563) builds a `World`, plus the
benchmark scenes of BASELINE.json / SURVEY.md §8(d).

This is the caller's side of the boundary (what a Rust `impl Renderer` would do when it
flattens its `World`); nothing here is on the hot path.
"""
import ctypes as C
import json
import math
import os

import numpy as np

from . import api
from .objmesh import load_obj

_DATA = os.path.join(os.path.dirname(__file__), "data")
EXTENDED_VISIBLE_RANGE = (370.0, 790.0)
BOUNDED_VISIBLE_RANGE = (380.0, 750.0)
_IDENTITY = np.eye(4, dtype=np.float64)


def _spectra():
    with open(os.path.join(_DATA, "spectra.json")) as f:
        return json.load(f)


# ---- Transform3 (math crate): from_stack(scale, rotate, translate) = T * R * S -------------------
# Inputs are f32 (Transform3Data holds f32 fields, src/parsing/instance.rs:17-31); the composition runs in f64 with a
# fixed operation order, the same as csrc/host/scene_file.cpp, so both front ends produce the same matrices bit for bit.
def _f32(v):
    return [float(np.float32(x)) for x in v]


def _mat4_mul(a, b):
    r = np.zeros((4, 4))
    for i in range(4):
        for j in range(4):
            s = 0.0
            for k in range(4):
                s += float(a[i][k]) * float(b[k][j])
            r[i, j] = s
    return r


def transform_from_scale(s):
    m = np.eye(4); m[0, 0], m[1, 1], m[2, 2] = _f32(s)
    return m


def transform_from_translation(t):
    m = np.eye(4); m[:3, 3] = _f32(t)
    return m


def transform_from_axis_angle(axis, angle_rad):
    x, y, z = _f32(axis)
    n = math.sqrt(x * x + y * y + z * z)
    x, y, z = x / n, y / n, z / n
    c, s = math.cos(angle_rad), math.sin(angle_rad)
    r = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                  [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                  [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]])
    m = np.eye(4); m[:3, :3] = r
    return m


def transform_from_data(scale=None, rotate=None, translate=None):
    """Transform3Data -> forward matrix (src/parsing/instance.rs:40-71): rotations given as
    [(axis, degrees), ...] are applied in list order."""
    m = np.eye(4)
    if scale is not None:
        m = _mat4_mul(transform_from_scale(scale), m)
    if rotate:
        base = None
        for axis, deg in rotate:
            t = transform_from_axis_angle(axis, math.pi * float(np.float32(deg)) / 180.0)
            base = t if base is None else _mat4_mul(t, base)
        m = _mat4_mul(base, m)
    if translate is not None:
        m = _mat4_mul(transform_from_translation(translate), m)
    return m


def transform_inverse(m):
    """Transform3::reverse: Gauss-Jordan with partial pivoting in f64, operation for operation the same as `inverse` in
    csrc/host/scene_file.cpp so that both front ends hand the engine the same bits."""
    a = [[float(m[i][j]) for j in range(4)] for i in range(4)]
    inv = [[1.0 if i == j else 0.0 for j in range(4)] for i in range(4)]
    for col in range(4):
        piv = col
        for r in range(col + 1, 4):
            if abs(a[r][col]) > abs(a[piv][col]):
                piv = r
        if a[piv][col] == 0.0:
            raise ValueError("singular transform")
        if piv != col:
            a[piv], a[col] = a[col], a[piv]
            inv[piv], inv[col] = inv[col], inv[piv]
        p = a[col][col]
        for j in range(4):
            a[col][j] /= p
            inv[col][j] /= p
        for r in range(4):
            if r == col:
                continue
            f = a[r][col]
            for j in range(4):
                a[r][j] -= f * a[col][j]
                inv[r][j] -= f * inv[col][j]
    return np.array(inv, dtype=np.float64)


class SceneBuilder:
    def __init__(self):
        self.curves = []          # api.Curve
        self.curve_data = []      # floats
        self.curve_names = {}
        self.layers = []
        self.texstacks = []
        self.texture_data = []
        self.texstack_names = {}
        self.materials = []
        self.material_ids = {}    # name -> packed MaterialId
        self.mediums = []         # api.Medium; MediumId = index + 1 (0 = vacuum)
        self.medium_ids = {}
        self.meshes = []
        self.vertices = []
        self.indices = []
        self.normals = []
        self.face_materials = []
        self.instances = []
        self.cameras = []
        self.environment = api.Environment()
        self.environment.kind = api.ENV_CONSTANT
        self.environment.curve = -1
        self.environment.texstack = -1
        self.env_sampling_probability = 0.5  # scene default (src/parsing/mod.rs:559)
        # material 0 = mauve error light (src/parsing/mod.rs:438-455, src/curves.rs:41-48, 71-77)
        mauve = self.curve_exponential("__mauve", [(650.0, 300.0, 300.0, 1.0), (460.0, 200.0, 400.0, 0.75)])
        void = self.curve_flat("__cie_e_0", 0.0)
        self.material_diffuse_light("error", mauve, void, api.SIDED_DUAL)

    # ---- curves (src/parsing/curves.rs:298-372)
    def _add_curve(self, name, kind, mode=0, p0=0.0, p1=0.0, data=()):
        c = api.Curve(kind, mode, p0, p1, len(self.curve_data), 0)
        data = [float(x) for x in data]
        if kind == api.CURVE_TABULATED:
            c.data_count = len(data) // 2
        elif kind in (api.CURVE_EXPONENTIAL, api.CURVE_INV_EXPONENTIAL):
            c.data_count = len(data) // 4
        else:
            c.data_count = len(data)
        self.curve_data.extend(data)
        self.curves.append(c)
        idx = len(self.curves) - 1
        if name is not None:
            self.curve_names[name] = idx
        return idx

    def curve_flat(self, name, strength):  # CurveData::Flat -> Curve::Linear over EXTENDED_VISIBLE_RANGE
        return self._add_curve(name, api.CURVE_LINEAR, api.INTERP_LINEAR, *EXTENDED_VISIBLE_RANGE, data=[strength])

    def curve_cauchy(self, name, a, b):
        return self._add_curve(name, api.CURVE_CAUCHY, 0, a, b)

    def curve_blackbody(self, name, temperature, strength):
        return self._add_curve(name, api.CURVE_BLACKBODY, 0, temperature, strength)

    def curve_exponential(self, name, spikes):  # [(lambda, left_taper, right_taper, strength)]
        return self._add_curve(name, api.CURVE_EXPONENTIAL, 0, data=[v for s in spikes for v in s])

    def curve_simple_spike(self, name, lam, left_taper, right_taper, strength):
        return self.curve_exponential(name, [(lam, left_taper, right_taper, strength)])

    def curve_tabulated(self, name, xs, ys, mode=api.INTERP_CUBIC, x_scale=1.0, x_offset=0.0, y_scale=1.0, y_offset=0.0):
        f32 = np.float32
        data = []
        for x, y in zip(xs, ys):  # DomainMapping: (x - offset) * scale, in f32 as the reference parses
            data.append(float((f32(x) - f32(x_offset)) * f32(x_scale)))
            data.append(float((f32(y) - f32(y_offset)) * f32(y_scale)))
        return self._add_curve(name, api.CURVE_TABULATED, mode, data=data)

    def curve_linear(self, name, start, step, ys, mode=api.INTERP_CUBIC, y_scale=1.0):
        f32 = np.float32
        end = float(f32(start) + f32(step) * f32(len(ys)))
        return self._add_curve(name, api.CURVE_LINEAR, mode, float(start), end, data=[float(f32(y) * f32(y_scale)) for y in ys])

    def curve(self, name):
        return self.curve_names[name]

    # ---- textures (src/parsing/texture.rs)
    def texstack_texture1(self, name, curve, texels=None):
        t = np.ones((1, 1), np.float32) if texels is None else np.asarray(texels, np.float32)
        layer = api.TextureLayer(api.TEXTURE1, (C.c_int32 * 4)(curve, -1, -1, -1), t.shape[1], t.shape[0], len(self.texture_data))
        self.texture_data.extend(t.reshape(-1).tolist())
        self.layers.append(layer)
        self.texstacks.append(api.TexStack(len(self.layers) - 1, 1))
        self.texstack_names[name] = len(self.texstacks) - 1
        return len(self.texstacks) - 1

    def texstack_texture4(self, name, curves, texels):
        t = np.asarray(texels, np.float32)
        assert t.ndim == 3 and t.shape[2] == 4
        layer = api.TextureLayer(api.TEXTURE4, (C.c_int32 * 4)(*curves), t.shape[1], t.shape[0], len(self.texture_data))
        self.texture_data.extend(t.reshape(-1).tolist())
        self.layers.append(layer)
        self.texstacks.append(api.TexStack(len(self.layers) - 1, 1))
        self.texstack_names[name] = len(self.texstacks) - 1
        return len(self.texstacks) - 1

    # ---- materials (src/parsing/material.rs:66-153; ids src/parsing/mod.rs:456-467)
    def _add_material(self, name, m, is_light):
        self.materials.append(m)
        idx = len(self.materials) - 1
        self.material_ids[name] = api.material_id(api.TAG_LIGHT if is_light else api.TAG_MATERIAL, idx)
        return self.material_ids[name]

    def material_lambertian(self, name, texstack):
        return self._add_material(name, api.Material(api.MATERIAL_LAMBERTIAN, texstack, 0.0, -1, -1, -1, -1, -1, 0.0, 0), False)

    def material_ggx(self, name, alpha, eta, eta_o, kappa, outer_medium=0, inner_medium=0):
        return self._add_material(name, api.Material(api.MATERIAL_GGX, -1, alpha, eta, eta_o, kappa, -1, -1, 0.0, 0, outer_medium, inner_medium), False)

    def material_passthrough(self, name, color, outer_medium=0, inner_medium=0):   # PassthroughFilter (src/materials/passthrough.rs)
        return self._add_material(name, api.Material(api.MATERIAL_PASSTHROUGH, -1, 0.0, -1, -1, -1, -1, color, 0.0, 0, outer_medium, inner_medium), False)

    # ---- mediums (src/parsing/medium.rs; MediumId = position + 1, src/parsing/mod.rs)
    def medium_hg(self, name, g, sigma_a, sigma_s):
        self.mediums.append(api.Medium(api.MEDIUM_HG, g, sigma_a, sigma_s, -1, 0.0))
        self.medium_ids[name] = len(self.mediums)
        return self.medium_ids[name]

    def medium_rayleigh(self, name, ior, corrective_factor):
        self.mediums.append(api.Medium(api.MEDIUM_RAYLEIGH, -1, -1, -1, ior, corrective_factor))
        self.medium_ids[name] = len(self.mediums)
        return self.medium_ids[name]

    def material_diffuse_light(self, name, emit, bounce, sidedness):
        return self._add_material(name, api.Material(api.MATERIAL_DIFFUSE_LIGHT, -1, 0.0, -1, -1, -1, emit, bounce, 0.0, sidedness), True)

    def material_sharp_light(self, name, emit, bounce, sharpness, sidedness):
        return self._add_material(name, api.Material(api.MATERIAL_SHARP_LIGHT, -1, 0.0, -1, -1, -1, emit, bounce, sharpness, sidedness), True)

    def material(self, name):
        return self.material_ids[name]

    # ---- geometry
    def add_mesh(self, positions, faces, normals=None, face_materials=None):
        p = np.asarray(positions, np.float32).reshape(-1, 3)
        f = np.asarray(faces, np.uint32).reshape(-1, 3)
        m = api.Mesh(len(self.vertices) // 3, p.shape[0], len(self.indices), f.shape[0], -1, -1)
        if normals is not None and len(normals):
            n = np.asarray(normals, np.float32).reshape(-1, 3)
            assert n.shape == p.shape
            m.normal_offset = len(self.normals) // 3
            self.normals.extend(n.reshape(-1).tolist())
        if face_materials is not None:
            fm = np.broadcast_to(np.asarray(face_materials, np.uint32), (f.shape[0],))
            m.face_material_offset = len(self.face_materials)
            self.face_materials.extend(fm.tolist())
        self.vertices.extend(p.reshape(-1).tolist())
        self.indices.extend(f.reshape(-1).tolist())
        self.meshes.append(m)
        return len(self.meshes) - 1

    def _instance(self, kind, material, transform):
        inst = api.Instance()
        inst.kind = kind
        inst.material = api.MATERIAL_NONE if material is None else material
        fwd = _IDENTITY if transform is None else np.asarray(transform, np.float64)
        inst.has_transform = 0 if transform is None else 1
        inst.forward = (C.c_float * 16)(*fwd.astype(np.float32).reshape(-1).tolist())
        inst.reverse = (C.c_float * 16)(*transform_inverse(fwd).astype(np.float32).reshape(-1).tolist())
        self.instances.append(inst)
        return inst

    def add_rect(self, size, origin, axis, two_sided, material, transform=None):
        inst = self._instance(api.SHAPE_RECT, material, transform)
        inst.size = (C.c_float * 2)(*size); inst.origin = (C.c_float * 3)(*origin)
        inst.axis = {"X": 0, "Y": 1, "Z": 2}[axis] if isinstance(axis, str) else axis
        inst.two_sided = int(two_sided)
        return len(self.instances) - 1

    def add_sphere(self, radius, origin, material, transform=None):
        inst = self._instance(api.SHAPE_SPHERE, material, transform)
        inst.radius = radius; inst.origin = (C.c_float * 3)(*origin)
        return len(self.instances) - 1

    def add_disk(self, radius, origin, two_sided, material, transform=None):
        inst = self._instance(api.SHAPE_DISK, material, transform)
        inst.radius = radius; inst.origin = (C.c_float * 3)(*origin); inst.two_sided = int(two_sided)
        return len(self.instances) - 1

    def add_mesh_instance(self, mesh, material=None, transform=None):
        inst = self._instance(api.SHAPE_MESH, material, transform)
        inst.mesh = mesh
        return len(self.instances) - 1

    def add_camera(self, look_from, look_at, vfov, focal_distance=10.0, aperture_diameter=0.01, v_up=(0.0, 0.0, 1.0)):
        self.cameras.append(api.Camera((C.c_float * 3)(*look_from), (C.c_float * 3)(*look_at), (C.c_float * 3)(*v_up),
                                       vfov, focal_distance, aperture_diameter))
        return len(self.cameras) - 1

    def add_panorama_camera(self, look_from, look_at, fov, v_up=(0.0, 0.0, 1.0)):
        """PanoramaCameraData (src/parsing/cameras.rs:86-94): fov = (horizontal, vertical) in degrees."""
        up = np.asarray(v_up, np.float32)
        up = up / np.sqrt(up[0] * up[0] + up[1] * up[1] + up[2] * up[2], dtype=np.float32)
        self.cameras.append(api.Camera((C.c_float * 3)(*look_from), (C.c_float * 3)(*look_at), (C.c_float * 3)(*up.tolist()),
                                       0.0, 0.0, 0.0, api.CAMERA_PANORAMA, (C.c_float * 2)(*fov)))
        return len(self.cameras) - 1

    def set_environment_constant(self, curve, strength):
        self.environment.kind = api.ENV_CONSTANT
        self.environment.curve = curve
        self.environment.strength = strength

    def set_environment_sun(self, curve, strength, angular_diameter, sun_direction):
        self.environment.kind = api.ENV_SUN
        self.environment.curve = curve
        self.environment.strength = strength
        self.environment.angular_diameter = angular_diameter
        d = np.asarray(sun_direction, np.float32)  # Vec3::normalized in f32 (src/parsing/environment.rs:91)
        n = np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2], dtype=np.float32)
        self.environment.sun_direction = (C.c_float * 3)(*(d / n).tolist())

    def set_environment_hdr(self, texstack, strength, rotate=None, importance=(0, 0), luminance_curve=-1):
        """EnvironmentData::HDRI (src/parsing/environment.rs:93-180): texture stack, rotation list [(axis, degrees)],
        importance map (width, height)."""
        self.environment.kind = api.ENV_HDR
        self.environment.curve = -1
        self.environment.texstack = texstack
        self.environment.strength = strength
        fwd = transform_from_data(rotate=rotate) if rotate else np.eye(4)
        self.environment.rotation_forward = (C.c_float * 16)(*fwd.astype(np.float32).reshape(-1).tolist())
        self.environment.rotation_reverse = (C.c_float * 16)(*transform_inverse(fwd).astype(np.float32).reshape(-1).tolist())
        self.environment.importance_width, self.environment.importance_height = importance
        self.environment.importance_luminance_curve = luminance_curve

    # ---- flatten
    def desc(self):
        keep = {}

        def arr(name, values, ctype):
            n = max(len(values), 1)
            a = (ctype * n)(*values) if len(values) else (ctype * n)()
            keep[name] = a
            return a

        d = api.SceneDesc()
        d.curve_count = len(self.curves); d.curves = arr("curves", self.curves, api.Curve)
        cd = np.asarray(self.curve_data, np.float32); keep["cd"] = cd
        d.curve_data_count = cd.size; d.curve_data = cd.ctypes.data_as(C.POINTER(C.c_float))
        d.layer_count = len(self.layers); d.layers = arr("layers", self.layers, api.TextureLayer)
        d.texstack_count = len(self.texstacks); d.texstacks = arr("texstacks", self.texstacks, api.TexStack)
        td = np.asarray(self.texture_data, np.float32); keep["td"] = td
        d.texture_data_count = td.size; d.texture_data = td.ctypes.data_as(C.POINTER(C.c_float))
        d.material_count = len(self.materials); d.materials = arr("materials", self.materials, api.Material)
        d.mesh_count = len(self.meshes); d.meshes = arr("meshes", self.meshes, api.Mesh)
        vs = np.asarray(self.vertices, np.float32); keep["vs"] = vs
        d.vertex_count = vs.size // 3; d.vertices = vs.ctypes.data_as(C.POINTER(C.c_float))
        ix = np.asarray(self.indices, np.uint32); keep["ix"] = ix
        d.index_count = ix.size; d.indices = ix.ctypes.data_as(C.POINTER(C.c_uint32))
        ns = np.asarray(self.normals, np.float32); keep["ns"] = ns
        d.normal_count = ns.size // 3; d.normals = ns.ctypes.data_as(C.POINTER(C.c_float))
        fm = np.asarray(self.face_materials, np.uint32); keep["fm"] = fm
        d.face_material_count = fm.size; d.face_materials = fm.ctypes.data_as(C.POINTER(C.c_uint32))
        d.instance_count = len(self.instances); d.instances = arr("instances", self.instances, api.Instance)
        d.camera_count = len(self.cameras); d.cameras = arr("cameras", self.cameras, api.Camera)
        d.environment = self.environment
        d.env_sampling_probability = self.env_sampling_probability
        d.medium_count = len(self.mediums); d.mediums = arr("mediums", self.mediums, api.Medium)
        return d, keep


# =================================================================== library of named assets
def add_library_curves(b, names):
    """The subset of data/lib_curves.toml the benchmark scenes use."""
    sp = _spectra()
    for n in names:
        if n in b.curve_names:
            continue
        if n == "flat_zero": b.curve_flat(n, 0.0)
        elif n == "flat_one": b.curve_flat(n, 1.0)
        elif n == "flat_78": b.curve_flat(n, 0.78)
        elif n == "E5": b.curve_flat(n, 5.0)
        elif n == "E10": b.curve_flat(n, 10.0)
        elif n == "D65": b.curve_tabulated(n, sp["tabulated"][n]["x"], sp["tabulated"][n]["y"])   # data/lib_curves.toml:1-5 (TabulatedCSV, column 1, Cubic)
        elif n == "540THz": b.curve_simple_spike(n, 555.17, 1.0, 1.0, 1.0)                         # data/lib_curves.toml:23-28
        elif n == "simple_blue": b.curve_simple_spike(n, 450.0, 50.0, 50.0, 0.55)                 # data/lib_curves.toml:91-96
        elif n == "simple_yellow": b.curve_simple_spike(n, 600.0, 50.0, 50.0, 0.55)               # data/lib_curves.toml:107-112
        elif n == "air_ior": b.curve_cauchy(n, 1.0002724293, 1.64748969205)
        elif n in ("cornell_white", "cornell_green", "cornell_red", "cornell_light", "srgb_r", "srgb_g", "srgb_b"):
            b.curve_tabulated(n, sp["tabulated"][n]["x"], sp["tabulated"][n]["y"])
        elif n in ("gold_n", "gold_k", "copper_n", "copper_k"):
            b.curve_tabulated(n, sp["tabulated"][n]["x"], sp["tabulated"][n]["y"], x_scale=1000.0)
        elif n == "simple_sky_blue": b.curve_simple_spike(n, 500.0, 100.0, 100.0, 0.55)
        elif n == "blackbody_5000k": b.curve_blackbody(n, 5000.0, 1.0)
        elif n == "blackbody_3000k_x5": b.curve_blackbody(n, 3000.0, 5.0)
        elif n == "fluorescent_x5":
            l = sp["linear"]["fluorescent"]; b.curve_linear(n, l["start"], l["step"], l["y"], y_scale=5.0)
        elif n == "xenon_x5":
            l = sp["linear"]["xenon_lamp"]; b.curve_linear(n, l["start"], l["step"], l["y"], y_scale=5.0)
        else:
            raise KeyError(n)


def add_library_material(b, name):
    """The subset of data/lib_materials.toml (+ lib_textures.toml) the benchmark scenes use."""
    if name in b.material_ids:
        return b.material_ids[name]
    if name.startswith("lambertian_"):
        curve = {"lambertian_white": "cornell_white", "lambertian_green": "cornell_green", "lambertian_red": "cornell_red",
                 "lambertian_blue": "simple_blue", "lambertian_yellow": "simple_yellow"}[name]   # data/lib_textures.toml:36-60
        add_library_curves(b, [curve])
        ts = b.texstack_texture1(name, b.curve(curve))  # Texture1 over single_pixel.png (1x1 white -> factor 1.0)
        return b.material_lambertian(name, ts)
    if name == "diffuse_light_cornell":
        add_library_curves(b, ["cornell_light", "flat_78"])
        return b.material_diffuse_light(name, b.curve("cornell_light"), b.curve("flat_78"), api.SIDED_REVERSE)
    if name == "diffuse_light":                 # data/lib_materials.toml:306-310
        add_library_curves(b, ["blackbody_5000k", "flat_78"])
        return b.material_diffuse_light(name, b.curve("blackbody_5000k"), b.curve("flat_78"), api.SIDED_DUAL)
    if name == "diffuse_light_flat_x5":
        add_library_curves(b, ["E5", "flat_78"])
        return b.material_diffuse_light(name, b.curve("E5"), b.curve("flat_78"), api.SIDED_DUAL)
    if name == "diffuse_light_flat_x10":        # data/lib_materials.toml:279-283
        add_library_curves(b, ["E10", "flat_78"])
        return b.material_diffuse_light(name, b.curve("E10"), b.curve("flat_78"), api.SIDED_DUAL)
    if name == "diffuse_540THz":                # data/lib_materials.toml:313-317 (the candela's monochromatic source)
        add_library_curves(b, ["540THz", "flat_zero"])
        return b.material_diffuse_light(name, b.curve("540THz"), b.curve("flat_zero"), api.SIDED_DUAL)
    if name == "ggx_air_glass":                 # data/lib_materials.toml:40-49: air inside, glass outside (the inner surface of a hollow ball)
        add_library_curves(b, ["air_ior", "flat_zero"])
        eta_o = b.curve_cauchy(name + ".eta_o", 1.4, 4500.0)
        return b.material_ggx(name, 0.0004, b.curve("air_ior"), eta_o, b.curve("flat_zero"))
    if name == "sharp_light_fluorescent":
        add_library_curves(b, ["fluorescent_x5", "flat_78"])
        return b.material_sharp_light(name, b.curve("fluorescent_x5"), b.curve("flat_78"), 40.0, api.SIDED_REVERSE)
    if name == "sharp_light_xenon":
        add_library_curves(b, ["xenon_x5", "flat_78"])
        return b.material_sharp_light(name, b.curve("xenon_x5"), b.curve("flat_78"), 100.0, api.SIDED_DUAL)
    if name == "sharp_light":
        add_library_curves(b, ["blackbody_5000k", "flat_78"])
        return b.material_sharp_light(name, b.curve("blackbody_5000k"), b.curve("flat_78"), 400.0, api.SIDED_REVERSE)
    ggx = {"ggx_glass": (0.0004, 1.4, 4500.0), "ggx_glass_rough": (0.2, 1.4, 4500.0),
           "ggx_glass_dispersive": (0.0004, 1.4, 50000.0), "ggx_moissanite": (0.0004, 2.4, 34000.0)}
    if name in ggx:
        alpha, a, c = ggx[name]
        add_library_curves(b, ["air_ior", "flat_zero"])
        eta = b.curve_cauchy(name + ".eta", a, c)
        return b.material_ggx(name, alpha, eta, b.curve("air_ior"), b.curve("flat_zero"))
    metals = {"ggx_gold": ("gold", 0.004), "ggx_copper": ("copper", 0.002)}
    if name in metals:
        base, alpha = metals[name]
        add_library_curves(b, ["air_ior", base + "_n", base + "_k"])
        return b.material_ggx(name, alpha, b.curve(base + "_n"), b.curve("air_ior"), b.curve(base + "_k"))
    raise KeyError(name)


# =================================================================== benchmark scenes
# The Cornell box measurements (Cornell University Program of Computer Graphics), millimetres, in the
# original axes (x right-to-left, y up, z depth).  The reference's scene file places its camera and light
# in metres with (x, y, z)_scene = (z, x, y)_cornell / 1000 (data/scenes/cornell_box.toml:13-38), and its
# data/meshes/cornell_box.obj is absent from the tree, so the mesh is authored here.
_CORNELL_QUADS = [
    ("floor", "lambertian_white", [(552.8, 0, 0), (0, 0, 0), (0, 0, 559.2), (549.6, 0, 559.2)]),
    ("ceiling", "lambertian_white", [(556, 548.8, 0), (556, 548.8, 559.2), (0, 548.8, 559.2), (0, 548.8, 0)]),
    ("back_wall", "lambertian_white", [(549.6, 0, 559.2), (0, 0, 559.2), (0, 548.8, 559.2), (556, 548.8, 559.2)]),
    ("right_wall", "lambertian_green", [(0, 0, 559.2), (0, 0, 0), (0, 548.8, 0), (0, 548.8, 559.2)]),
    ("left_wall", "lambertian_red", [(552.8, 0, 0), (549.6, 0, 559.2), (556, 548.8, 559.2), (556, 548.8, 0)]),
    ("short_block", "lambertian_white", [
        (130, 165, 65), (82, 165, 225), (240, 165, 272), (290, 165, 114),
        (290, 0, 114), (290, 165, 114), (240, 165, 272), (240, 0, 272),
        (130, 0, 65), (130, 165, 65), (290, 165, 114), (290, 0, 114),
        (82, 0, 225), (82, 165, 225), (130, 165, 65), (130, 0, 65),
        (240, 0, 272), (240, 165, 272), (82, 165, 225), (82, 0, 225)]),
    ("tall_block", "lambertian_white", [
        (423, 330, 247), (265, 330, 296), (314, 330, 456), (472, 330, 406),
        (423, 0, 247), (423, 330, 247), (472, 330, 406), (472, 0, 406),
        (472, 0, 406), (472, 330, 406), (314, 330, 456), (314, 0, 456),
        (314, 0, 456), (314, 330, 456), (265, 330, 296), (265, 0, 296),
        (265, 0, 296), (265, 330, 296), (423, 330, 247), (423, 0, 247)]),
]


def cornell_obj_text():
    """The authored cornell_box.obj (+ usemtl names from data/lib_materials.toml), scene axes, metres."""
    lines = ["# Cornell box, authored from the published measurements; metres; (x,y,z) = (depth, width, up)",
             "mtllib cornell_box.mtl"]
    nv = 0
    for name, mtl, pts in _CORNELL_QUADS:
        lines.append("o " + name)
        lines.append("usemtl " + mtl)
        uniq = []
        for p in pts:
            if p not in uniq:
                uniq.append(p)
        for (x, y, z) in uniq:
            lines.append("v %.4f %.4f %.4f" % (z / 1000.0, x / 1000.0, y / 1000.0))
        for q in range(0, len(pts), 4):
            lines.append("f " + " ".join(str(nv + 1 + uniq.index(p)) for p in pts[q:q + 4]))
        nv += len(uniq)
    return "\n".join(lines) + "\n"


def cornell_box():
    """data/scenes/cornell_box.toml with the authored mesh: C1 / C2 / C5 of BASELINE.json."""
    import tempfile
    b = SceneBuilder()
    add_library_curves(b, ["flat_zero"])
    b.set_environment_constant(b.curve("flat_zero"), 0.0)
    b.env_sampling_probability = 0.0
    light = add_library_material(b, "diffuse_light_cornell")
    for n in ("lambertian_green", "lambertian_red", "lambertian_white"):
        add_library_material(b, n)
    b.add_rect((0.105, 0.13), (0.278, 0.2795, 0.5487), "Z", False, light)
    with tempfile.NamedTemporaryFile("w", suffix=".obj", delete=False) as f:
        f.write(cornell_obj_text())
        path = f.name
    try:
        models = load_obj(path)
    finally:
        os.unlink(path)
    # mesh bundle "cornell_box;i": one instance per tobj model, material from the .mtl name (src/parsing/mod.rs:504-535)
    for m in models:
        p, n, f = m.arrays()
        mesh = b.add_mesh(p, f, n, face_materials=b.material(m.material))
        b.add_mesh_instance(mesh, None, None)
    b.add_camera((-0.8, 0.278, 0.273), (0.0, 0.278, 0.273), 37.8, focal_distance=1.1, aperture_diameter=0.01)
    return b


def _npz_mesh(name):
    z = np.load(os.path.join(_DATA, "meshes", name + ".npz"))
    n = z["normals"]
    return z["positions"], z["faces"], (n if n.shape[0] else None), str(z["material"])


def cornell_gem(gem_z=-0.7):
    """data/scenes/cornell_box_diamond_gem.toml (C3): env replaced by Constant 0 (its HDRI file is absent and
    env_sampling_probability = 0).  `gem_z`: the gem's translation along z — the scene file's -0.7 (culet below the floor);
    the reference's showcase/moissanite_gem_1080p.png shows an earlier placement with the whole gem above the floor (tests only)."""
    b = SceneBuilder()
    add_library_curves(b, ["flat_zero"])
    b.set_environment_constant(b.curve("flat_zero"), 0.0)
    b.env_sampling_probability = 0.0
    light = add_library_material(b, "sharp_light_fluorescent")
    white = add_library_material(b, "lambertian_white")
    red = add_library_material(b, "lambertian_red")
    green = add_library_material(b, "lambertian_green")
    gem = add_library_material(b, "ggx_moissanite")
    b.add_rect((0.4, 0.4), (0.0, 0.0, 0.9), "Z", False, light)
    b.add_rect((2, 2), (0.0, 0.0, 1.0), "Z", True, white)
    b.add_rect((2, 2), (0.0, 0.0, -1.0), "Z", True, white)
    b.add_rect((2, 2), (0.0, 1.0, 0.0), "Y", True, red)
    b.add_rect((2, 2), (0.0, -1.0, 0.0), "Y", True, green)
    b.add_rect((2, 2), (1.0, 0.0, 0.0), "X", True, white)
    p, f, n, mtl = _npz_mesh("brilliant_diamond")
    mesh = b.add_mesh(p, f, n, face_materials=b.material(mtl))
    b.add_mesh_instance(mesh, gem, transform_from_data(scale=(0.5, 0.5, 0.5), translate=(0.0, 0.0, gem_z)))
    b.add_camera((-5.0, 0.0, 0.0), (0.0, 0.0, 0.0), 27.8, focal_distance=5.0, aperture_diameter=0.02)
    return b


def white_furnace(material="ggx_glass_rough"):
    """data/scenes/white_furnace.toml: camera inside a non-absorbing sphere in a constant environment."""
    b = SceneBuilder()
    add_library_curves(b, ["simple_sky_blue"])
    b.set_environment_constant(b.curve("simple_sky_blue"), 1.0)
    b.env_sampling_probability = 1.0
    m = add_library_material(b, material)
    b.add_sphere(1.0, (0.0, 0.0, 0.0), m)
    b.add_camera((0.5, 0.0, 0.0), (0.0, 0.0, 0.0), 70.4, focal_distance=0.5, aperture_diameter=0.001)
    return b


def mixed_primitives():
    """A small scene that exercises every primitive kind, transforms, metals and a dual-sided light
    (parity-test coverage for Instance/Aggregate dispatch; not a reference scene)."""
    b = SceneBuilder()
    add_library_curves(b, ["simple_sky_blue"])
    b.set_environment_constant(b.curve("simple_sky_blue"), 0.3)
    b.env_sampling_probability = 0.25
    light = add_library_material(b, "diffuse_light_flat_x5")
    white = add_library_material(b, "lambertian_white")
    red = add_library_material(b, "lambertian_red")
    gold = add_library_material(b, "ggx_gold")
    glass = add_library_material(b, "ggx_glass_rough")
    b.add_rect((6, 6), (0.0, 0.0, -1.0), "Z", True, white)
    b.add_rect((1.0, 1.0), (0.0, 0.0, 2.5), "Z", True, light)
    b.add_disk(0.7, (0.0, 0.0, 0.0), True, light,
               transform_from_data(rotate=[((0, 1, 0), 70.0)], translate=(2.0, 1.5, 1.0)))
    b.add_sphere(0.6, (0.0, -1.2, -0.4), gold)
    b.add_sphere(0.5, (0.0, 0.0, 0.0), glass, transform_from_data(scale=(1.0, 1.4, 0.8), translate=(0.3, 1.1, -0.3)))
    b.add_rect((2.0, 3.0), (0.0, 2.5, 0.5), "Y", True, red)
    p, f, n, mtl = _npz_mesh("gem")
    mesh = b.add_mesh(p, f, n, face_materials=api.material_id(api.TAG_MATERIAL, 0))
    b.add_mesh_instance(mesh, glass, transform_from_data(scale=(0.5, 0.5, 0.5), rotate=[((0, 0, 1), 30.0), ((1, 0, 0), 15.0)],
                                                         translate=(-0.8, 0.0, -0.5)))
    b.add_camera((-5.0, 0.3, 0.8), (0.0, 0.0, 0.0), 35.0, focal_distance=5.0, aperture_diameter=0.05)
    return b


def _octahedron():
    """Unit octahedron with per-vertex normals (8 faces)."""
    p = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)
    f = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]], np.uint32)
    return p, f, p.copy()


def mixed_small():
    """mixed_primitives at leaf-sweep size (<= 64 leaves in total, csrc/pt_device.h world_hit_sweep): every primitive kind,
    transformed and untransformed meshes, vertex normals, axis-aligned and slanted triangles (not a reference scene)."""
    b = SceneBuilder()
    add_library_curves(b, ["simple_sky_blue"])
    b.set_environment_constant(b.curve("simple_sky_blue"), 0.3)
    b.env_sampling_probability = 0.25
    light = add_library_material(b, "diffuse_light_flat_x5")
    white = add_library_material(b, "lambertian_white")
    red = add_library_material(b, "lambertian_red")
    gold = add_library_material(b, "ggx_gold")
    glass = add_library_material(b, "ggx_glass_rough")
    b.add_rect((6, 6), (0.0, 0.0, -1.0), "Z", True, white)
    b.add_rect((1.0, 1.0), (0.0, 0.0, 2.5), "Z", True, light)
    b.add_disk(0.7, (0.0, 0.0, 0.0), True, light,
               transform_from_data(rotate=[((0, 1, 0), 70.0)], translate=(2.0, 1.5, 1.0)))
    b.add_sphere(0.6, (0.0, -1.2, -0.4), gold)
    b.add_sphere(0.5, (0.0, 0.0, 0.0), glass, transform_from_data(scale=(1.0, 1.4, 0.8), translate=(0.3, 1.1, -0.3)))
    b.add_rect((2.0, 3.0), (0.0, 2.5, 0.5), "Y", True, red)
    p, f, n = _octahedron()
    octa = b.add_mesh(p, f, n, face_materials=api.material_id(api.TAG_MATERIAL, 0))
    b.add_mesh_instance(octa, glass, transform_from_data(scale=(0.5, 0.6, 0.7), rotate=[((0, 0, 1), 30.0), ((1, 0, 0), 15.0)],
                                                         translate=(-0.8, 0.0, -0.3)))
    b.add_mesh_instance(octa, gold, transform_from_data(scale=(0.3, 0.3, 0.3), translate=(0.2, -0.2, 0.9)))
    # an untransformed mesh with faces in the coordinate planes (zero-thickness leaf boxes) and per-face materials
    q = np.array([[1.0, -2.0, -1.0], [2.0, -2.0, -1.0], [2.0, -1.0, -1.0], [1.0, -1.0, -1.0], [1.0, -2.0, 0.2], [2.0, -2.0, 0.2]], np.float32)
    g = np.array([[0, 1, 4], [1, 5, 4], [0, 4, 3], [1, 2, 5], [3, 4, 5], [3, 5, 2]], np.uint32)
    fm = np.array([red, white, red, white, gold, gold], np.uint32)
    wedge = b.add_mesh(q, g, None, face_materials=fm)
    b.add_mesh_instance(wedge)
    b.add_camera((-5.0, 0.3, 0.8), (0.0, 0.0, 0.0), 35.0, focal_distance=5.0, aperture_diameter=0.05)
    return b


def big_sphere_light():
    """A sphere light of radius 20 000 hanging 0.02 above a floor (not a reference scene): from the floor under it a light sample's hit
    distance is 0.02 .. 1 while the terms of the sphere's quadratic are ~4e8 (|oc|^2 - r^2) and ~2e4 (-b, the root), so the computed
    distance is good to ~1e-3 at best — far more than any margin relative to the distance.  A box that holds the sphere must never
    be culled by a distance computed from the sphere itself (csrc/pt_device.h `beyond`)."""
    b = SceneBuilder()
    add_library_curves(b, ["simple_sky_blue"])
    b.set_environment_constant(b.curve("simple_sky_blue"), 0.0)
    b.env_sampling_probability = 0.0
    light = add_library_material(b, "diffuse_light_flat_x5")
    white = add_library_material(b, "lambertian_white")
    red = add_library_material(b, "lambertian_red")
    b.add_rect((40, 40), (0.0, 0.0, 0.0), "Z", True, white)
    b.add_sphere(20000.0, (0.0, 0.0, 20000.02), light)
    b.add_rect((1.0, 1.0), (3.0, 0.0, 0.5), "X", True, red)
    b.add_rect((1.0, 1.0), (0.0, 3.0, 0.5), "Y", True, red)
    b.add_camera((-6.0, 0.5, 0.4), (0.0, 0.0, 0.1), 40.0, focal_distance=6.0, aperture_diameter=0.01)
    return b


def disk_lamp():
    """One disk lamp (one-sided, facing down) 1e-4 under a ceiling rect, a floor, a Lambertian ball and a wall (not a reference scene): the lean vertex form of a scene
    with ONE light of a shape other than a rect — its light-sample rays are tested against the lamp where they are made (pt_stages.h, the one-light test), and the ceiling's
    vertices, whose rays start below the lamp they aim at, are the case that test exists for."""
    b = SceneBuilder()
    add_library_curves(b, ["flat_zero"])
    b.set_environment_constant(b.curve("flat_zero"), 0.0)
    b.env_sampling_probability = 0.0
    light = add_library_material(b, "diffuse_light_flat_x5")
    white = add_library_material(b, "lambertian_white")
    red = add_library_material(b, "lambertian_red")
    b.add_disk(0.25, (0.0, 0.0, 0.9999), False, light)
    b.add_rect((2.0, 2.0), (0.0, 0.0, 1.0), "Z", True, white)      # ceiling
    b.add_rect((2.0, 2.0), (0.0, 0.0, 0.0), "Z", True, white)      # floor
    b.add_rect((2.0, 1.0), (1.0, 0.0, 0.5), "X", True, red)        # back wall
    b.add_sphere(0.25, (0.2, 0.3, 0.25), white)
    b.add_camera((-2.4, 0.0, 0.5), (0.0, 0.0, 0.45), 45.0, focal_distance=2.4, aperture_diameter=0.01)
    return b


def fog_ball():
    """The medium-aware walk (pt_render_desc.medium_aware; not a reference scene): a ball of forward-scattering HG fog and one of
    Rayleigh-scattering air behind PassthroughFilter boundaries, a rough glass ball filled with absorbing fog (GGX carries medium ids
    too), a white floor, a constant sky as the only light — the reference's medium-aware colour panics at any light vertex that faces
    the path."""
    b = SceneBuilder()
    add_library_curves(b, ["simple_sky_blue"])
    b.set_environment_constant(b.curve("simple_sky_blue"), 1.0)
    b.env_sampling_probability = 1.0
    white = add_library_material(b, "lambertian_white")
    one = b.curve_flat("flat_one_fog", 1.0)
    g_forward = b.curve_flat("hg_g_forward", 1.6)     # the library stores g + 1 (hg.rs:20-22)
    zero = b.curve_flat("flat_zero_fog", 0.0)
    dense = b.curve_flat("sigma_s_dense", 1.5)
    thin = b.curve_flat("sigma_a_thin", 0.4)
    air = b.curve_cauchy("air_ior_fog", 1.5, 0.0)
    fog = b.medium_hg("fog", g_forward, zero, dense)
    haze = b.medium_rayleigh("haze", air, 0.6)
    murk = b.medium_hg("murk", b.curve_flat("hg_g_iso", 1.0), thin, b.curve_flat("sigma_s_murk", 0.8))
    boundary_fog = b.material_passthrough("fog_boundary", one, 0, fog)
    boundary_haze = b.material_passthrough("haze_boundary", one, 0, haze)
    eta = b.curve_cauchy("glass_eta_fog", 1.45, 3540.0)
    glass = b.material_ggx("ggx_glass_murky", 0.2, eta, b.curve_flat("eta_o_fog", 1.0), zero, 0, murk)
    b.add_rect((12, 12), (0.0, 0.0, -1.0), "Z", True, white)
    b.add_sphere(0.9, (0.0, -1.1, 0.0), boundary_fog)
    b.add_sphere(0.8, (0.3, 1.0, -0.2), boundary_haze)
    b.add_sphere(0.5, (-1.2, 0.1, -0.5), glass)
    b.add_camera((-5.0, 0.0, 0.6), (0.0, 0.0, -0.1), 40.0, focal_distance=5.0, aperture_diameter=0.02)
    return b


def empty_env():
    """No instance at all: a constant environment and a camera (edge case: empty BVH, empty light list)."""
    b = SceneBuilder()
    add_library_curves(b, ["flat_one"])
    b.set_environment_constant(b.curve("flat_one"), 1.0)
    b.add_camera((0.0, 0.0, 0.0), (1.0, 0.0, 0.0), 40.0)
    return b


def panorama_test():
    """The Cornell box seen by a PanoramaCamera from its centre (not a reference scene)."""
    b = cornell_box()
    b.cameras.clear()
    b.add_panorama_camera((0.28, 0.28, 0.27), (1.0, 0.28, 0.27), (360.0, 180.0))
    return b


def sun_test():
    """data/scenes/sun_test.toml of this repository: Sun environment with a literal blackbody colour, a metal sphere on a
    ground rect (not a reference scene; covers EnvironmentData::Sun and camera defaults)."""
    b = SceneBuilder()
    sun = b.curve_blackbody(None, 5800.0, 1.0)
    b.set_environment_sun(sun, 2.0, 0.1, (0.3, -0.2, 1.0))
    white = add_library_material(b, "lambertian_white")
    copper = add_library_material(b, "ggx_copper")
    b.add_rect((8, 8), (0.0, 0.0, -1.0), "Z", True, white)
    b.add_sphere(0.8, (0.0, 0.0, -0.2), copper)
    b.add_camera((-5.0, 0.0, 1.0), (0.0, 0.0, 0.0), 30.0)
    return b


def synthetic_hdri(width=1024, height=512):
    """Deterministic stand-in for the absent data/hdri/*.hdr: a sky gradient, a warm ground and one
    Gaussian "sun", linear RGB + alpha 0, float32, generated from a closed formula (no RNG).  Texel (x, y) is looked up
    with uv = (x / width, y / height); u is the azimuth, v the polar angle from +Z (math::uv_to_direction)."""
    u = (np.arange(width, dtype=np.float64) + 0.5) / width
    v = (np.arange(height, dtype=np.float64) + 0.5) / height
    uu, vv = np.meshgrid(u, v)
    up = np.cos(np.pi * vv)                                  # +1 zenith, -1 nadir
    sky = np.clip(up, 0, 1)[..., None] * np.array([0.25, 0.45, 1.0]) + (1 - np.clip(up, 0, 1))[..., None] * np.array([0.9, 0.85, 0.8])
    ground = np.array([0.22, 0.18, 0.12]) * (0.4 + 0.6 * np.clip(-up, 0, 1))[..., None]
    rgb = np.where((up > 0)[..., None], 0.6 * sky, ground)
    su, sv = 0.65, 0.27                                      # sun position
    d2 = (np.minimum(np.abs(uu - su), 1 - np.abs(uu - su)) * 2 * np.sin(np.pi * vv)) ** 2 + (vv - sv) ** 2
    sun = 60.0 * np.exp(-d2 / (2 * 0.012 ** 2))
    rgb = rgb + sun[..., None] * np.array([1.0, 0.92, 0.8])
    out = np.zeros((height, width, 4), np.float32)
    out[..., :3] = rgb
    return out


def hdri_test(mesh="monkey", hdri_size=(1024, 512), importance=(1024, 1024), env_sampling_probability=0.9):
    """data/scenes/hdri_test.toml (C4): unit lambertian sphere + one mesh from data/meshes, synthetic HDRI environment with a
    baked importance map."""
    b = SceneBuilder()
    add_library_curves(b, ["srgb_r", "srgb_g", "srgb_b", "flat_zero"])
    tex = synthetic_hdri(*hdri_size)
    ts = b.texstack_texture4("synthetic_hdri", [b.curve("srgb_r"), b.curve("srgb_g"), b.curve("srgb_b"), b.curve("flat_zero")], tex)
    b.set_environment_hdr(ts, 1.0, importance=importance)
    b.env_sampling_probability = env_sampling_probability
    white = add_library_material(b, "lambertian_white")
    b.add_sphere(1.0, (0.0, 0.0, 0.0), white)
    if mesh:
        gold = add_library_material(b, "ggx_gold")
        p, f, n, mtl = _npz_mesh(mesh)
        m = b.add_mesh(p, f, n, face_materials=api.material_id(api.TAG_MATERIAL, 0))
        b.add_mesh_instance(m, gold, transform_from_data(scale=(0.6, 0.6, 0.6), rotate=[((0, 0, 1), -60.0)], translate=(-0.6, 1.6, -0.2)))
    b.add_camera((-5.0, 0.3, 0.4), (0.0, 0.4, -0.3), 24.0, focal_distance=5.0, aperture_diameter=0.001)
    return b


def hdri_small():
    """C4 at test size: gem mesh, 64x32 HDRI, 32x32 importance map."""
    return hdri_test(mesh="gem", hdri_size=(64, 32), importance=(32, 32))


def hdri_c4_small():
    """The C4 scene (monkey mesh, 4188 triangles) with a small HDRI / importance map, for parity tests."""
    return hdri_test(mesh="monkey", hdri_size=(128, 64), importance=(64, 64))


def test_prism(hdri_size=(1024, 512), importance=(1024, 1024)):
    """data/scenes/test_prism.toml of the reference tree: the rect room of the gem scene, a thin xenon SharpLight (Dual, cos^101), the 836-triangle prism.obj
    in dispersive glass under a transform stack (scale 0.9, rotate 90 degrees about z, translate), camera inside the room, an HDRI environment with
    env_sampling_probability 0.1 — its `low_res_hdri` file is absent from the tree, so the synthetic HDRI of C4 stands in.  Transforms AND
    lights AND environment sampling: the scene class that takes the GENERAL kernel forms (nothing the specialised forms lack is missing here)."""
    b = SceneBuilder()
    add_library_curves(b, ["srgb_r", "srgb_g", "srgb_b", "flat_zero"])
    ts = b.texstack_texture4("synthetic_hdri", [b.curve("srgb_r"), b.curve("srgb_g"), b.curve("srgb_b"), b.curve("flat_zero")], synthetic_hdri(*hdri_size))
    b.set_environment_hdr(ts, 1.0, importance=importance)
    b.env_sampling_probability = 0.1
    light = add_library_material(b, "sharp_light_xenon")
    white = add_library_material(b, "lambertian_white")
    red = add_library_material(b, "lambertian_red")
    green = add_library_material(b, "lambertian_green")
    glass = add_library_material(b, "ggx_glass_dispersive")
    b.add_rect((0.7, 0.01), (0.0, 0.0, 0.9), "Z", False, light)
    b.add_rect((2, 2), (0.0, 0.0, 1.0), "Z", True, white)
    b.add_rect((2, 2), (0.0, 0.0, -1.0), "Z", True, white)
    b.add_rect((2, 2), (0.0, 1.0, 0.0), "Y", True, red)
    b.add_rect((2, 2), (0.0, -1.0, 0.0), "Y", True, green)
    b.add_rect((2, 2), (1.0, 0.0, 0.0), "X", True, white)
    p, f, n, mtl = _npz_mesh("prism")
    mesh = b.add_mesh(p, f, n, face_materials=api.material_id(api.TAG_MATERIAL, 0))
    b.add_mesh_instance(mesh, glass, transform_from_data(scale=(0.9, 0.9, 0.9), rotate=[((0, 0, 1), 90.0)], translate=(0.0, 0.0, -0.1)))
    b.add_camera((0.5, 0.0, 0.0), (0.0, 0.0, 0.0), 70.4, focal_distance=0.5, aperture_diameter=0.001)
    return b


def test_prism_small():
    """test_prism with a small HDRI / importance map, for parity tests."""
    return test_prism(hdri_size=(64, 32), importance=(32, 32))


def test_bokeh(hdri_size=(1024, 512), importance=(1024, 1024), lights_per_row=41):
    """data/scenes/test_bokeh.toml of the reference tree (G2): two rows of 41 sphere lights of radius 0.01 (`diffuse_light`: blackbody 5000 K, Dual) at x = -0.5 / +0.5,
    y = -20 .. 20, z = 0, seen through a wide thin lens (aperture_diameter 0.1, focal_distance 5) under an HDRI of strength 0.1 with env_sampling_probability 0.5 —
    its `autumn_park_8k` file is absent from the tree, so the synthetic HDRI of C4 stands in.  82 instances > 64: no leaf-sweep table, so every ray takes
    the TOP-LEVEL BVH WALK (src/accelerator/lbvh.rs:172-213, mod.rs:86-178) and a light sample picks one of 82 lights (world/mod.rs:100-124).  The scene file's
    Bladed aperture is sampled as circular."""
    b = SceneBuilder()
    add_library_curves(b, ["srgb_r", "srgb_g", "srgb_b", "flat_zero"])
    ts = b.texstack_texture4("synthetic_hdri", [b.curve("srgb_r"), b.curve("srgb_g"), b.curve("srgb_b"), b.curve("flat_zero")], synthetic_hdri(*hdri_size))
    b.set_environment_hdr(ts, 0.1, importance=importance)
    b.env_sampling_probability = 0.5
    light = add_library_material(b, "diffuse_light")
    half = lights_per_row // 2
    for x in (-0.5, 0.5):                       # file order: the x = -0.5 row first, y ascending
        for k in range(-half, half + 1):
            b.add_sphere(0.01, (x, float(k), 0.0), light)
    b.add_camera((0.0, -5.0, 0.5), (0.0, 0.0, 0.2), 45.0, focal_distance=5.0, aperture_diameter=0.1)
    return b


def test_bokeh_small():
    """test_bokeh with a small HDRI / importance map, for parity tests (still 82 instances: the top-level walk)."""
    return test_bokeh(hdri_size=(64, 32), importance=(32, 32))


def test_bokeh_floor(hdri_size=(1024, 512), importance=(1024, 1024), lights_per_row=41):
    """NOT a reference scene: test_bokeh with a white Lambertian floor 0.05 under the lights and three non-emissive spheres on it.  In test_bokeh itself every
    surface is a light — a light vertex takes no light samples (pt.rs:512-561), so its renders never trace a light-sample ray; with the floor the 82-entry
    light list is sampled at every floor vertex and the light-sample rays take the top-level walk too (G2F of DESIGN section 6)."""
    b = test_bokeh(hdri_size, importance, lights_per_row)
    cam = b.cameras.pop()
    white = add_library_material(b, "lambertian_white")
    glass = add_library_material(b, "ggx_glass_rough")
    gold = add_library_material(b, "ggx_gold")
    b.add_rect((60.0, 60.0), (0.0, 0.0, -0.05), "Z", True, white)
    b.add_sphere(0.15, (0.0, -1.0, 0.1), glass)
    b.add_sphere(0.12, (-0.3, 0.5, 0.07), gold)
    b.add_sphere(0.10, (0.35, 1.5, 0.05), white)
    b.cameras.append(cam)
    return b


def test_bokeh_floor_58(hdri_size=(1024, 512), importance=(1024, 1024)):
    """NOT a reference scene, a measurement: test_bokeh_floor with 29 lights per row — 62 instances, so the scene HAS a leaf-sweep table; with PT_AMD_NO_SWEEP=1 it takes the
    top-level walk as G2F does: the two forms on one scene (profiles/r6_experiments.md section 10)."""
    return test_bokeh_floor(hdri_size, importance, lights_per_row=29)


def rect_room(spheres=28):
    """NOT a reference scene, a measurement: a closed room of six rectangles (1 x 1 x 1, Lambertian) with a rectangular lamp under its ceiling and `spheres` Lambertian and
    GGX spheres of radius 0.04 on a grid inside — a DENSE scene of analytic instances (every ray hits something), against test_bokeh_floor's sparse one: which form of the
    closest-hit and light-sample kernels such a scene wants (profiles/r6_experiments.md section 10)."""
    b = SceneBuilder()
    add_library_curves(b, ["flat_zero"])
    b.set_environment_constant(b.curve("flat_zero"), 0.0)
    b.env_sampling_probability = 0.0
    light = add_library_material(b, "diffuse_light_cornell")
    white, red, green = (add_library_material(b, n) for n in ("lambertian_white", "lambertian_red", "lambertian_green"))
    gold = add_library_material(b, "ggx_gold")
    b.add_rect((0.3, 0.3), (0.5, 0.5, 0.999), "Z", False, light)
    b.add_rect((1.0, 1.0), (0.5, 0.5, 0.0), "Z", True, white); b.add_rect((1.0, 1.0), (0.5, 0.5, 1.0), "Z", True, white)
    b.add_rect((1.0, 1.0), (0.0, 0.5, 0.5), "X", True, white); b.add_rect((1.0, 1.0), (1.0, 0.5, 0.5), "X", True, white)
    b.add_rect((1.0, 1.0), (0.5, 0.0, 0.5), "Y", True, red); b.add_rect((1.0, 1.0), (0.5, 1.0, 0.5), "Y", True, green)
    for k in range(spheres):
        i, j, l = k % 4, (k // 4) % 4, k // 16
        b.add_sphere(0.04, (0.35 + 0.15 * i, 0.2 + 0.2 * j, 0.15 + 0.3 * l), gold if k % 3 == 0 else white)
    b.add_camera((0.02, 0.5, 0.5), (1.0, 0.5, 0.45), 60.0, focal_distance=0.6, aperture_diameter=0.005)
    return b


def _bokeh_floor_rows(n):
    """(measurement scenes: test_bokeh_floor with n lights per row, 2 n + 4 instances — where the leaf sweep stops paying, profiles/r6_experiments.md section 10)"""
    return lambda hdri_size=(1024, 512), importance=(1024, 1024): test_bokeh_floor(hdri_size, importance, lights_per_row=n)


def test_bokeh_floor_gem(hdri_size=(1024, 512), importance=(1024, 1024)):
    """NOT a reference scene: test_bokeh_floor with the brilliant-cut gem (302 triangles, moissanite) standing on the floor — more than 64 instances AND a mesh: the top-level
    walk parks at the mesh, the parked kernels walk it in full waves (top_walk_run / top_walk_resume), and the wave's last top-level walkers are evicted."""
    b = test_bokeh_floor(hdri_size, importance)
    cam = b.cameras.pop()
    gem = add_library_material(b, "ggx_moissanite")
    p, f, n, mtl = _npz_mesh("gem")
    m = b.add_mesh(p, f, n, face_materials=api.material_id(api.TAG_MATERIAL, 0))
    b.add_mesh_instance(m, gem, transform_from_data(scale=(0.5, 0.5, 0.5), translate=(0.1, -2.0, 0.2)))
    b.cameras.append(cam)
    return b


def test_bokeh_floor_gem_small():
    return test_bokeh_floor_gem(hdri_size=(64, 32), importance=(32, 32))


def test_bokeh_floor_small():
    return test_bokeh_floor(hdri_size=(64, 32), importance=(32, 32))


def hdri_emissive_mesh():
    """An HDRI environment and a mesh whose *instance* overrides its material with a light (not a reference scene): the mesh is not in the
    light list (world/mod.rs:45-54 looks at analytic instances' ids and mesh face ids only), so the list is empty — yet its hits carry the
    Light tag, emit, and take no light samples (pt.rs:512-561).  The kernel forms "without lights" must not be chosen for it."""
    b = hdri_test(mesh=None, hdri_size=(64, 32), importance=(32, 32))
    lamp = add_library_material(b, "diffuse_light_flat_x5")
    p, f, n, mtl = _npz_mesh("gem")
    m = b.add_mesh(p, f, n, face_materials=api.material_id(api.TAG_MATERIAL, 0))
    b.add_mesh_instance(m, lamp, transform_from_data(scale=(0.5, 0.5, 0.5), translate=(-0.6, 1.5, -0.1)))
    return b



# =================================================================== the reference tree's self-contained scene files
# data/scenes/*.toml of the reference that need nothing the tree does not hold (no OBJ, no HDRI): the same instances, materials,
# environment and camera, value for value (tests/test_reference_fixtures.py renders the reference's own file and the builder on the
# oracle and compares the films bit for bit where /root/reference exists; the GPU tier renders the builders on the engine).
def _room(b, light=None, front=False):
    """The 2 x 2 x 2 room of the reference's cornell_box_* / test_blackbox scene files: two-sided rects, white ceiling, floor and back,
    red at y = +1, green at y = -1 (and a white wall behind the camera when `front`)."""
    white = add_library_material(b, "lambertian_white")
    red = add_library_material(b, "lambertian_red")
    green = add_library_material(b, "lambertian_green")
    b.add_rect((2, 2), (0.0, 0.0, 1.0), "Z", True, white)
    b.add_rect((2, 2), (0.0, 0.0, -1.0), "Z", True, white)
    b.add_rect((2, 2), (0.0, 1.0, 0.0), "Y", True, red)
    b.add_rect((2, 2), (0.0, -1.0, 0.0), "Y", True, green)
    b.add_rect((2, 2), (1.0, 0.0, 0.0), "X", True, white)
    if front:
        b.add_rect((2, 2), (-1.0, 0.0, 0.0), "X", True, white)


def _dark_sun(b):
    """environment = Sun of strength 0 towards +z, colour D65, never sampled (test_lighting_north, test_nee_sphere, test_sampling_methods)."""
    add_library_curves(b, ["D65"])
    b.set_environment_sun(b.curve("D65"), 0.0, 0.0565, (0.0, 0.0, 1.0))
    b.env_sampling_probability = 0.0


def ref_candela_calibration():
    """data/scenes/candela_calibration.toml: a unit sphere that emits the 540 THz spike, seen from x = -5 (rendered by
    data/config_test_candela_calibration.toml with wavelength_bounds [555, 560] and only_direct)."""
    b = SceneBuilder()
    add_library_curves(b, ["D65"])
    b.set_environment_constant(b.curve("D65"), 0.0)
    b.env_sampling_probability = 0.0
    b.add_sphere(1.0, (0.0, 0.0, 0.0), add_library_material(b, "diffuse_540THz"))
    b.add_camera((-5.0, 0.0, 0.0), (0.0, 0.0, 0.0), 27.8, focal_distance=5.0, aperture_diameter=0.02)
    return b


def ref_cornell_box_parallel_prism():
    """data/scenes/cornell_box_parallel_prism.toml (its prism instance is commented out in the file): the room under a Sun environment
    sampled with probability 0.5 and a narrow two-sided sharp light under the ceiling."""
    b = SceneBuilder()
    add_library_curves(b, ["D65"])
    b.set_environment_sun(b.curve("D65"), 0.4, 0.0565, (1.0, 0.0, 1.0))
    b.env_sampling_probability = 0.5
    b.add_rect((0.4, 0.1), (0.0, 0.8, 0.9), "Z", True, add_library_material(b, "sharp_light"))
    _room(b)
    b.add_camera((-5.0, 0.0, 0.0), (0.0, 0.0, 0.0), 27.8, focal_distance=5.0, aperture_diameter=0.02)
    return b


def ref_cornell_box_single_orb_caustic():
    """data/scenes/cornell_box_single_orb_caustic.toml: the room, a one-sided sharp light (cos^401) and a dispersive glass ball."""
    b = SceneBuilder()
    add_library_curves(b, ["D65"])
    b.set_environment_constant(b.curve("D65"), 0.0)
    b.env_sampling_probability = 0.0
    b.add_rect((0.2, 0.2), (0.0, 0.0, 0.9), "Z", False, add_library_material(b, "sharp_light"))
    _room(b)
    b.add_sphere(0.3, (0.1, 0.1, -0.15), add_library_material(b, "ggx_glass_dispersive"))
    b.add_camera((-5.0, 0.0, 0.0), (0.0, 0.0, 0.0), 27.8, focal_distance=5.0, aperture_diameter=0.02)
    return b


def ref_test_blackbox():
    """data/scenes/test_blackbox.toml: the closed room seen from inside, no light in it, a D65 sky of strength 1 outside and
    env_sampling_probability 1 — every light sample is an environment sample, and every one of them is blocked."""
    b = SceneBuilder()
    add_library_curves(b, ["D65"])
    b.set_environment_constant(b.curve("D65"), 1.0)
    b.env_sampling_probability = 1.0
    _room(b, front=True)
    b.add_camera((0.5, 0.0, 0.0), (0.0, 0.0, 0.0), 70.4, focal_distance=0.5, aperture_diameter=0.001)
    return b


def ref_test_lighting_north():
    """data/scenes/test_lighting_north.toml: a lamp at z = 10 over a white unit sphere — with the Cornell box's camera, which sits INSIDE the sphere."""
    b = SceneBuilder()
    _dark_sun(b)
    b.add_rect((0.5, 0.5), (0.0, 0.0, 10.0), "Z", True, add_library_material(b, "diffuse_light_flat_x10"))
    b.add_sphere(1.0, (0.0, 0.0, 0.0), add_library_material(b, "lambertian_white"))
    b.add_camera((-0.8, 0.278, 0.273), (0.0, 0.278, 0.273), 37.8, focal_distance=1.1, aperture_diameter=0.01)
    return b


def ref_test_nee_sphere():
    """data/scenes/test_nee_sphere.toml: a sphere LIGHT of radius 5 (Sphere::sample / psa_pdf, src/geometry/sphere.rs:95-152) over six
    Lambertian balls, a floor and a wall."""
    b = SceneBuilder()
    _dark_sun(b)
    b.add_sphere(5.0, (3.0, -6.0, 8.0), add_library_material(b, "diffuse_light_flat_x10"))
    white = add_library_material(b, "lambertian_white")
    for y in (-2.3, 0.0, 2.3):
        b.add_sphere(1.0, (5.0, y, 0.0), white)
    for y, m in ((-1.4, "lambertian_red"), (0.0, "lambertian_green"), (1.4, "lambertian_blue")):
        b.add_sphere(0.4, (3.5, y, -0.6), add_library_material(b, m))
    b.add_rect((40, 40), (0.0, 0.0, -1.0), "Z", True, white)
    b.add_rect((20, 20), (16.0, 0.0, 0.0), "X", True, white)
    b.add_camera((0.5, 0.0, 0.0), (0.0, 0.0, 0.0), 70.4, focal_distance=0.5, aperture_diameter=0.001)
    return b


def ref_test_rtiow_scene_2():
    """data/scenes/test_rtiow_scene_2.toml: a hollow glass ball (ggx_glass outside, ggx_air_glass inside at radius 0.48), a blue and a
    gold ball on a yellow floor under a sphere light and a sky-blue constant environment that is never sampled (probability 0)."""
    b = SceneBuilder()
    add_library_curves(b, ["simple_sky_blue"])
    b.set_environment_constant(b.curve("simple_sky_blue"), 1.0)
    b.env_sampling_probability = 0.0
    b.add_sphere(5.0, (3.0, -6.0, 8.0), add_library_material(b, "diffuse_light_flat_x10"))
    b.add_sphere(0.5, (3.5, -1.0, 0.0), add_library_material(b, "ggx_glass"))
    b.add_sphere(0.48, (3.5, -1.0, 0.0), add_library_material(b, "ggx_air_glass"))
    b.add_sphere(0.5, (3.5, 0.0, 0.0), add_library_material(b, "lambertian_blue"))
    b.add_sphere(0.5, (3.5, 1.0, 0.0), add_library_material(b, "ggx_gold"))
    b.add_rect((40, 40), (0.0, 0.0, -0.5), "Z", True, add_library_material(b, "lambertian_yellow"))
    b.add_camera((0.5, 0.0, 0.0), (0.0, 0.0, 0.0), 70.4, focal_distance=0.5, aperture_diameter=0.001)
    return b


def ref_test_sampling_methods():
    """data/scenes/test_sampling_methods.toml: two rect lamps of different size either side of a white ball on a floor (two entries in the light list)."""
    b = SceneBuilder()
    _dark_sun(b)
    lamp = add_library_material(b, "diffuse_light_flat_x10")
    b.add_rect((1, 1), (0.0, 2.0, 0.5), "Y", True, lamp)
    b.add_rect((0.5, 0.5), (0.0, -1.0, 0.5), "Y", True, lamp)
    white = add_library_material(b, "lambertian_white")
    b.add_sphere(0.5, (0.0, 0.0, 0.5), white)
    b.add_rect((10, 10), (0.0, 0.0, 0.0), "Z", True, white)
    b.add_camera((0.5, 0.0, 0.0), (0.0, 0.0, 0.0), 70.4, focal_distance=0.5, aperture_diameter=0.001)
    return b


def ref_sun_test():
    """data/scenes/sun_test.toml of the REFERENCE tree (this repository's own data/scenes/sun_test.toml is `sun_test` above): no light but
    the Sun environment (strength 0.4, D65, towards (1, 0, 1)), sampled with probability 0.5; eight balls — dispersive glass, gold, copper,
    red — on a floor."""
    b = SceneBuilder()
    add_library_curves(b, ["D65"])
    b.set_environment_sun(b.curve("D65"), 0.4, 0.0565, (1.0, 0.0, 1.0))
    b.env_sampling_probability = 0.5
    b.add_rect((20, 20), (0.0, 10.0, -1.0), "Z", True, add_library_material(b, "lambertian_white"))
    for side in (1.0, -1.0):
        for x, y, m in ((-0.6, 0.3, "ggx_glass_dispersive"), (-0.6, 0.8, "ggx_gold"), (0.6, 0.3, "ggx_copper"), (0.6, 0.8, "lambertian_red")):
            b.add_sphere(0.2, (x, side * y, -0.8), add_library_material(b, m))
    b.add_camera((-5.0, 0.0, 0.4), (0.0, 0.0, -0.7), 16.2, focal_distance=5.0, aperture_diameter=0.001)
    return b


REFERENCE_TREE_SCENES = {"candela_calibration": ref_candela_calibration, "cornell_box_parallel_prism": ref_cornell_box_parallel_prism,
                         "cornell_box_single_orb_caustic": ref_cornell_box_single_orb_caustic, "sun_test": ref_sun_test, "test_blackbox": ref_test_blackbox,
                         "test_lighting_north": ref_test_lighting_north, "test_nee_sphere": ref_test_nee_sphere, "test_rtiow_scene_2": ref_test_rtiow_scene_2,
                         "test_sampling_methods": ref_test_sampling_methods}   # file name in the reference's data/scenes -> builder


SCENES = {"rect_room": rect_room, "rect_room_12": lambda: rect_room(12), "rect_room_44": lambda: rect_room(44), "test_bokeh_floor_10": _bokeh_floor_rows(5), "test_bokeh_floor_18": _bokeh_floor_rows(9), "test_bokeh_floor_26": _bokeh_floor_rows(13), "test_bokeh_floor_42": _bokeh_floor_rows(21),
          "test_bokeh_floor_58": test_bokeh_floor_58, "test_bokeh_floor_gem": test_bokeh_floor_gem, "test_bokeh_floor_gem_small": test_bokeh_floor_gem_small, "test_bokeh_floor": test_bokeh_floor, "test_bokeh_floor_small": test_bokeh_floor_small, "test_bokeh": test_bokeh, "test_bokeh_small": test_bokeh_small, "test_prism": test_prism, "test_prism_small": test_prism_small, "hdri_emissive_mesh": hdri_emissive_mesh, "hdri_test": hdri_test, "hdri_small": hdri_small, "hdri_c4_small": hdri_c4_small, "cornell_box": cornell_box, "cornell_gem": cornell_gem, "white_furnace": white_furnace,
          "mixed_primitives": mixed_primitives, "mixed_small": mixed_small, "sun_test": sun_test, "panorama_test": panorama_test, "empty_env": empty_env,
          "big_sphere_light": big_sphere_light, "disk_lamp": disk_lamp, "fog_ball": fog_ball}
SCENES.update({"ref_" + name: fn for name, fn in REFERENCE_TREE_SCENES.items()})
