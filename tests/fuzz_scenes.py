"""Seeded random scenes for differential testing (engine / emulation vs oracle): every primitive kind, random transform
stacks (non-uniform scale, rotation lists), one-sided and two-sided surfaces, lights of both kinds, metals and dielectrics,
small meshes with and without shading normals, every environment kind.  Not reference scenes: their only purpose is to
reach corners of the traversal and shading code that the authored scenes do not."""
import importlib

import numpy as np


def random_scene(seed):
    pkg = importlib.import_module("rust-pathtracer_amd")
    S, api = pkg.scene, pkg.api
    rng = np.random.default_rng(seed)
    b = S.SceneBuilder()
    env = rng.integers(0, 3)
    if env == 0:
        S.add_library_curves(b, ["simple_sky_blue"])
        b.set_environment_constant(b.curve("simple_sky_blue"), float(rng.uniform(0.0, 1.0)))
    elif env == 1:
        sun = b.curve_blackbody(None, float(rng.uniform(3000, 7000)), 1.0)
        d = rng.normal(size=3); d[2] = abs(d[2]) + 0.2
        b.set_environment_sun(sun, float(rng.uniform(0.5, 3.0)), float(rng.uniform(0.05, 0.3)), d.tolist())
    else:
        S.add_library_curves(b, ["srgb_r", "srgb_g", "srgb_b", "flat_zero"])
        ts = b.texstack_texture4("hdri", [b.curve(n) for n in ("srgb_r", "srgb_g", "srgb_b", "flat_zero")], S.synthetic_hdri(32, 16))
        rot = [((0, 0, 1), float(rng.uniform(-180, 180)))] if rng.random() < 0.5 else None
        imp = (16, 16) if rng.random() < 0.7 else (0, 0)
        b.set_environment_hdr(ts, float(rng.uniform(0.3, 1.5)), rotate=rot, importance=imp)
    b.env_sampling_probability = float(rng.choice([0.0, 0.3, 0.5, 1.0])) if env != 0 or rng.random() < 0.8 else 0.5
    mats = [S.add_library_material(b, n) for n in ("lambertian_white", "lambertian_red", "lambertian_green", "ggx_gold", "ggx_copper", "ggx_glass", "ggx_glass_rough",
                                                   "ggx_moissanite")]
    lights = [S.add_library_material(b, n) for n in ("diffuse_light_flat_x5", "diffuse_light_cornell", "sharp_light", "sharp_light_fluorescent")]

    def transform():
        if rng.random() < 0.4:
            return None
        scale = rng.uniform(0.4, 1.6, 3).tolist() if rng.random() < 0.6 else None
        rot = [(rng.normal(size=3).tolist(), float(rng.uniform(-180, 180))) for _ in range(rng.integers(0, 3))] or None
        tr = rng.uniform(-1.0, 1.0, 3).tolist() if rng.random() < 0.8 else None
        if scale is None and rot is None and tr is None:
            return None
        return S.transform_from_data(scale, rot, tr)

    def material(light_ok=True):
        if light_ok and rng.random() < 0.25:
            return lights[rng.integers(len(lights))]
        return mats[rng.integers(len(mats))]

    b.add_rect((8, 8), (0.0, 0.0, -1.5), "Z", True, mats[0])  # a floor, so that most paths bounce
    n_light = 0
    for _ in range(rng.integers(2, 9)):
        kind = rng.integers(0, 4)
        if 190000 <= seed < 200000 and kind == 3: kind = 1   # (the one-glass-body class below: no other mesh that the host could certify)
        origin = rng.uniform(-1.2, 1.2, 3).tolist()
        m = material()
        n_light += (m >> 16) == 1
        if kind == 0:
            b.add_rect(tuple(rng.uniform(0.3, 2.0, 2).tolist()), origin, "XYZ"[rng.integers(3)], bool(rng.integers(2)), m, transform())
        elif kind == 1:
            b.add_sphere(float(rng.uniform(0.2, 0.8)), origin, m, transform())
        elif kind == 2:
            b.add_disk(float(rng.uniform(0.2, 0.9)), origin, bool(rng.integers(2)), m, transform())
        else:
            p, f, n = S._octahedron()
            if rng.random() < 0.5:
                n = None
            p = p * rng.uniform(0.3, 0.9, 3).astype(np.float32)
            fm = np.array([mats[rng.integers(len(mats))] for _ in range(len(f))], np.uint32)   # mesh faces cannot be lights (mesh.rs:213-232)
            mesh = b.add_mesh(p, f, n, face_materials=fm)
            b.add_mesh_instance(mesh, material(light_ok=False) if rng.random() < 0.5 else None, transform())
    if n_light == 0:
        b.add_rect((1.0, 1.0), (0.0, 0.0, 2.0), "Z", True, lights[0])
    if rng.random() < 0.12:  # more than 64 instances: no sweep table, the two-level BVH walk is the default form
        for _ in range(70):
            o3 = rng.uniform(-2.5, 2.5, 3).tolist()
            if rng.random() < 0.7:
                b.add_sphere(float(rng.uniform(0.05, 0.25)), o3, material(light_ok=False), transform() if rng.random() < 0.2 else None)
            else:
                b.add_rect(tuple(rng.uniform(0.1, 0.5, 2).tolist()), o3, "XYZ"[rng.integers(3)], True, material(light_ok=False))
    if seed >= 600000:  # (a seed space of its own, round 5) the scene class of data/scenes/test_bokeh.toml: MANY small lights — more than pt_tuning::light_prepass_max, so that
        # light-sample rays are plain closest-hit searches — in rows a ray can graze (long, uneven top-level walks: the eviction of a wave's last walkers), more than 64 instances
        rng4 = np.random.default_rng(seed + 1357)
        if rng4.random() < 0.6:
            rows = int(rng4.integers(1, 4)); per = int(rng4.integers(16, 46))
            z0 = float(rng4.uniform(-1.3, 0.5))
            for r in range(rows):
                x0 = float(rng4.uniform(-1.0, 1.0)); radius = float(rng4.choice([0.01, 0.03, 0.08]))
                for k in range(per):
                    lit = rng4.random() < 0.8
                    b.add_sphere(radius, (x0, -4.0 + 8.0 * k / per, z0), lights[int(rng4.integers(len(lights)))] if lit else mats[int(rng4.integers(len(mats)))])
    if rng.random() < 0.3:  # a bigger mesh: the sweep table can no longer inline every triangle (walked mesh, parked rays)
        p, f, n, _ = S._npz_mesh("gem")
        mesh = b.add_mesh(p, f, n, face_materials=api.material_id(api.TAG_MATERIAL, mats[0] & 0xFFFF))
        b.add_mesh_instance(mesh, mats[rng.integers(len(mats))], S.transform_from_data((0.5, 0.5, 0.5), None, rng.uniform(-0.8, 0.8, 3).tolist()))
    if seed >= 100000:  # (a seed space of its own, so that the scenes of the seeds already in the tests stay what they are)
        # several different meshes too big for the sweep table together: a wave resumes parked rays of more than one of them
        rng2 = np.random.default_rng(seed + 7777)
        p, f, n, _ = S._npz_mesh("gem")
        for k in range(int(rng2.integers(1, 4))):
            scale = rng2.uniform(0.3, 0.7, 3)
            mesh = b.add_mesh((p * scale.astype(np.float32)).astype(np.float32), f, n if k % 2 == 0 else None,
                              face_materials=api.material_id(api.TAG_MATERIAL, mats[k % len(mats)] & 0xFFFF))
            b.add_mesh_instance(mesh, mats[int(rng2.integers(len(mats)))] if rng2.random() < 0.5 else None,
                                S.transform_from_data(None, None, rng2.uniform(-1.0, 1.0, 3).tolist()) if rng2.random() < 0.7 else None)
    if 140000 <= seed < 200000:  # (a seed space of its own, round 6; below 200000: rendered by the plain walk) closed CONVEX bodies of every kind the host certifies or refuses
        # (140000 .. 149999: the same under MIRRORING transforms — scales of either sign: the winding seen from outside is reversed, the normals are not)
        # (pt_blob.h PT_INST_CONVEX_*): the brilliant cut (flat-shaded, sharp edges), the prism (smooth-shaded: vertex normals up to 47 degrees off their faces), cubes, octahedra
        # with and without vertex normals — under random transform stacks (uneven scales, rotations), in glass, metal and Lambertian, several instances of one mesh, bodies that
        # overlap, lights of every kind next to, above and INSIDE their boxes, skies that light samples pick
        rng5 = np.random.default_rng(seed + 2468)
        cube_p = np.array([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)], np.float32) - 0.5
        cube_f = np.array([(0, 2, 1), (0, 3, 2), (4, 5, 6), (4, 6, 7), (0, 1, 5), (0, 5, 4), (1, 2, 6), (1, 6, 5), (2, 3, 7), (2, 7, 6), (3, 0, 4), (3, 4, 7)], np.uint32)
        shared = {}
        # 190000 ..: ONE body, of glass — the scene's only certified instance: the path segments refracted into it carry PT_PATH_INSIDE_MARK and its sweep ends at the first
        # interior acceptance (pt_device.h mesh_walk `inside`); the floor, the lamps and the other shapes may run through it
        solo = seed >= 190000
        for k in range(1 if solo else int(rng5.integers(1, 5))):
            kind = int(rng5.integers(0, 5))
            if kind == 0: pm, fm_, nm, _ = S._npz_mesh("brilliant_diamond"); size = 0.5
            elif kind == 1: pm, fm_, nm, _ = S._npz_mesh("prism"); size = 2.5
            elif kind == 2: pm, fm_, nm = cube_p, cube_f, None; size = 0.8
            elif kind == 3: pm, fm_, nm = S._octahedron(); size = 0.7
            else: pm, fm_, nm = S._octahedron(); nm = None; size = 0.7
            if kind in shared and rng5.random() < 0.5: mesh = shared[kind]            # a second instance of a mesh already there
            else:
                mesh = b.add_mesh(pm, fm_, nm, face_materials=api.material_id(api.TAG_MATERIAL, mats[int(rng5.integers(len(mats)))] & 0xFFFF))
                shared[kind] = mesh
            scale = (rng5.uniform(0.5, 1.5, 3) * size).tolist() if rng5.random() < 0.6 else (size, size, size)
            if seed < 150000: scale = [float(v) * float(rng5.choice([-1.0, 1.0])) for v in scale]
            rot = [(rng5.normal(size=3).tolist(), float(rng5.uniform(-180, 180))) for _ in range(int(rng5.integers(0, 3)))] or None
            at = rng5.uniform(-1.0, 1.0, 3)
            b.add_mesh_instance(mesh, mats[5 + int(rng5.integers(3))] if solo else (mats[int(rng5.integers(len(mats)))] if rng5.random() < 0.7 else None), S.transform_from_data(scale, rot, at.tolist()))
            if rng5.random() < 0.4:   # a small lamp next to, above, or inside the body's box
                off = rng5.normal(size=3) * float(rng5.choice([0.2, 0.6, 1.5]))
                lm = lights[int(rng5.integers(len(lights)))]
                if rng5.random() < 0.5: b.add_sphere(float(rng5.uniform(0.03, 0.15)), (at + off).tolist(), lm)
                else: b.add_rect((float(rng5.uniform(0.1, 0.4)), float(rng5.uniform(0.1, 0.4))), (at + off).tolist(), "XYZ"[int(rng5.integers(3))], True, lm)
    if seed >= 200000:  # (its own seed space again) participating media for the medium-aware walk: HG and Rayleigh mediums behind passthrough
        # boundaries and inside glass, nested and overlapping, so that the list of tracked mediums grows, shrinks and overflows
        rng3 = np.random.default_rng(seed + 4242)
        ids = []
        for k in range(int(rng3.integers(1, 6))):
            if rng3.random() < 0.7:
                ids.append(b.medium_hg("hg%d" % k, b.curve_flat(None, float(rng3.uniform(0.2, 1.9))), b.curve_flat(None, float(rng3.choice([0.0, 0.1, 0.6]))),
                                       b.curve_flat(None, float(rng3.uniform(0.05, 3.0)))))
            else:
                ids.append(b.medium_rayleigh("ray%d" % k, b.curve_cauchy(None, float(rng3.uniform(1.1, 1.8)), float(rng3.uniform(0.0, 5000.0))), float(rng3.uniform(0.05, 2.0))))
        tint = b.curve_flat(None, float(rng3.uniform(0.5, 1.0)))
        eta = b.curve_cauchy(None, 1.45, 3540.0); one = b.curve_flat(None, 1.0); zero = b.curve_flat(None, 0.0)
        for k in range(int(rng3.integers(2, 7))):
            inner = int(rng3.choice(ids)); outer = int(rng3.choice([0, 0, int(rng3.choice(ids))]))
            if rng3.random() < 0.6:
                m = b.material_passthrough("pass%d" % k, tint, outer, inner)
            else:
                m = b.material_ggx("glass%d" % k, float(rng3.choice([0.01, 0.1, 0.4])), eta, one, zero, outer, inner)
            o3 = rng3.uniform(-1.0, 1.0, 3).tolist()
            if rng3.random() < 0.7:
                b.add_sphere(float(rng3.uniform(0.3, 1.2)), o3, m, transform() if rng3.random() < 0.3 else None)
            else:
                p2, f2, n2 = S._octahedron()
                mesh = b.add_mesh((p2 * rng3.uniform(0.5, 1.2, 3).astype(np.float32)).astype(np.float32), f2, None, face_materials=api.material_id(api.TAG_MATERIAL, 0))
                b.add_mesh_instance(mesh, m, S.transform_from_data(None, None, o3))
    eye = rng.normal(size=3); eye = eye / np.linalg.norm(eye) * rng.uniform(3.0, 6.0); eye[2] = abs(eye[2]) * 0.5
    if rng.random() < 0.2:
        b.add_panorama_camera(eye.tolist(), (0.0, 0.0, 0.0), (float(rng.uniform(60, 360)), float(rng.uniform(40, 180))))
    else:
        b.add_camera(eye.tolist(), (0.0, 0.0, -0.3), float(rng.uniform(25, 60)), focal_distance=float(np.linalg.norm(eye)), aperture_diameter=float(rng.choice([0.0, 0.01, 0.1])))
    return b


def medium_aware(seed):
    """The seeds whose scenes hold media are rendered with the medium-aware walk."""
    return seed >= 200000


def random_rays(seed, n):
    rng = np.random.default_rng(seed + 1000)
    o = rng.normal(0, 1.5, (n, 3)).astype(np.float32)
    d = rng.normal(0, 1, (n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    # a share of axis-parallel rays and of rays that start on z = -1.5 (the floor plane)
    d[: n // 16] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, n // 16)] * rng.choice([-1.0, 1.0], (n // 16, 1)).astype(np.float32)
    o[n // 16: n // 8, 2] = -1.5
    return o, d
