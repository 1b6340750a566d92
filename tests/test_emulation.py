"""CPU: the HIP engine's lane logic + host planning (emulated sequentially, tests/host_emulation/ptemu.cpp)
against the oracle.  This is what proves the wavefront decomposition before the GPU is involved."""
import os
import subprocess

import numpy as np
import pytest

import parity_suite as ps

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "host_emulation")
CSRC = os.path.join(HERE, "..", "rust-pathtracer_amd", "csrc")


@pytest.fixture(scope="session")
def emu(pkg):
    lib = os.path.join(EMU_DIR, "libptemu.so")
    srcs = [os.path.join(EMU_DIR, "ptemu.cpp"), os.path.join(CSRC, "pt_scene_host.cpp"), os.path.join(CSRC, "pt_plan.cpp")]
    deps = srcs + [os.path.join(CSRC, h) for h in ("pt_device.h", "pt_stages.h", "pt_blob.h", "pt_plan.h", "pt_scene_host.h")] + \
        [os.path.join(HERE, "..", "include", h) for h in ("pt_api.h", "pt_numerics.h")]
    if not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
                               "-Wno-unused-function", "-o", lib] + srcs)
    return pkg.api.Library(lib, "ptemu_", optional=("render_device", "device_info"))


@pytest.mark.parametrize("scene", ["cornell_box", "cornell_gem", "mixed_primitives", "mixed_small", "white_furnace", "hdri_small", "hdri_c4_small"])
def test_closest_hits_bit_exact(emu, oracle, scene):
    ps.intersect_parity(emu, oracle, scene)


@pytest.mark.parametrize("scene", ["cornell_box", "cornell_gem", "panorama_test", "hdri_small"])
def test_camera_samples_bit_exact(emu, oracle, scene):
    ps.camera_parity(emu, oracle, scene, seed=3, wavelength=(400.0, 700.0))


@pytest.mark.parametrize("scene", ["cornell_box", "cornell_gem", "mixed_primitives"])
def test_materials_bit_exact(emu, oracle, scene):
    ps.material_parity(emu, oracle, scene)


def test_tabulated_curves_through_their_cell_tables(emu, oracle):
    ps.curve_table_parity(emu, oracle)


@pytest.mark.parametrize("scene,w,h,spp,mb,kw", [
    ("cornell_box", 48, 40, 13, 4, {}),                       # C1 shape at reduced size; spp not a multiple of 10
    ("cornell_box", 33, 21, 4, 8, {"tile": (16, 16)}),        # remnant tiles (tiled.rs:236-277)
    ("cornell_box", 24, 24, 5, 6, {"only_direct": True}),
    ("cornell_box", 24, 24, 5, 6, {"light_samples": 0}),
    ("cornell_box", 24, 24, 5, 3, {"min_bounces": 4}),        # roulette never active
    ("cornell_gem", 40, 24, 6, 12, {}),                       # C3 shape: dispersive GGX + transformed mesh
    ("mixed_primitives", 32, 32, 8, 6, {"light_samples": 3, "seed": 5}),
    ("mixed_small", 32, 32, 8, 6, {"light_samples": 3, "seed": 6}),
    ("panorama_test", 48, 24, 6, 5, {}),                      # PanoramaCamera
    ("cornell_box", 24, 20, 23, 4, {"phase_samples": 23, "tile": (24, 20)}),   # NaiveRenderer: one sum over all samples (naive.rs:82-103)
    ("empty_env", 1, 1, 1, 0, {}),                            # edge cases: no instances, 1x1 film, no bounces
    ("empty_env", 5, 3, 2, 4, {"light_samples": 3}),
    ("cornell_box", 3, 2, 1, 1, {"tile": (64, 64)}),          # film smaller than a tile
    ("hdri_c4_small", 24, 24, 4, 4, {"light_samples": 3}),    # monkey mesh: walked through its BVH (the triangles do not fit the sweep table)
    ("white_furnace", 16, 16, 12, 8, {"light_samples": 6}),
    ("cornell_box", 40, 36, 11, 6, {"hero_wavelengths": 4}),  # C5 shape: hero wavelength + 3 passengers
    ("mixed_primitives", 32, 32, 6, 6, {"hero_wavelengths": 4, "light_samples": 3}),
    ("mixed_small", 32, 32, 6, 6, {"hero_wavelengths": 4, "light_samples": 2}),
    ("hdri_small", 32, 32, 8, 4, {"light_samples": 6}),       # C4 shape: HDR environment, importance map, env NEE + MIS
    ("hdri_small", 24, 24, 4, 4, {"light_samples": 8}),       # the most light samples an item can hold
    ("test_bokeh_small", 48, 48, 6, 8, {"light_samples": 2}),      # G2: 82 instances, no sweep table — the top-level walk
    ("test_bokeh_floor_small", 48, 40, 6, 8, {"light_samples": 3}),
    ("test_bokeh_floor_gem_small", 48, 40, 6, 8, {"light_samples": 2}),   # ... and a mesh: parked at the mesh, evicted from the top-level walk
    ("test_prism_small", 40, 40, 6, 8, {"light_samples": 3}),     # the reference tree's test_prism.toml: transform stack + lights + environment sampling — the general kernel forms
    ("hdri_emissive_mesh", 32, 32, 6, 4, {"light_samples": 3}),   # empty light list, but a mesh instance overridden with a light material: its hits emit
    ("disk_lamp", 40, 28, 6, 5, {"light_samples": 2, "seed": 2}),   # one disk lamp: the lean form's light test at the vertex (the ceiling's rays start below the lamp), the list of live items
    ("fog_ball", 48, 32, 8, 8, {"medium_aware": True}),       # random_walk_medium — HG fog and Rayleigh haze behind passthrough boundaries, fog in glass
    ("fog_ball", 32, 24, 6, 12, {"medium_aware": True, "light_samples": 3, "min_bounces": 3, "seed": 9}),
    ("fog_ball", 24, 24, 5, 6, {"medium_aware": True, "light_samples": 0}),
    ("cornell_box", 32, 32, 6, 6, {"medium_aware": True}),    # no medium in the scene: the medium-aware walk's own vertex rules (a light vertex adds nothing)
    ("mixed_small", 32, 32, 6, 6, {"medium_aware": True, "light_samples": 3}),
    # the reference tree's self-contained scene files (pkg.scene.REFERENCE_TREE_SCENES; the GPU tier renders all nine, with and without hero wavelengths)
    ("ref_candela_calibration", 32, 32, 20, 12, {"wavelength": (555.0, 560.0), "only_direct": True, "light_samples": 1, "min_bounces": 2}),
    ("ref_cornell_box_single_orb_caustic", 40, 32, 8, 8, {}),
    ("ref_sun_test", 40, 32, 8, 8, {}),
    ("ref_test_blackbox", 32, 32, 6, 6, {}),
    ("ref_test_nee_sphere", 40, 32, 8, 8, {}),
    ("ref_test_rtiow_scene_2", 40, 32, 8, 8, {"hero_wavelengths": 4}),
    ("ref_test_sampling_methods", 40, 32, 8, 8, {}),
    ("cornell_box", 12, 12, 1024, 8, {}),                     # C2's sample count: 102 phases of ten and one of four per pixel (the GPU tier: 128 x 128)
])
def test_film_parity(emu, oracle, scene, w, h, spp, mb, kw):
    ps.render_parity(emu, oracle, scene, w, h, spp, mb, **kw)


def test_pass_planning_is_invisible(emu, oracle, monkeypatch):
    """Chunking pixels and samples into passes (tiny batch capacity) changes nothing."""
    monkeypatch.setenv("PTEMU_BATCH", "1024")
    m = ps.render_parity(emu, oracle, "cornell_box", 40, 36, 27, 4)
    monkeypatch.setenv("PTEMU_BATCH", "100000")
    ps.render_parity(emu, oracle, "cornell_box", 40, 36, 27, 4)


def test_shards_and_sample_ranges(emu):
    ps.shards_and_ranges(emu)


def test_golden_vectors(emu):
    for name in ps.GOLDEN_RENDERS:
        film, prof, ref, counters = ps.golden_render(emu, name)
        ps.check_film(film, ref, prof, counters)
    for scene in ("cornell_box", "mixed_primitives", "cornell_gem"):
        got, want = ps.golden_hits(emu, scene)
        ps.assert_hits_equal(got, want)
        ps.golden_materials(emu, scene)


@pytest.mark.parametrize("scene", ["cornell_box", "mixed_small", "white_furnace", "cornell_gem", "mixed_primitives", "hdri_small", "hdri_c4_small"])
def test_leaf_sweep_equals_bvh_walk(emu, pkg, monkeypatch, scene):
    """Scenes of <= 64 instances take world_hit_sweep (with the triangle leaves of small meshes in the table and the BVHs of
    big meshes walked from it); the pure BVH walk (flag 16), the exact slab test (2) and no culling (4) must give the same
    bits: hits, films and ray counters.  So must phase 3 as independent unbounded tests + ordered replay (128: the logic of the
    wave-pooled kernels; with 4, the masks are not culled by the bound either; in a table with walked meshes 128 is the parked kernels'
    protocol instead — park at the mesh, resume, leave the walk after every triangle test and go on from the cursor; 144 = 128 + 16: the same
    protocol over the top-level tree, for scenes without a sweep table)."""
    b = pkg.scene.SCENES[scene]()
    o, d = ps.golden_rays(scene, 4096, 33)
    # axis-parallel directions take the undecided path for every box
    d[:64] = np.eye(3, dtype=np.float32)[np.arange(64) % 3] * np.where(np.arange(64) % 2, -1.0, 1.0)[:, None].astype(np.float32)
    rd = pkg.api.render_desc(24, 24, 4, 5, light_samples=2)
    results = []
    # (64: big meshes walked instead of swept through their group boxes; 256: the nearest light tested again in phase 3)
    for flags in ("0", "16", "2", "18", "4", "128", "132", "64", "66", "256", "258", "80", "192", "194", "144", "146", "148"):
        monkeypatch.setenv("PTEMU_FLAGS", flags)
        sc = emu.create_scene(b)
        assert sc.uses_leaf_sweep() == (flags not in ("16", "18", "80", "144", "146", "148"))
        film, prof = sc.render(rd)
        results.append((sc.intersect(o, d), film, (prof.camera_rays, prof.bounce_rays, prof.shadow_rays, prof.env_hits)))
    for hits, film, counts in results[1:]:
        ps.assert_hits_equal(hits, results[0][0])
        assert np.array_equal(film.view(np.uint32), results[0][1].view(np.uint32))
        assert counts == results[0][2]


@pytest.mark.parametrize("scene,L,hero", [("test_bokeh_floor_small", 3, 1), ("test_bokeh_floor_small", 2, 4), ("mixed_primitives", 3, 1), ("cornell_box", 2, 1), ("test_prism_small", 2, 1)])
def test_light_prepass_changes_nothing(emu, pkg, monkeypatch, scene, L, hero):
    """A light-sample ray is bounded by the nearest hit among all lights before it is traced (few lights: the early stop at the first occluder) or traced as a
    plain closest-hit search (blob flag 1024 = PT_FLAG_NO_LIGHT_PREPASS, what a scene with more lights than pt_tuning::light_prepass_max gets): the same film
    and counters, with and without the sweep table (16) and the parked protocol (144)."""
    b = pkg.scene.SCENES[scene]()
    rd = pkg.api.render_desc(40, 32, 6, 6, light_samples=L, seed=11, hero_wavelengths=hero)
    results = []
    for flags in ("0", "1024", "1040", "1168"):
        monkeypatch.setenv("PTEMU_FLAGS", flags)
        film, prof = emu.create_scene(b).render(rd)
        results.append((film, (prof.camera_rays, prof.bounce_rays, prof.shadow_rays, prof.env_hits)))
    for film, counts in results[1:]:
        assert np.array_equal(film.view(np.uint32), results[0][0].view(np.uint32))
        assert counts == results[0][1]


def _inner_ball(emu, b, every=False):
    sc = emu.create_scene(b)
    info = lambda k: int(sc.library._debug_scene_info(sc.handle, k))
    f = np.array([info(k) for k in range(8, 13)], np.uint32).view(np.float32)
    balls = [(f[:3].astype(np.float64), float(f[3]))]
    off, count = info(13), info(14)
    for k in range(count):   # the further balls (pt_blob.h PT_MESH_MORE_*)
        q = np.array([info(1000000 + off + 4 * k + j) for j in range(4)], np.uint32).view(np.float32)
        balls.append((q[:3].astype(np.float64), float(q[3])))
    return (balls, float(f[4])) if every else (balls[0][0], balls[0][1], float(f[4]))


@pytest.mark.parametrize("mesh", ["brilliant_diamond", "monkey", "prism", "gem"])
def test_closed_meshes_get_an_inner_ball(emu, pkg, mesh):
    """mesh_surely_blocks (pt_device.h) rests on what the host found: a ball strictly INSIDE a closed mesh.  Checked here with arithmetic of its own (numpy, f64): the surface is
    closed (every undirected edge, by position, in exactly two triangles), every triangle is farther from the centre than the radius, and the centre is inside — an odd
    number of crossings along seven random rays."""
    p, f, n, mtl = pkg.scene._npz_mesh(mesh)
    p = np.asarray(p, np.float64).reshape(-1, 3); f = np.asarray(f).reshape(-1, 3)
    b = pkg.scene.hdri_test(mesh=None, hdri_size=(16, 8), importance=(0, 0))
    m = b.add_mesh(p.astype(np.float32), f, None, face_materials=pkg.api.material_id(pkg.api.TAG_MATERIAL, 0))
    b.add_mesh_instance(m, None, None)
    balls, reach = _inner_ball(emu, b, every=True)
    assert balls[0][1] > 0.02 * (p.max(0) - p.min(0)).max(), (mesh, balls[0][1])
    assert 1 <= len(balls) <= 8 and all(r >= 0.25 * 0.98 * balls[0][1] for _, r in balls), [r for _, r in balls]
    assert abs(reach - np.linalg.norm(p.max(0) - p.min(0))) < 1e-3 * reach
    tri = p.astype(np.float32).astype(np.float64)[f]                      # the vertices as the engine holds them
    g = np.array([(i, j, 12 - i - j) for i in range(13) for j in range(13 - i)], np.float64) / 12.0
    pts = np.einsum("gk,tkx->tgx", g, tri).reshape(-1, 3)                 # dense samples of every triangle
    rng = np.random.default_rng(5)
    for c, r in balls:
        for _ in range(7):   # crossing parity along random directions (independent of the faces' orientation: the authored test mesh mixes windings): odd = inside
            d = rng.normal(size=3); d /= np.linalg.norm(d)
            e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
            pv = np.cross(d, e2); det = np.einsum("ij,ij->i", e1, pv)
            ok = np.abs(det) > 1e-14
            tv = c - tri[:, 0]
            u = np.einsum("ij,ij->i", tv, pv) / np.where(ok, det, 1.0)
            qv = np.cross(tv, e1)
            v = (qv @ d) / np.where(ok, det, 1.0)
            t = np.einsum("ij,ij->i", e2, qv) / np.where(ok, det, 1.0)
            crossings = int((ok & (u > 0) & (v > 0) & (u + v < 1) & (t > 0)).sum())
            assert crossings % 2 == 1, (mesh, crossings)
        # no triangle comes closer to the centre than r / 0.98 (the host's shrink): the samples' distances bound the true distance from above
        assert np.linalg.norm(pts - c, axis=1).min() >= r / 0.98 * (1.0 - 1e-4), (mesh, r)
    for i, (ci, ri) in enumerate(balls):   # a further ball's centre lies outside the balls before it: a ball somewhere else in the body
        assert all(np.linalg.norm(ci - cj) > rj * 0.999 for cj, rj in balls[:i])


def test_an_open_mesh_gets_no_inner_ball(emu, pkg):
    """One triangle missing: three edges with a single triangle — no inside to speak of.  (A mesh with a light among its faces cannot be created at all, mesh.rs:213-232; a
    mesh INSTANCE overridden with a light keeps its ball and mesh_surely_blocks looks at the instance's material: scene.hdri_emissive_mesh in the film cases.)"""
    p, f, n, mtl = pkg.scene._npz_mesh("gem")
    f = np.asarray(f).reshape(-1, 3)
    b = pkg.scene.hdri_test(mesh=None, hdri_size=(16, 8), importance=(0, 0))
    m = b.add_mesh(p, f[:-1], None, face_materials=pkg.api.material_id(pkg.api.TAG_MATERIAL, 0))
    b.add_mesh_instance(m, None, None)
    assert _inner_ball(emu, b)[1] == 0.0


def _cube(split_edge=False):
    """A closed unit cube of 12 triangles, outward winding.  `split_edge`: the edge (0,0,0)-(1,0,0) carries a T-junction — the bottom face's triangle along it is cut
    in two at the edge's midpoint while the front face's keeps the whole edge, and the gap is "filled" with a zero-area triangle: every undirected edge is still in
    exactly two triangles."""
    p = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
    f = [(0, 2, 1), (0, 3, 2), (4, 5, 6), (4, 6, 7), (0, 1, 5), (0, 5, 4), (1, 2, 6), (1, 6, 5), (2, 3, 7), (2, 7, 6), (3, 0, 4), (3, 4, 7)]
    if split_edge:
        p.append((0.5, 0, 0))                                  # 8: the midpoint
        f[0:1] = [(0, 2, 8), (8, 2, 1)]                         # the bottom triangle (0, 2, 1) cut at the midpoint
        f.append((0, 8, 1))                                     # the sliver between the two halves and the front face's whole edge
    return np.asarray(p, np.float32), np.asarray(f, np.uint32)


def _mesh_scene(pkg, p, f, n=None, transform=None, light_at=(0.0, 0.0, 3.0), sky=False, lit_disk=None):
    b = pkg.scene.SceneBuilder()
    pkg.scene.add_library_curves(b, ["flat_zero", "flat_one"])
    b.set_environment_constant(b.curve("flat_one" if sky else "flat_zero"), 1.0 if sky else 0.0)
    b.env_sampling_probability = 0.5 if sky else 0.0
    lamp = pkg.scene.add_library_material(b, "diffuse_light_flat_x5")
    white = pkg.scene.add_library_material(b, "lambertian_white")
    glass = pkg.scene.add_library_material(b, "ggx_glass_rough")
    b.add_rect((1.0, 1.0), light_at, "Z", True, lamp)
    if lit_disk is not None:
        b.add_disk(lit_disk[0], lit_disk[1], True, lamp)
    b.add_rect((12, 12), (0.0, 0.0, -2.0), "Z", True, white)
    m = b.add_mesh(p, f, n, face_materials=pkg.api.material_id(pkg.api.TAG_MATERIAL, 0))
    b.add_mesh_instance(m, glass, transform)
    b.add_camera((-6.0, 0.5, 1.0), (0.5, 0.5, 0.5), 30.0)
    return b


def test_a_sliver_at_a_t_junction_is_not_closed(emu, pkg):
    """Round-5 advisor: a T-junction filled with a zero-area triangle pairs every edge twice, but the watertight triangle test is watertight only across edges computed from
    the SAME vertex pair — a ray can pass through such a junction.  Such a mesh is not "closed": no inner ball, no convex certificate; the plain cube has both."""
    CONVEX = 16 | 32
    for split, closed in ((False, True), (True, False)):
        p, f = _cube(split)
        sc = emu.create_scene(_mesh_scene(pkg, p, f))
        info = lambda k: int(sc.library._debug_scene_info(sc.handle, k))
        radius = float(np.array([info(11)], np.uint32).view(np.float32)[0])
        assert (radius > 0.0) == closed and ((info(15) & CONVEX) != 0) == closed, (split, radius, info(15))


def test_convex_certificates(emu, pkg):
    """pt_blob.h PT_INST_CONVEX_*: which instances the host certifies, checked against what the scenes are.  The brilliant cut in the Cornell room: OUT and IN, 238 faces safe
    all over and 64 (those at its 97-degree edges) safe away from their edges; with a lamp inside its bounding box: OUT only; a cube: OUT, and IN for the inside of its faces
    only (at a 90-degree edge a point moved inward along one face lies ON the next face's plane); scaled unevenly and rotated: still convex, still certified; the monkey (not
    convex), a smooth-shaded gem (hit normals are not the faces'), a cube seen inside out (normals inward): nothing."""
    OUT, IN = 16, 32
    def flags(b):
        sc = emu.create_scene(b)
        info = lambda k: int(sc.library._debug_scene_info(sc.handle, k))
        return info(15) & (OUT | IN), info(16), info(17)
    assert flags(pkg.scene.cornell_gem()) == (OUT | IN, 238, 64)
    lamp_inside = pkg.scene.cornell_gem()
    lamp_inside.add_rect((0.05, 0.05), (0.45, 0.45, -0.55), "Z", True, lamp_inside.material("sharp_light_fluorescent"))   # in a corner of the gem's box, outside the gem
    assert flags(lamp_inside)[0] == OUT
    p, f = _cube()
    assert flags(_mesh_scene(pkg, p, f)) == (OUT | IN, 0, 12)
    xf = pkg.scene.transform_from_data(scale=(0.5, 2.0, 1.25), rotate=[((0.3, 1.0, 0.2), 37.0)], translate=(0.2, -0.4, 0.1))
    assert flags(_mesh_scene(pkg, p, f, transform=xf)) == (OUT | IN, 0, 12)
    assert flags(_mesh_scene(pkg, p, f[:, ::-1].copy()))[0] == 0                        # inside out
    assert flags(_mesh_scene(pkg, p, f, light_at=(0.5, 0.5, 1.0005)))[0] == OUT          # a lamp 5e-4 above the cube: its box touches the cube's
    # a lit DISK beside the cube whose rim reaches into it: the reference's box of a disk has HALF its radius (disk.rs:24-28, a kept quirk) and clears the cube's — the
    # certificate asks where the light is, not where its box is (found by the GPU soak, seed 197995: inward light rays killed that reach the rim inside the cube)
    assert flags(_mesh_scene(pkg, p, f, lit_disk=(0.55, (1.3, 0.5, 0.5))))[0] == OUT
    assert flags(_mesh_scene(pkg, p, f, lit_disk=(0.55, (2.3, 0.5, 0.5))))[0] == (OUT | IN)  # (the same disk a unit farther away: clear)
    pm, fm, nm, _ = pkg.scene._npz_mesh("monkey")
    assert flags(_mesh_scene(pkg, pm, fm))[0] == 0
    pg, fg, ng, _ = pkg.scene._npz_mesh("gem")
    assert ng is not None and flags(_mesh_scene(pkg, pg, fg, ng))[0] == 0               # (the 64-triangle gem is not convex)
    # a SMOOTH-shaded convex body (the reference tree's prism.obj: 836 triangles, vertex normals up to 47 degrees off their faces): certified face by face, each face with
    # its own outward threshold (0.02 + the sine of its normals' largest deviation) and an inward margin that grows with it
    pp, fp, npr, _ = pkg.scene._npz_mesh("prism")
    got = flags(_mesh_scene(pkg, pp, fp, npr, transform=pkg.scene.transform_from_data(scale=(3.0, 3.0, 3.0))))
    assert got[0] == (OUT | IN) and got[1] + got[2] > 600, got
    # the certificate's geometry once more, with arithmetic of its own (numpy, f64), for the gem as the scene places it: convex within the slack, the safe faces really safe
    pd, fd, nd, _ = pkg.scene._npz_mesh("brilliant_diamond")
    P = pd.astype(np.float64) * 0.5 + np.array([0.0, 0.0, -0.7]); F = np.asarray(fd)
    a, b_, c = P[F[:, 0]], P[F[:, 1]], P[F[:, 2]]
    n = np.cross(b_ - a, c - a); n /= np.linalg.norm(n, axis=1)[:, None]
    d = (n * a).sum(1)
    assert (P @ n.T - d[None, :]).max() <= 2e-4
    Q = P[F] - 1e-3 * n[:, None, :]                       # every corner of every face, moved inward
    inside = ((Q.reshape(-1, 3) @ n.T - d[None, :]).max(axis=1) <= -1e-4).reshape(-1, 3).all(axis=1)
    assert int(inside.sum()) == 238


@pytest.mark.parametrize("case", ["cornell_gem", "cornell_gem_hero", "cube", "cube_transformed", "cube_lamp_close", "cube_sky", "cube_sky_hero", "prism_smooth", "prism_smooth_sky", "test_prism_glass", "cube_lit_disk"])
def test_convex_certificates_change_nothing(emu, oracle, pkg, monkeypatch, case):
    """A light-sample ray that leaves a certified body inward is dead where it is made, one that leaves it outward drops the body from its leaf mask (stage_shade,
    world_hit_sweep): with the certificates ignored (PTEMU_NO_CONVEX = pt_tuning's PT_TUNE_NO_CONVEX) the film and the counters are the same bit for bit — and both are the oracle's."""
    p, f = _cube()
    xf = pkg.scene.transform_from_data(scale=(0.5, 2.0, 1.25), rotate=[((0.3, 1.0, 0.2), 37.0)], translate=(0.2, -0.4, 0.1))
    b = {"cornell_gem": pkg.scene.cornell_gem, "cornell_gem_hero": pkg.scene.cornell_gem, "cube": lambda: _mesh_scene(pkg, p, f),
         "cube_transformed": lambda: _mesh_scene(pkg, p, f, transform=xf), "cube_lamp_close": lambda: _mesh_scene(pkg, p, f, light_at=(0.5, 0.5, 1.2)),
         # (environment rays: they leave on the normal's side but START on the side of their direction's WORLD z — pt.rs:256 —, so half of them start inside the body)
         "cube_sky": lambda: _mesh_scene(pkg, p, f, transform=xf, sky=True), "cube_sky_hero": lambda: _mesh_scene(pkg, p, f, sky=True),
         # (a smooth-shaded convex body: the hit normals are up to 47 degrees off their faces, every face has its own outward threshold)
         "prism_smooth": lambda: _mesh_scene(pkg, *pkg.scene._npz_mesh("prism")[:3], transform=pkg.scene.transform_from_data(scale=(3.0, 3.0, 3.0), rotate=[((0, 0, 1), 90.0)])),
         "prism_smooth_sky": lambda: _mesh_scene(pkg, *pkg.scene._npz_mesh("prism")[:3], transform=pkg.scene.transform_from_data(scale=(2.0, 3.0, 2.5)), sky=True),
         # (G1's scene: the prism in GLASS, 836 triangles — walked, not swept: the inside rule in mesh_walk's while-while form)
         "test_prism_glass": pkg.scene.test_prism_small,
         # (a lit disk whose rim reaches into the cube while its reference box — half the radius — clears the cube's: inward light rays reach it)
         # (with the certificate as it was before the fix this film is off by 3.4e-4)
         "cube_lit_disk": lambda: _mesh_scene(pkg, p, f, lit_disk=(0.55, (1.3, 0.5, 0.5)))}[case]()
    rd = pkg.api.render_desc(48, 40, 8, 10, light_samples=3, seed=12, hero_wavelengths=4 if case.endswith("hero") else 1)
    with_cert, pw = emu.create_scene(b).render(rd)
    monkeypatch.setenv("PTEMU_NO_CONVEX", "1")
    without, po = emu.create_scene(b).render(rd)
    monkeypatch.delenv("PTEMU_NO_CONVEX")
    assert np.array_equal(with_cert.view(np.uint32), without.view(np.uint32))
    assert (pw.bounce_rays, pw.shadow_rays, pw.env_hits) == (po.bounce_rays, po.shadow_rays, po.env_hits)
    # a path segment that starts inside the body ends the body's sweep at its first interior acceptance (PT_PATH_INSIDE_MARK, mesh_walk's `inside`): with the mark ignored, the same bits
    sc = emu.create_scene(b)
    stops = lambda: int(sc.library._debug_scene_info(sc.handle, 18))
    before = stops()
    monkeypatch.setenv("PTEMU_NO_INSIDE", "1")
    full_sweep, pf = sc.render(rd)
    monkeypatch.delenv("PTEMU_NO_INSIDE")
    assert stops() == before
    sc.render(rd)
    assert (stops() > before + 1000) == (case.startswith("cornell_gem") or case == "test_prism_glass"), (case, stops() - before)   # (a glass body's inner bounces; a Lambertian body has none)
    assert np.array_equal(with_cert.view(np.uint32), full_sweep.view(np.uint32))
    assert (pw.bounce_rays, pw.shadow_rays, pw.env_hits) == (pf.bounce_rays, pf.shadow_rays, pf.env_hits)
    ref, pr = oracle.create_scene(b).render(rd)
    ps.check_film(with_cert, ref, pw, pr)
    assert with_cert[..., :3].max() > 0.0


def test_reference_known_answers_on_the_lane_logic(emu, oracle, pkg):
    """The reference's own known-answer tests for the path, which tests/test_oracle.py runs on the oracle, on the
    engine's lane logic; tests/test_gpu_parity.py runs the same list on the GPU."""
    import test_oracle as kat
    kat.check_ggx_properties(pkg, emu)
    kat.test_ggx_sample_matches_eval(pkg, emu)
    kat.test_sharp_light_pdf_integrates_to_one(pkg, emu)
    kat.test_lambertian_and_light(pkg, emu)
    kat.test_curves(pkg, emu)
    kat.test_world_intersection(pkg, emu)
    kat.test_intersection_against_brute_force(pkg, emu)
    kat.test_reference_instance_case(pkg, emu)
    kat.test_panorama_camera_directions(pkg, emu)
    kat.test_white_furnace(pkg, emu, cmf=oracle)


def test_specialised_shade_forms_change_nothing(emu, pkg, monkeypatch):
    """The shade forms compiled without the environment-sampling branch (env_sampling_probability = 0) and without the GGX code (no
    GGX material in the scene) give the bits of the general form."""
    for scene, hero in (("cornell_box", 1), ("cornell_gem", 1), ("cornell_box", 4)):
        b = pkg.scene.SCENES[scene]()
        rd = pkg.api.render_desc(24, 20, 5, 6, hero_wavelengths=hero)
        lean, plean = emu.create_scene(b).render(rd)
        for form in ("1", "2"):
            monkeypatch.setenv("PTEMU_SHADE_FORM", form)
            general, pgen = emu.create_scene(b).render(rd)
            monkeypatch.delenv("PTEMU_SHADE_FORM")
            assert np.array_equal(lean.view(np.uint32), general.view(np.uint32)), (scene, form)
            assert (plean.bounce_rays, plean.shadow_rays) == (pgen.bounce_rays, pgen.shadow_rays)


def many_analytic_scene(pkg, n=80, seed=5, disks=True):
    """More than 64 instances, none a mesh — rectangles of every axis, spheres (some under uneven scales and rotations), disks (one lit: the reference's box of a disk has
    half its radius, so nothing is culled at the top level beside one) — the scenes that have no sweep table and park at nothing."""
    rng = np.random.default_rng(seed)
    b = pkg.scene.SceneBuilder()
    pkg.scene.add_library_curves(b, ["flat_zero", "flat_one"])
    b.set_environment_constant(b.curve("flat_one"), 0.3)
    b.env_sampling_probability = 0.3
    lamp = pkg.scene.add_library_material(b, "diffuse_light_flat_x5")
    mats = [pkg.scene.add_library_material(b, m) for m in ("lambertian_white", "lambertian_red", "ggx_gold", "ggx_glass_rough")]
    b.add_rect((10, 10), (0.0, 0.0, -1.2), "Z", True, mats[0])
    b.add_rect((1.0, 1.0), (0.0, 0.0, 2.5), "Z", True, lamp)
    for k in range(n - 2):
        at = rng.uniform(-1.5, 1.5, 3).tolist()
        m = lamp if k % 9 == 0 else mats[int(rng.integers(len(mats)))]
        xf = pkg.scene.transform_from_data(rng.uniform(0.5, 1.5, 3).tolist(), [(rng.normal(size=3).tolist(), float(rng.uniform(-180, 180)))], None) if k % 4 == 0 else None
        kind = int(rng.integers(3)) if disks else int(rng.integers(2))
        if kind == 0: b.add_sphere(float(rng.uniform(0.05, 0.25)), at, m, xf)
        elif kind == 1: b.add_rect(tuple(rng.uniform(0.1, 0.6, 2).tolist()), at, "XYZ"[int(rng.integers(3))], bool(rng.integers(2)), m, xf)
        else: b.add_disk(float(rng.uniform(0.1, 0.4)), at, bool(rng.integers(2)), m, xf)
    b.add_camera((-5.0, 0.3, 0.8), (0.0, 0.0, 0.0), 40.0)
    return b


@pytest.mark.parametrize("case", ["many_analytic", "many_analytic_no_disk", "many_analytic_300"])
def test_many_analytic_instances(emu, oracle, pkg, case):
    """More than 64 instances of every analytic kind, none a mesh, through the top-level walk in the parked kernels' protocol: the oracle's film.  (The scenes were made for
    round 6's top level by GROUPS — the leaves of the tree six at a time under the union of their boxes, bit-identical on them and on the GPU, and slower than the walk:
    profiles/r6_experiments.md section 13, the code in profiles/r6_scripts/r6_topgroups.patch.)"""
    b = {"many_analytic": lambda: many_analytic_scene(pkg), "many_analytic_no_disk": lambda: many_analytic_scene(pkg, 90, 8, disks=False),
         "many_analytic_300": lambda: many_analytic_scene(pkg, 300, 11)}[case]()
    rd = pkg.api.render_desc(48, 40, 6, 8, light_samples=3, seed=9, hero_wavelengths=4 if case == "many_analytic_no_disk" else 1)
    film, prof = emu.create_scene(b).render(rd)
    ref, rp = oracle.create_scene(b).render(rd)
    ps.check_film(film, ref, prof, rp)
    assert film[..., :3].max() > 0.0


@pytest.mark.parametrize("scene", ["cornell_gem", "hdri_c4_small", "test_prism_small", "test_bokeh_floor_gem_small"])
def test_mesh_shortcuts_change_nothing(emu, pkg, monkeypatch, scene):
    """mesh_surely_blocks (a closed mesh's inner balls) and mesh_surely_missed (its 18-DOP slabs) decide a ray at a mesh without the triangle tests the reference runs —
    on f32 error budgets.  With both switched off (PTEMU_NO_MESH_SHORTCUTS = PT_TUNE_NO_MESH_SHORTCUTS: the mesh records lose their ball and their slab table, the searches run
    in full) the film and the counters are the same bit for bit — every decision of theirs, not only those that move the film by more than the parity bar."""
    b = pkg.scene.SCENES[scene]()
    rd = pkg.api.render_desc(64, 48, 6, 8, light_samples=3, seed=17)
    film, prof = emu.create_scene(b).render(rd)
    monkeypatch.setenv("PTEMU_NO_MESH_SHORTCUTS", "1")
    full_sc = emu.create_scene(b)
    assert int(full_sc.library._debug_scene_info(full_sc.handle, 11)) == 0   # (the first mesh's inner ball radius: gone)
    full, pf = full_sc.render(rd)
    monkeypatch.delenv("PTEMU_NO_MESH_SHORTCUTS")
    assert np.array_equal(film.view(np.uint32), full.view(np.uint32))
    assert (prof.bounce_rays, prof.shadow_rays, prof.env_hits) == (pf.bounce_rays, pf.shadow_rays, pf.env_hits)
