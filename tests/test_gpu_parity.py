"""GPU: the HIP engine, called through the C ABI (libptamd.so), against the CPU oracle and the golden vectors.
Run on the MI355X box:  python -m pytest tests -m gpu -x -q"""
import ctypes as C
import os

import numpy as np
import pytest

import parity_suite as ps
from util import fptr, numerics

pytestmark = pytest.mark.gpu


def test_engine_is_the_hip_library(engine, pkg):
    assert engine.path == pkg.LIBRARY_PATH and engine.prefix == "pt_"
    info = engine.device_info()
    assert "gfx950" in info, info


def test_device_arithmetic_matches_x86(engine, oracle):
    """The numeric contract on the device: elementary functions, IEEE divide / sqrt, no fma contraction."""
    rng = np.random.default_rng(0)
    n = 1 << 16
    fn = engine.lib.pt_debug_numerics
    fn.restype = C.c_int32
    fn.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]

    def dev(which, x, y):
        out = np.zeros_like(x)
        engine.check(fn(which, x.size, fptr(x), fptr(y), fptr(out)))
        return out
    cases = {0: (-7, 7), 1: (-7, 7), 2: (-100, 89), 4: (-1, 1), 6: (-40, 40), 7: (1e-6, 1e6)}
    for which, (lo, hi) in cases.items():
        x = rng.uniform(lo, hi, n).astype(np.float32); y = np.zeros_like(x)
        assert np.array_equal(dev(which, x, y).view(np.uint32), numerics(oracle, which, x, y).view(np.uint32)), which
    x = rng.uniform(0, 1, n).astype(np.float32); y = rng.uniform(1, 500, n).astype(np.float32)
    assert np.array_equal(dev(3, x, y).view(np.uint32), numerics(oracle, 3, x, y).view(np.uint32))
    x = rng.uniform(-1, 1, n).astype(np.float32); y = rng.uniform(-1, 1, n).astype(np.float32)
    assert np.array_equal(dev(5, x, y).view(np.uint32), numerics(oracle, 5, x, y).view(np.uint32))
    x = (rng.standard_normal(n) * 10.0 ** rng.uniform(-20, 20, n)).astype(np.float32)
    y = (rng.standard_normal(n) * 10.0 ** rng.uniform(-20, 20, n)).astype(np.float32)
    with np.errstate(all="ignore"):
        assert np.array_equal(dev(8, x, y).view(np.uint32), (x / y).view(np.uint32))            # IEEE division, denormals kept
        assert np.array_equal(dev(9, np.abs(x), y).view(np.uint32), np.sqrt(np.abs(x)).view(np.uint32))
        assert np.array_equal(dev(10, x, y).view(np.uint32), ((x * y).astype(np.float32) + x).view(np.uint32))  # not fused


def test_tabulated_curves_through_their_cell_tables(engine, oracle):
    ps.curve_table_parity(engine, oracle)


def test_colour_matching_fit_for_every_wavelength(engine, oracle):
    """k_accumulate's cheaper evaluation of the CIE fit (csrc/pt_device.h: gaussian64_fast) on the device against the oracle's, for every f32 of the range it is
    used in (10 027 009 wavelengths x 3 outputs), and the contract's form outside the range."""
    from test_xyz_bar import FAST_LO, FAST_HI, every_f32, oracle_xyz
    fn = engine.lib.pt_debug_numerics
    fn.restype = C.c_int32
    fn.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    rng = np.random.default_rng(5)
    outside = np.concatenate([every_f32(3599.0, 3600.0)[:-1], every_f32(8000.0, 8001.0)[1:], rng.uniform(0.0, 3600.0, 20000).astype(np.float32),
                              rng.uniform(8000.0, 30000.0, 20000).astype(np.float32)])
    for xs in list(np.array_split(every_f32(FAST_LO, FAST_HI), 4)) + [outside]:
        xs = np.ascontiguousarray(xs); y = np.zeros_like(xs)
        want = oracle_xyz(oracle, xs)
        for c in range(3):
            out = np.zeros_like(xs)
            engine.check(fn(11 + c, xs.size, fptr(xs), fptr(y), fptr(out)))
            assert np.array_equal(out.view(np.uint32), np.ascontiguousarray(want[:, c]).view(np.uint32)), c


@pytest.mark.parametrize("scene", ["cornell_box", "cornell_gem", "mixed_primitives", "mixed_small", "white_furnace", "hdri_small", "hdri_c4_small"])
def test_closest_hits_bit_exact(engine, oracle, scene):
    ps.intersect_parity(engine, oracle, scene, n=1 << 16)


@pytest.mark.parametrize("scene", ["cornell_box", "cornell_gem", "panorama_test", "hdri_test", "mixed_primitives"])
def test_camera_samples_bit_exact(engine, oracle, scene):
    """Camera trait surface (SURVEY a22 + the head of a1 / a2): thin-lens rays with the aperture rejection loop, panorama rays, jitter and wavelength —
    the stage k_generate runs, probed through pt_camera_samples against the oracle, bit for bit."""
    ps.camera_parity(engine, oracle, scene, width=1920, height=1080, n=1 << 16)
    ps.camera_parity(engine, oracle, scene, width=33, height=21, n=4096, seed=5, wavelength=(555.0, 560.0), camera_index=0)


@pytest.mark.parametrize("scene", ["cornell_box", "cornell_gem", "mixed_primitives"])
def test_materials_bit_exact(engine, oracle, scene):
    ps.material_parity(engine, oracle, scene, n=1 << 14)


@pytest.mark.parametrize("scene,w,h,spp,mb,kw", [
    ("cornell_box", 256, 256, 16, 4, {}),                     # C1 of BASELINE.json at full size
    ("cornell_box", 33, 21, 4, 8, {"tile": (16, 16)}),
    ("cornell_box", 64, 64, 5, 8, {"only_direct": True}),
    ("cornell_box", 64, 64, 5, 8, {"light_samples": 0}),
    ("cornell_box", 64, 64, 13, 3, {"min_bounces": 4}),
    ("cornell_gem", 96, 54, 8, 12, {}),                       # C3 shape at reduced size
    ("mixed_primitives", 64, 64, 8, 6, {"light_samples": 3, "seed": 5}),
    ("mixed_small", 64, 64, 8, 6, {"light_samples": 3, "seed": 6}),
    ("panorama_test", 128, 64, 8, 6, {}),                     # PanoramaCamera
    ("cornell_box", 96, 64, 23, 4, {"phase_samples": 23, "tile": (96, 64)}),   # NaiveRenderer: one sum over all samples (naive.rs:82-103)
    ("empty_env", 1, 1, 1, 0, {}),                            # edge cases: no instances, 1x1 film, no bounces
    ("empty_env", 5, 3, 2, 4, {"light_samples": 3}),
    ("cornell_box", 3, 2, 1, 1, {"tile": (64, 64)}),          # film smaller than a tile
    ("white_furnace", 32, 32, 12, 8, {"light_samples": 6}),
    ("hdri_small", 64, 64, 8, 4, {"light_samples": 6}),
    ("test_prism_small", 128, 128, 8, 8, {"light_samples": 3}),     # the reference tree's test_prism.toml (transform stack + lights + environment sampling): the GENERAL kernel forms, off the tuned path
    ("test_prism_small", 96, 64, 6, 6, {"light_samples": 2, "hero_wavelengths": 4}),
    ("test_bokeh_small", 128, 128, 8, 8, {"light_samples": 2}),      # G2, the reference tree's test_bokeh.toml: 82 sphere lights > 64 instances = no sweep table, the top-level BVH walk
    ("test_bokeh_floor_small", 128, 96, 8, 8, {"light_samples": 3}),  # ... with a floor under the lights: the 82-entry light list sampled, light-sample rays through the top-level walk
    ("test_bokeh_floor_small", 96, 64, 6, 6, {"light_samples": 2, "hero_wavelengths": 4}),
    ("test_bokeh_floor_gem_small", 128, 96, 8, 8, {"light_samples": 2}),  # ... and a mesh: rays park at the mesh AND are evicted from the top-level walk
    ("test_bokeh_floor_gem_small", 96, 64, 6, 6, {"light_samples": 3, "hero_wavelengths": 4}),
    ("hdri_emissive_mesh", 96, 96, 8, 4, {"light_samples": 3}),     # empty light list, but a mesh instance overridden with a light material (round-3 advisor): its hits emit, take no item
    ("hdri_emissive_mesh", 64, 64, 6, 4, {"light_samples": 2, "hero_wavelengths": 4}),
    ("disk_lamp", 160, 112, 8, 5, {"light_samples": 2, "seed": 2}),   # one disk lamp 1e-4 under its ceiling: the lean form tests a light-sample ray against the scene's only light at the vertex, the ceiling's items die there and are not listed
    ("disk_lamp", 96, 64, 6, 6, {"light_samples": 3, "hero_wavelengths": 4}),
    ("cornell_box", 128, 128, 12, 8, {"hero_wavelengths": 4}),      # C5 shape: four wavelengths per path
    ("cornell_gem", 64, 48, 6, 12, {"hero_wavelengths": 4}),
    ("hdri_small", 48, 48, 6, 4, {"hero_wavelengths": 4, "light_samples": 3}),
    ("hdri_c4_small", 96, 96, 6, 4, {"light_samples": 6}),   # C4 scene (monkey mesh: blob too big for LDS -> HBM/L2 path)
    ("hdri_c4_small", 80, 64, 5, 4, {"light_samples": 8}),   # the most light samples an item can hold (the per-wave lists of live rays at their longest)
    ("cornell_gem", 64, 48, 5, 8, {"light_samples": 8}),
    ("cornell_box", 64, 48, 5, 6, {"light_samples": 8}),
    ("fog_ball", 160, 120, 12, 10, {"medium_aware": True}),  # the medium-aware walk (k_shade_medium)
    ("fog_ball", 96, 64, 8, 12, {"medium_aware": True, "light_samples": 3, "min_bounces": 3, "seed": 9}),
    ("cornell_box", 96, 96, 8, 6, {"medium_aware": True}),
    ("cornell_gem", 96, 54, 6, 8, {"medium_aware": True}),   # (parked traversal under the medium-aware vertex kernel)
    ("hdri_c4_small", 64, 64, 6, 4, {"medium_aware": True, "light_samples": 3}),
])
def test_film_parity(engine, oracle, scene, w, h, spp, mb, kw):
    ps.render_parity(engine, oracle, scene, w, h, spp, mb, flat=(scene, w, spp) == ("cornell_box", 256, 16), **kw)   # (C1: the flat bar)


# BASELINE.json quotes its configurations at 1024 / 2048 / 4096 samples per pixel (src/renderer/tiled.rs:347-398: 100-400 phases of ten samples per pixel, each phase's
# sum added to the pixel, one division at the end).  north_star's bar — XYZ film L-inf < 1e-4 at matched seeds — is therefore shown AT those sample counts, flat (no ulp
# allowance), counters equal, on films small enough for the oracle to finish in under a minute on the GPU box's host cores (round-5 verdict, item 1).
@pytest.mark.parametrize("scene,w,h,spp,mb,kw", [
    ("cornell_box", 128, 128, 1024, 8, {}),                                   # C2's sample count and depth
    ("cornell_box", 96, 96, 1024, 8, {"hero_wavelengths": 4}),                # C5's
    ("cornell_gem", 96, 54, 4096, 12, {}),                                    # C3's (the film's 16:9 shape)
    ("hdri_c4_small", 64, 64, 2048, 4, {"light_samples": 6}),                 # C4's
    ("ref_candela_calibration", 128, 128, 1024, 12, {"wavelength": (555.0, 560.0), "only_direct": True, "light_samples": 1, "min_bounces": 2}),   # data/config_test_candela_calibration.toml as it is written
])
def test_film_parity_at_baseline_sample_counts(engine, oracle, scene, w, h, spp, mb, kw):
    m = ps.render_parity(engine, oracle, scene, w, h, spp, mb, flat=True, **kw)
    assert m["brightest"] > 0.0


@pytest.mark.parametrize("batch", [None, "300000"])
def test_bench_sample_windows_against_the_oracle(engine, oracle, pkg, monkeypatch, batch):
    """bench.py's steps are windows of one long render (first_sample = k * 1024 of 25 600 samples per pixel; the film of a window is the un-normalised sum of its phases).
    Its last window — sample indices 24 576 .. 25 599, the largest Philox counters and the largest sums the bench produces — against the oracle's same window, flat 1e-4 on
    sums of several hundred, counters equal; with PT_AMD_BATCH small enough to cut the window into several passes too (the same film bit for bit)."""
    if batch:
        monkeypatch.setenv("PT_AMD_BATCH", batch)
    b = pkg.scene.cornell_box()
    rd = pkg.api.render_desc(64, 64, 25600, 8, first_sample=24576, sample_count=1024)
    film, prof = engine.create_scene(b).render(rd)
    if batch:
        monkeypatch.delenv("PT_AMD_BATCH")
        assert prof.kernel_launches[0] >= 4, prof.kernel_launches[0]          # several passes: k_generate launched once per pass
        whole, pw = engine.create_scene(b).render(rd)
        assert np.array_equal(film.view(np.uint32), whole.view(np.uint32)) and (prof.bounce_rays, prof.shadow_rays) == (pw.bounce_rays, pw.shadow_rays)
    ref, rprof = oracle.create_scene(b).render(rd)
    # The window's film is a SUM of 1024 samples (up to 1340 where the camera looks into the lamp: one unit in the last place of such a value is 1.2e-4, so the flat bar is
    # not a statement about sums).  Held to two bars: as the sums are — 1e-4 or 8 ulp, relative 2e-5, counters equal —, and flat 1e-4 as the film those 1024 samples make
    # (the sum / 1024: a power of two, the same bits with another exponent), which is what north_star's bar speaks of.
    m = ps.check_film(film, ref, prof, rprof)
    assert m["brightest"] > 100.0, m                                           # an un-normalised sum of 1024 samples
    scale = np.float32(1.0 / 1024.0)
    ps.check_film(film * scale, ref * scale, flat=True)
    first, pf = engine.create_scene(b).render(pkg.api.render_desc(64, 64, 25600, 8, first_sample=0, sample_count=1024))
    assert not np.array_equal(first, film) and pf.camera_rays == prof.camera_rays == 64 * 64 * 1024


# The self-contained scene files of the reference tree (data/scenes/*.toml that need no OBJ / HDRI the tree lacks; SURVEY section 4's validation list) on the engine.
# The builders are those files value for value: tests/test_reference_fixtures.py::test_reference_scene_builders_are_the_reference_scene_files renders the reference's
# own file and the builder on the oracle, bit for bit, wherever /root/reference exists.
@pytest.mark.parametrize("hero", [1, 4])
@pytest.mark.parametrize("name", ["candela_calibration", "cornell_box_parallel_prism", "cornell_box_single_orb_caustic", "sun_test", "test_blackbox", "test_lighting_north",
                                  "test_nee_sphere", "test_rtiow_scene_2", "test_sampling_methods"])
def test_reference_tree_scenes_on_the_engine(engine, oracle, pkg, name, hero):
    assert name in pkg.scene.REFERENCE_TREE_SCENES
    kw = {"wavelength": (555.0, 560.0), "only_direct": True, "light_samples": 1} if name == "candela_calibration" else {"light_samples": 2}
    m = ps.render_parity(engine, oracle, "ref_" + name, 160, 128, 32 if hero == 1 else 12, 8, flat=True, hero_wavelengths=hero, seed=3, **kw)
    # (two of the nine are black by construction: test_lighting_north's camera sits inside its opaque unit sphere, and cornell_box_parallel_prism's two-sided lamp of
    # Reverse sidedness shows every ray its dark face while the room's ceiling and back wall hide the sun)
    assert (m["brightest"] > 0.0) == (name not in ("test_lighting_north", "cornell_box_parallel_prism")), m
    ps.intersect_parity(engine, oracle, "ref_" + name, n=1 << 14)


def test_golden_vectors(engine):
    for name in ps.GOLDEN_RENDERS:
        film, prof, ref, counters = ps.golden_render(engine, name)
        ps.check_film(film, ref, prof, counters)
    for scene in ("cornell_box", "mixed_primitives", "cornell_gem"):
        got, want = ps.golden_hits(engine, scene)
        ps.assert_hits_equal(got, want)
        ps.golden_materials(engine, scene)


def test_reference_known_answers_on_the_engine(engine, oracle, pkg):
    """the known-answer tests the reference's own test modules hold for this path (GGX properties, regression
    seed and fixed pairs of ggx.rs:637-926; the cos^n normalisation of sharp_light.rs:229; the world ray of world/mod.rs:282;
    the instance case of instance.rs; the white furnace scene) replayed on the HIP engine — the same checks tests/test_oracle.py
    runs on the oracle, here through the C ABI of the product."""
    import test_oracle as kat
    kat.check_ggx_properties(pkg, engine)
    kat.test_ggx_sample_matches_eval(pkg, engine)
    kat.test_sharp_light_pdf_integrates_to_one(pkg, engine)
    kat.test_lambertian_and_light(pkg, engine)
    kat.test_curves(pkg, engine)
    kat.test_world_intersection(pkg, engine)
    kat.test_intersection_against_brute_force(pkg, engine)
    kat.test_reference_instance_case(pkg, engine)
    kat.test_panorama_camera_directions(pkg, engine)
    kat.test_hero_wavelengths_follow_the_single_wavelength_path(pkg, engine)
    kat.test_white_furnace(pkg, engine, cmf=oracle)


def test_shards_and_sample_ranges(engine):
    ps.shards_and_ranges(engine)


def test_batching_and_lds_staging_are_invisible(engine, pkg, monkeypatch):
    """Same film bit for bit whatever the pass size, and whether the scene blob is read from LDS or HBM."""
    b = pkg.scene.cornell_box()
    rd = pkg.api.render_desc(96, 80, 23, 5)
    base, pbase = engine.create_scene(b).render(rd)
    monkeypatch.setenv("PT_AMD_BATCH", "20000")
    small, psmall = engine.create_scene(b).render(rd)
    assert np.array_equal(base, small) and pbase.bounce_rays == psmall.bounce_rays and pbase.shadow_rays == psmall.shadow_rays
    monkeypatch.delenv("PT_AMD_BATCH")
    monkeypatch.setenv("PT_AMD_NO_LDS", "1")
    nolds, _ = engine.create_scene(b).render(rd)
    assert np.array_equal(base, nolds)
    again, _ = engine.create_scene(b).render(rd)
    assert np.array_equal(nolds, again)


def test_specialised_shade_forms_change_nothing(engine, pkg, monkeypatch):
    """Scenes with env_sampling_probability = 0 run the k_shade form compiled without the environment-sampling branch;
    forcing the general form must give the same film bit for bit."""
    for scene, hero in (("cornell_box", 1), ("cornell_gem", 1), ("cornell_box", 4)):
        b = pkg.scene.SCENES[scene]()
        rd = pkg.api.render_desc(96, 80, 11, 8, hero_wavelengths=hero)
        lean, plean = engine.create_scene(b).render(rd)
        for form in ("1", "2"):
            monkeypatch.setenv("PT_AMD_SHADE_FORM", form)
            general, pgen = engine.create_scene(b).render(rd)
            monkeypatch.delenv("PT_AMD_SHADE_FORM")
            assert np.array_equal(lean.view(np.uint32), general.view(np.uint32)), (scene, form)
            assert (plean.bounce_rays, plean.shadow_rays) == (pgen.bounce_rays, pgen.shadow_rays)


def test_filtered_slab_test_and_culling_change_nothing(engine, pkg, monkeypatch):
    """The fast paths of the BVH walk (reciprocal-filtered slab test, culling by the closest hit) must give the films and
    hits of the plain six-division walk bit for bit."""
    import parity_suite
    rd = pkg.api.render_desc(128, 96, 11, 8)
    for scene in ("cornell_box", "cornell_gem", "mixed_primitives"):
        b = pkg.scene.SCENES[scene]()
        o, d = parity_suite.golden_rays(scene, 1 << 16, 77)
        fast = engine.create_scene(b)
        film_fast, prof_fast = fast.render(rd)
        hits_fast = fast.intersect(o, d)
        monkeypatch.setenv("PT_AMD_EXACT_SLAB", "1"); monkeypatch.setenv("PT_AMD_NO_CULL", "1")
        plain = engine.create_scene(b)
        monkeypatch.delenv("PT_AMD_EXACT_SLAB"); monkeypatch.delenv("PT_AMD_NO_CULL")
        film_plain, prof_plain = plain.render(rd)
        assert np.array_equal(film_fast, film_plain), scene
        assert (prof_fast.bounce_rays, prof_fast.shadow_rays) == (prof_plain.bounce_rays, prof_plain.shadow_rays)
        parity_suite.assert_hits_equal(hits_fast, plain.intersect(o, d))


@pytest.mark.parametrize("scene,L", [("cornell_box", 2), ("mixed_small", 3), ("mixed_primitives", 3), ("cornell_gem", 2)])
def test_known_light_changes_nothing(engine, pkg, monkeypatch, scene, L):
    """Phase 3 of a light-sample ray takes the distance of the light that bounds it from the light pre-pass instead of testing that light
    again (sweep_run's known_inst).  Switched off: the same film, counters and hits bit for bit."""
    import parity_suite
    b = pkg.scene.SCENES[scene]()
    rd = pkg.api.render_desc(160, 120, 9, 8, light_samples=L, seed=4)
    o, d = parity_suite.golden_rays(scene, 1 << 15, 5)
    ref = engine.create_scene(b)
    base, pbase = ref.render(rd)
    hits = ref.intersect(o, d)
    for env in ({"PT_AMD_NO_KNOWN_LIGHT": "1"},):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        other = engine.create_scene(b)
        for k in env:
            monkeypatch.delenv(k)
        film, prof = other.render(rd)
        assert np.array_equal(base.view(np.uint32), film.view(np.uint32)), env
        assert (pbase.bounce_rays, pbase.shadow_rays, pbase.env_hits) == (prof.bounce_rays, prof.shadow_rays, prof.env_hits), env
        parity_suite.assert_hits_equal(hits, other.intersect(o, d))


@pytest.mark.parametrize("scene,L", [("cornell_gem", 2), ("hdri_c4_small", 6), ("mixed_primitives", 3)])
def test_traversal_forms_change_nothing(engine, pkg, monkeypatch, scene, L):
    """Hybrid scenes (sweep table + walked meshes): parked rays resumed in full waves, walked meshes in line, mesh sweep
    off, no sweep table (PT_AMD_NO_SWEEP: the parked kernels over the top-level tree; with PT_AMD_NO_PARK the per-lane two-level walk), core-only / no LDS
    staging — all the same film bit for bit.  One workgroup per CU makes the
    segments long enough for the park queue to fill and drain several times per launch."""
    b = pkg.scene.SCENES[scene]()
    rd = pkg.api.render_desc(256, 192, 12, 8, light_samples=L, seed=9)
    monkeypatch.setenv("PT_AMD_BLOCKS_PER_CU", "1")
    ref_scene = engine.create_scene(b)
    assert ref_scene.uses_leaf_sweep()
    base, pbase = ref_scene.render(rd)
    # (the default: workgroups of 256 everywhere — the gem scene's light-sample kernel took 512 with the whole 66 KB blob in LDS until round 6, and still does when the convex
    # certificate that keeps its light rays off the mesh is switched off; the C4 scene's 675 KB blob never)
    assert scene == "mixed_primitives" or pbase.stage_items[6] == 0, pbase.stage_items[6]
    if scene == "cornell_gem":
        monkeypatch.setenv("PT_AMD_NO_CONVEX", "1")
        film, prof = engine.create_scene(b).render(rd)
        monkeypatch.delenv("PT_AMD_NO_CONVEX")
        assert prof.stage_items[6] == 512 and np.array_equal(base.view(np.uint32), film.view(np.uint32))
    for env in ({"PT_AMD_NO_PARK": "1"}, {"PT_AMD_NO_MESH_SWEEP": "1"}, {"PT_AMD_NO_SWEEP": "1"}, {"PT_AMD_NO_LDS": "1"}, {"PT_AMD_NO_CORE_LDS": "1"},
                {"PT_AMD_BLOCKS_PER_CU": "32"}, {"PT_AMD_PARK_DYNAMIC": "0"}, {"PT_AMD_PARK_DYNAMIC": "1"}, {"PT_AMD_PARK_DYNAMIC": "1", "PT_AMD_PARK_BLOCKS_PER_CU": "1"},
                {"PT_AMD_LDS_ALL_LIMIT": "65536"}, {"PT_AMD_LDS_ALL_LIMIT": "65536", "PT_AMD_PARK_DYNAMIC": "0"}, {"PT_AMD_LDS_ALL_LIMIT": "4096"},
                {"PT_AMD_NO_MESH_SWEEP": "1", "PT_AMD_NO_PARK": "1"}, {"PT_AMD_NO_SWEEP": "1", "PT_AMD_NO_PARK": "1"}, {"PT_AMD_NO_SWEEP": "1", "PT_AMD_NO_AXIS_SCAN": "1"},
                {"PT_AMD_WALK_EVICT_BELOW": "1", "PT_AMD_WALK_SEARCH_BELOW": "1"}, {"PT_AMD_WALK_EVICT_BELOW": "64", "PT_AMD_WALK_SEARCH_BELOW": "64"},
                {"PT_AMD_NO_AXIS_SCAN": "1"}, {"PT_AMD_NO_AXIS_SCAN": "1", "PT_AMD_WALK_EVICT_BELOW": "1", "PT_AMD_WALK_SEARCH_BELOW": "1"},
                {"PT_AMD_PARK_BLOCK": "256"}, {"PT_AMD_PARK_BLOCK": "512"}, {"PT_AMD_PARK_BLOCK": "1024"}, {"PT_AMD_PARK_BLOCK": "512", "PT_AMD_BLOCKS_PER_CU": "32"},
                {"PT_AMD_PARK_BLOCK": "1024", "PT_AMD_BLOCKS_PER_CU": "8"}):
        # (PT_AMD_PARK_BLOCK, round 4: the parked kernels in workgroups of 512 / 1024 threads that stage the whole blob — the gem scene's 65 KB — while k_shade keeps
        # its core-only staging; hdri_c4_small's blob is too big for that and keeps its 256-thread forms)
        # (the parked kernels with one static segment per workgroup / with units taken from a counter by persistent workgroups; the whole
        # blob staged in LDS however large — the gem's is 65 KB, staged as its core section by default — / only ever the core section;
        # the mesh walks of a resumed wave without eviction and short searches / leaving at every chance; axis-parallel rays walked instead of
        # scanned by the wave — hdri_c4_small's environment samples at the pole of the map are such rays)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        film, prof = engine.create_scene(b).render(rd)
        for k in env:
            monkeypatch.delenv(k) if k != "PT_AMD_BLOCKS_PER_CU" else monkeypatch.setenv(k, "1")
        assert np.array_equal(base.view(np.uint32), film.view(np.uint32)), env
        assert (pbase.bounce_rays, pbase.shadow_rays, pbase.env_hits) == (prof.bounce_rays, prof.shadow_rays, prof.env_hits), env
        # (the big-workgroup form really ran where it can — the gem scene's 66 KB blob — and not for hdri_c4_small's 675 KB: never a silent fall-back)
        if "PT_AMD_PARK_BLOCK" in env and scene != "mixed_primitives":
            assert prof.stage_items[6] == (int(env["PT_AMD_PARK_BLOCK"]) % 256 and 0 or (int(env["PT_AMD_PARK_BLOCK"]) if env["PT_AMD_PARK_BLOCK"] != "256" else 0) if scene == "cornell_gem" else 0), (env, prof.stage_items[6])


@pytest.mark.parametrize("scene,L,hero", [("test_bokeh_floor_small", 3, 1), ("test_bokeh_floor_small", 2, 4), ("mixed_primitives", 3, 1), ("test_prism_small", 2, 1), ("cornell_gem", 2, 1)])
def test_light_prepass_changes_nothing(engine, pkg, monkeypatch, scene, L, hero):
    """pt_tuning::light_prepass_max: a scene with more lights than that traces its light-sample rays as plain closest-hit searches instead of bounding each by
    the nearest hit among all lights first (the default for test_bokeh's 82 lights).  Forced either way — and through the general walk and the per-lane walk —
    the film, the counters and the hits are the same bit for bit."""
    import parity_suite
    b = pkg.scene.SCENES[scene]()
    rd = pkg.api.render_desc(160, 128, 8, 6, light_samples=L, seed=23, hero_wavelengths=hero)
    base, pbase = engine.create_scene(b).render(rd)
    for env in ({"PT_AMD_LIGHT_PREPASS_MAX": "1"}, {"PT_AMD_LIGHT_PREPASS_MAX": "4294967295"}, {"PT_AMD_LIGHT_PREPASS_MAX": "1", "PT_AMD_NO_SWEEP": "1"},
                {"PT_AMD_LIGHT_PREPASS_MAX": "4294967295", "PT_AMD_NO_SWEEP": "1", "PT_AMD_NO_PARK": "1"}, {"PT_AMD_LIGHT_PREPASS_MAX": "1", "PT_AMD_NO_LDS": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        film, prof = engine.create_scene(b).render(rd)
        for k in env:
            monkeypatch.delenv(k)
        assert np.array_equal(base.view(np.uint32), film.view(np.uint32)), env
        assert (pbase.bounce_rays, pbase.shadow_rays, pbase.env_hits) == (prof.bounce_rays, prof.shadow_rays, prof.env_hits), env


@pytest.mark.parametrize("scene,L", [("cornell_gem", 2), ("hdri_small", 3)])
def test_hero_wavelengths_through_every_parked_form(engine, pkg, monkeypatch, scene, L):
    """Four wavelengths per path through the parked kernels: the hybrid form, the same with axis-parallel rays walked, the top-level-tree form
    (PT_AMD_NO_SWEEP) and the per-lane walk (PT_AMD_NO_SWEEP + PT_AMD_NO_PARK) give one film, bit for bit."""
    b = pkg.scene.SCENES[scene]()
    rd = pkg.api.render_desc(160, 128, 8, 6, light_samples=L, seed=21, hero_wavelengths=4)
    monkeypatch.setenv("PT_AMD_BLOCKS_PER_CU", "1")
    base, pbase = engine.create_scene(b).render(rd)
    for env in ({"PT_AMD_NO_AXIS_SCAN": "1"}, {"PT_AMD_NO_SWEEP": "1"}, {"PT_AMD_NO_SWEEP": "1", "PT_AMD_NO_PARK": "1"}, {"PT_AMD_NO_PARK": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        film, prof = engine.create_scene(b).render(rd)
        for k in env:
            monkeypatch.delenv(k)
        assert np.array_equal(base.view(np.uint32), film.view(np.uint32)), env
        assert (pbase.bounce_rays, pbase.shadow_rays, pbase.env_hits) == (prof.bounce_rays, prof.shadow_rays, prof.env_hits), env


@pytest.mark.parametrize("scene,L,hero", [("cornell_box", 2, 1), ("cornell_box", 3, 4), ("disk_lamp", 2, 1), ("white_furnace", 6, 1), ("hdri_small", 3, 1), ("hdri_emissive_mesh", 3, 1)])
def test_forms_without_transforms_change_nothing(engine, pkg, monkeypatch, scene, L, hero):
    """A scene in which no instance carries a transform (the Cornell box) runs kernel forms with the matrix paths compiled out
    (PT_SCENE_NO_XF: k_extend / k_shadow in their sweep forms, the lean k_shade).  The general forms (PT_AMD_GENERAL_FORMS=1) give the same
    film, counters and hits bit for bit; a scene that does hold a transform never gets the lean forms."""
    import parity_suite
    b = pkg.scene.SCENES[scene]()
    rd = pkg.api.render_desc(192, 160, 10, 8, light_samples=L, seed=4, hero_wavelengths=hero)
    o, d = parity_suite.golden_rays(scene, 1 << 14, 5)
    ref = engine.create_scene(b)
    base, pbase = ref.render(rd)
    hits = ref.intersect(o, d)
    for env in ({"PT_AMD_GENERAL_FORMS": "1"}, {"PT_AMD_NO_FUSE": "1"}, {"PT_AMD_NO_FUSE": "1", "PT_AMD_BLOCKS_PER_CU": "1"}, {"PT_AMD_BLOCKS_PER_CU": "2"},
                {"PT_AMD_NO_LIVE_LIST": "1"}, {"PT_AMD_NO_LIVE_LIST": "1", "PT_AMD_NO_FUSE": "1"},   # (every light-sample item read, not the list of those with a live ray)
                {"PT_AMD_NO_ONE_LIGHT": "1"}, {"PT_AMD_NO_ONE_LIGHT": "1", "PT_AMD_NO_LIVE_LIST": "1"},   # (the only light NOT tested at the vertex: the light-sample kernel drops those rays itself)
                {"PT_AMD_BLOCKS_PER_CU": "1", "PT_AMD_GENERAL_FORMS": "1"}):
        # (the fused form — k_shade tracing its own segments, the default for single-wavelength scenes of this kind — against k_extend + k_shade;
        # long segments: many rounds per workgroup)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        other = engine.create_scene(b)
        for k in env:
            monkeypatch.delenv(k)
        film, prof = other.render(rd)
        assert np.array_equal(base.view(np.uint32), film.view(np.uint32)), env
        assert (pbase.bounce_rays, pbase.shadow_rays, pbase.env_hits) == (prof.bounce_rays, prof.shadow_rays, prof.env_hits), env
        assert (prof.kernel_launches[1] == 0) == (scene in ("cornell_box", "disk_lamp") and "PT_AMD_NO_FUSE" not in env and "PT_AMD_GENERAL_FORMS" not in env), env   # (fused with hero wavelengths too since round 4)
        parity_suite.assert_hits_equal(hits, other.intersect(o, d))


@pytest.mark.parametrize("case", ["cornell_gem", "cornell_gem_hero", "fuzz_49682", "fuzz_100003", "fuzz_197995", "cube_sky", "prism_smooth_sky", "test_prism_small", "cube_lit_disk"])
def test_convex_certificates_change_nothing(engine, oracle, pkg, monkeypatch, case):
    """Round 6 (pt_blob.h PT_INST_CONVEX_*): a light-sample ray that leaves a certified closed convex mesh instance inward is dead where it is made, one that leaves it outward
    drops the instance from its leaf mask and never parks at it.  With the certificates ignored (PT_AMD_NO_CONVEX = PT_TUNE_NO_CONVEX) — and through the other traversal
    forms, which do not use the marks — the film and the counters are the same bit for bit, and they are the oracle's.  fuzz_49682: octahedra under a sky that light samples
    pick (an environment ray starts on the side of its direction's world z, pt.rs:256: half of the outward ones start INSIDE the body — the case that caught the first version).
    fuzz_197995, cube_lit_disk: a lit disk whose rim reaches into a glass cube while its reference box (half the radius, disk.rs:24-28) clears the cube's — the soak's find."""
    import fuzz_scenes
    from test_emulation import _cube, _mesh_scene
    if case.startswith("fuzz_"):
        b = fuzz_scenes.random_scene(int(case[5:]))
    elif case == "cube_sky":
        p, f = _cube()
        b = _mesh_scene(pkg, p, f, transform=pkg.scene.transform_from_data(scale=(0.5, 2.0, 1.25), rotate=[((0.3, 1.0, 0.2), 37.0)], translate=(0.2, -0.4, 0.1)), sky=True)
    elif case == "prism_smooth_sky":   # (a smooth-shaded convex body: every face with its own outward threshold)
        b = _mesh_scene(pkg, *pkg.scene._npz_mesh("prism")[:3], transform=pkg.scene.transform_from_data(scale=(2.0, 3.0, 2.5)), sky=True)
    elif case == "test_prism_small":   # (G1's scene: the reference tree's test_prism.toml)
        b = pkg.scene.test_prism_small()
    elif case == "cube_lit_disk":
        p, f = _cube()
        b = _mesh_scene(pkg, p, f, lit_disk=(0.55, (1.3, 0.5, 0.5)))
    else:
        b = pkg.scene.cornell_gem()
    rd = pkg.api.render_desc(192, 160, 10, 10, light_samples=3, seed=12, hero_wavelengths=4 if case.endswith("hero") else 1)
    monkeypatch.setenv("PT_AMD_BLOCKS_PER_CU", "2")   # (segments long enough for the park lists to fill)
    base, pbase = engine.create_scene(b).render(rd)
    for env in ({"PT_AMD_NO_CONVEX": "1"}, {"PT_AMD_NO_SWEEP": "1"}, {"PT_AMD_NO_PARK": "1"}, {"PT_AMD_NO_CONVEX": "1", "PT_AMD_NO_SWEEP": "1"}, {"PT_AMD_PARK_DYNAMIC": "1", "PT_AMD_LDS_ALL_LIMIT": "98304"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        film, prof = engine.create_scene(b).render(rd)
        for k in env:
            monkeypatch.delenv(k)
        assert np.array_equal(base.view(np.uint32), film.view(np.uint32)), env
        assert (pbase.bounce_rays, pbase.shadow_rays, pbase.env_hits) == (prof.bounce_rays, prof.shadow_rays, prof.env_hits), env
    ref, rprof = oracle.create_scene(b).render(rd)
    ps.check_film(base, ref, pbase, rprof)


def test_tuning_is_taken_at_scene_creation(engine, pkg, monkeypatch):
    """pt_tuning: the engine's switches as an explicit struct (what a Rust host sets per scene).  pt_scene_create_tuned with a flag set does what
    the variable does through pt_tuning_default; a variable that changes AFTER the scene exists changes nothing (the environment is read once,
    when the scene is created, never during a render); a struct with reserved words set is refused."""
    b = pkg.scene.cornell_box()
    rd = pkg.api.render_desc(96, 64, 5, 6, seed=2)
    default = engine.create_scene(b)
    film0, p0 = default.render(rd)
    assert p0.kernel_launches[1] == 0            # the fused form: no k_extend launches
    t = engine.tuning_default()
    assert t.flags == 0 and t.batch_slots == 0 and t.park_dynamic == -1
    t.flags |= pkg.api.TUNE_NO_FUSE
    t.blocks_per_cu = 2
    film1, p1 = engine.create_scene(b, t).render(rd)
    assert p1.kernel_launches[1] > 0 and np.array_equal(film0.view(np.uint32), film1.view(np.uint32))
    monkeypatch.setenv("PT_AMD_NO_FUSE", "1")
    assert engine.tuning_default().flags & pkg.api.TUNE_NO_FUSE
    film2, p2 = default.render(rd)               # the scene was created before the variable was set: still fused
    assert p2.kernel_launches[1] == 0 and np.array_equal(film0.view(np.uint32), film2.view(np.uint32))
    film3, p3 = engine.create_scene(b).render(rd)
    assert p3.kernel_launches[1] > 0
    monkeypatch.delenv("PT_AMD_NO_FUSE")
    bad = engine.tuning_default()
    bad.reserved[1] = 1
    with pytest.raises(pkg.api.PtError):
        engine.create_scene(b, bad)
    bad = engine.tuning_default()
    bad.shade_form = 7
    with pytest.raises(pkg.api.PtError):
        engine.create_scene(b, bad)


def test_whole_node_render_from_one_call(engine, pkg):
    """pt_render_multi (the one blocking call a Rust `impl Renderer` makes): every device of the mask renders its tiles on its own
    host thread and stream, the device films are summed with RCCL.  On the one GPU of this box: mask 0 and mask 1 give pt_render's film
    bit for bit, with the RCCL reduce forced (PT_TUNE_MULTI_RCCL: single-process communicator, ncclReduce on the render stream) too;
    a mask that names no visible device and a sharded desc are refused."""
    n = engine.lib.pt_device_count()
    assert n >= 1
    b = pkg.scene.cornell_box()
    sc = engine.create_scene(b)
    rd = pkg.api.render_desc(160, 96, 7, 6, seed=5)
    base, pbase = sc.render(rd)
    for mask in (0, 1):
        film, prof = sc.render_multi(rd, mask)
        assert np.array_equal(film.view(np.uint32), base.view(np.uint32)), mask
        assert (prof.camera_rays, prof.bounce_rays, prof.shadow_rays) == (pbase.camera_rays, pbase.bounce_rays, pbase.shadow_rays)
    t = engine.tuning_default()
    t.flags |= pkg.api.TUNE_MULTI_RCCL
    rc = engine.create_scene(b, t)
    film, prof = rc.render_multi(rd, 1)
    assert np.array_equal(film.view(np.uint32), base.view(np.uint32))
    assert prof.seconds > 0 and prof.camera_rays == pbase.camera_rays
    setup_first = prof.kernel_seconds[5]
    film, prof = rc.render_multi(rd, 1)   # the communicator, the stream and the device film are the scene's now
    assert np.array_equal(film.view(np.uint32), base.view(np.uint32))
    assert prof.kernel_seconds[5] < 1e-3 and prof.kernel_seconds[5] <= setup_first, (setup_first, prof.kernel_seconds[5])
    with pytest.raises(pkg.api.PtError):
        sc.render_multi(rd, 1 << 40)
    with pytest.raises(pkg.api.PtError):
        sc.render_multi(pkg.api.render_desc(160, 96, 7, 6, seed=5, shard=(0, 2)), 0)


def test_whole_node_on_two_physical_devices(engine, pkg):
    """pt_render_multi over two real devices (runs only where the box has them): tiles dealt to both, one RCCL reduce over xGMI, the film
    of pt_render bit for bit; a second call reuses the communicator."""
    if engine.lib.pt_device_count() < 2:
        pytest.skip("one HIP device on this box")
    b = pkg.scene.cornell_box()
    sc = engine.create_scene(b)
    rd = pkg.api.render_desc(256, 160, 9, 6, seed=5)
    base, pbase = sc.render(rd)
    for call in range(2):
        film, prof = sc.render_multi(rd, 0b11)
        assert np.array_equal(film.view(np.uint32), base.view(np.uint32)), call
        assert (prof.camera_rays, prof.bounce_rays, prof.shadow_rays) == (pbase.camera_rays, pbase.bounce_rays, pbase.shadow_rays)
    assert prof.kernel_seconds[5] < 1e-3


@pytest.mark.parametrize("scene,virt,rccl", [("cornell_box", 2, False), ("cornell_box", 8, False), ("cornell_gem", 3, False), ("hdri_small", 4, True)])
def test_whole_node_with_virtual_devices(engine, pkg, scene, virt, rccl):
    """pt_render_multi with N > 1 on the one GPU there is: pt_tuning::multi_virtual = k treats the device as k devices — k host threads,
    k streams, k scene replicas with their own queues, shard_index = 0..k-1 of shard_count = k, each into its own device film — and the
    films are summed on the device (and then, with PT_TUNE_MULTI_RCCL, handed to the RCCL reduce as a node's first device would).  The
    thread-per-device / replica / shard path of a node with N devices, with the reduce between physical devices left out.  The film equals
    pt_render's bit for bit, the counters add up, a second call pays no set-up, and the caller's current device is left as it was."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")   # (the process's HIP runtime, already loaded by the engine: the caller's current device as HIP itself reports it — round-4 advisor:
                                          # torch's view raised when torch's runtime came up after the engine's, and the guard below was skipped)

    def current_device():
        dev = ctypes.c_int(-1)
        assert hip.hipGetDevice(ctypes.byref(dev)) == 0
        return dev.value
    b = pkg.scene.SCENES[scene]()
    rd = pkg.api.render_desc(200, 136, 9, 6, light_samples=3, seed=11)
    base, pbase = engine.create_scene(b).render(rd)
    t = engine.tuning_default()
    t.multi_virtual = virt
    if rccl:
        t.flags |= pkg.api.TUNE_MULTI_RCCL
    sc = engine.create_scene(b, t)
    before = current_device()
    for call in range(2):
        film, prof = sc.render_multi(rd, 1)
        assert np.array_equal(film.view(np.uint32), base.view(np.uint32)), (virt, call)
        assert (prof.camera_rays, prof.bounce_rays, prof.shadow_rays, prof.env_hits) == (pbase.camera_rays, pbase.bounce_rays, pbase.shadow_rays, pbase.env_hits)
        assert prof.kernel_launches[2] == virt * pbase.kernel_launches[2]   # every virtual device ran its own pipeline
        if call == 1:
            assert prof.kernel_seconds[5] < 1e-3, prof.kernel_seconds[5]    # set-up: replicas, streams and films are cached on the scene
    assert current_device() == before


def test_full_size_cornell_properties(engine, oracle, pkg):
    """BASELINE.json C2 geometry (1024x1024, max_bounces 8, L = 2) at 2 spp (2 M paths: seconds for the threaded oracle on
    the GPU box's host): the north-star bar directly — film within 1e-4 L-inf of the oracle at matched seeds, ray counters
    equal — plus size-independent properties: shards partition the film exactly, counters add up, the film is finite and
    non-negative, and a 64x64 box-downsample agrees statistically with an independent 64x64 oracle render."""
    b = pkg.scene.cornell_box()
    sc = engine.create_scene(b)
    rd = pkg.api.render_desc(1024, 1024, 2, 8)
    whole, pw = sc.render(rd)
    assert np.isfinite(whole).all() and whole.min() >= 0 and (whole[..., 3] == 0).all()
    assert pw.camera_rays == 1024 * 1024 * 2
    full_ref, pref = oracle.create_scene(b).render(rd)
    ps.check_film(whole, full_ref, pw, pref, flat=True)
    acc = np.zeros_like(whole); rays = 0
    for k in range(4):
        part, pp = sc.render(pkg.api.render_desc(1024, 1024, 2, 8, shard=(k, 4)))
        acc += part; rays += pp.bounce_rays + pp.shadow_rays
    assert np.array_equal(acc, whole) and rays == pw.bounce_rays + pw.shadow_rays
    ref, _ = oracle.create_scene(b).render(pkg.api.render_desc(64, 64, 64, 8, seed=3))
    down = whole[..., 1].reshape(64, 16, 64, 16).mean(axis=(1, 3))
    rel = abs(down.mean() - ref[..., 1].mean()) / ref[..., 1].mean()
    assert rel < 0.05, rel


def test_c4_frame_dealt_over_eight_shards(engine, pkg):
    """BASELINE.json's 8-GPU configuration as eight shards on one device: the C4 scene at 1024 x 1024, its 32 x 32 tiles dealt along diagonals over shard = (k, 8)
    (PT_TILE_SHARD, include/pt_api.h; the split of src/renderer/tiled.rs:190-277 into independent tile sets).  The eight partial films are disjoint, their sum is the
    one-device film bit for bit, the counters add up, and no shard gets more than 1/8 of the tiles + one per diagonal's remainder."""
    sc = engine.create_scene(pkg.scene.hdri_test())
    whole, pw = sc.render(pkg.api.render_desc(1024, 1024, 2, 4, light_samples=6, seed=8))
    acc = np.zeros_like(whole)
    covered = np.zeros(whole.shape[:2], np.int32)
    rays = [0, 0, 0]
    for k in range(8):
        part, pp = sc.render(pkg.api.render_desc(1024, 1024, 2, 4, light_samples=6, seed=8, shard=(k, 8)))
        owned = (part != 0).any(axis=2)
        tiles = owned.reshape(32, 32, 32, 32).any(axis=(1, 3))          # which 32 x 32 tiles this shard touched
        assert 120 <= tiles.sum() <= 136, (k, tiles.sum())               # 1024 tiles over 8 shards: 128 each, give or take the diagonals' ends
        covered += owned
        acc += part
        rays[0] += pp.bounce_rays; rays[1] += pp.shadow_rays; rays[2] += pp.env_hits
        assert pp.camera_rays == 2 * 1024 * 1024 // 8
    assert covered.max() == 1                                            # disjoint
    assert np.array_equal(acc.view(np.uint32), whole.view(np.uint32))
    assert tuple(rays) == (pw.bounce_rays, pw.shadow_rays, pw.env_hits)


@pytest.mark.parametrize("scene,w,h,mb,kw", [
    ("cornell_box", 1024, 1024, 8, {"spp": 10}),                      # C2, one whole 10-sample phase
    ("cornell_gem", 1920, 1080, 12, {}),                              # C3
    ("hdri_test", 1024, 1024, 4, {"light_samples": 6}),               # C4
    ("cornell_box", 1024, 1024, 8, {"hero_wavelengths": 4}),          # C5
    ("test_bokeh", 1024, 1024, 8, {"light_samples": 2}),              # G2 (round-4 verdict, item 1): the reference tree's scene with more than 64 instances — the top-level BVH walk
    ("test_bokeh_floor", 1024, 1024, 8, {"light_samples": 2}),        # G2F: the same walk under light-sample rays (82-entry light list)
    ("test_prism", 1024, 1024, 8, {"light_samples": 2}),              # G1 (round-3 verdict, item 5): a reference-tree scene that takes the general kernel forms
])
def test_full_size_baseline_configs(engine, oracle, scene, w, h, mb, kw):
    """The other BASELINE.json configurations at their full film size, 1 spp (C2: 10), against the oracle at matched seeds."""
    kw = dict(kw)
    # north_star's bar as stated: L-inf < 1e-4 on every pixel, flat — no BASELINE configuration may need the 8-ulp allowance for bright pixels
    ps.render_parity(engine, oracle, scene, w, h, kw.pop("spp", 1), mb, flat=True, **kw)


def test_error_behaviour(engine, pkg):
    b = pkg.scene.cornell_box()
    sc = engine.create_scene(b)
    for bad in (dict(camera_index=3), dict(shard=(2, 2)), dict(hero_wavelengths=3), dict(light_samples=9)):
        with pytest.raises(pkg.api.PtError) as e:
            sc.render(pkg.api.render_desc(8, 8, 1, 2, **bad))
        assert e.value.status == 1 and str(e.value)
    bad_scene = pkg.scene.SceneBuilder()
    bad_scene.add_camera((0, 0, 0), (1, 0, 0), 40.0)
    with pytest.raises(pkg.api.PtError):
        engine.create_scene(bad_scene)   # environment curve missing


def test_bench_line_contract(pkg):
    """bench.py prints ONE JSON line with the driver's keys, the roofline of the dominant kernel (HIP-event time inside the
    timed region) and the CPU baseline of the oracle."""
    import json
    import subprocess
    import sys
    root = pkg.REPO_ROOT
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--width", "128", "--height", "128",
                        "--spp-per-step", "20", "--cpu-seconds", "0.5"], capture_output=True, text=True, cwd=root, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().split("\n") if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"].startswith("Msamples/s") and d["unit"] == "Msamples/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 128 * 128 * 20 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-6
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert "traffic" in rf and rf["kernel"].startswith("k_")
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "Msamples/s" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb


@pytest.mark.parametrize("case", ["many_analytic", "many_analytic_no_disk", "many_analytic_300"])
def test_many_analytic_instances(engine, oracle, pkg, monkeypatch, case):
    """More than 64 instances of every analytic kind, none a mesh (the emulation tier's case of the same name): the top-level walk in the parked kernels, through the per-lane
    walk kernels (PT_AMD_NO_PARK), with the light pre-pass forced on (bounded searches that stop at the first opaque hit), without culling, with the exact box test and with the
    blob in HBM: one film and one set of counters, bit for bit — and the oracle's."""
    from test_emulation import many_analytic_scene
    b = {"many_analytic": lambda: many_analytic_scene(pkg), "many_analytic_no_disk": lambda: many_analytic_scene(pkg, 90, 8, disks=False),
         "many_analytic_300": lambda: many_analytic_scene(pkg, 300, 11)}[case]()
    rd = pkg.api.render_desc(192, 160, 8, 8, light_samples=3, seed=9, hero_wavelengths=4 if case == "many_analytic_no_disk" else 1)
    monkeypatch.setenv("PT_AMD_BLOCKS_PER_CU", "2")
    base, pbase = engine.create_scene(b).render(rd)
    for env in ({"PT_AMD_NO_PARK": "1"}, {"PT_AMD_LIGHT_PREPASS_MAX": "4294967295"}, {"PT_AMD_NO_CULL": "1"}, {"PT_AMD_EXACT_SLAB": "1"}, {"PT_AMD_NO_LDS": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        film, prof = engine.create_scene(b).render(rd)
        for k in env:
            monkeypatch.delenv(k)
        assert np.array_equal(base.view(np.uint32), film.view(np.uint32)), env
        assert (pbase.bounce_rays, pbase.shadow_rays, pbase.env_hits) == (prof.bounce_rays, prof.shadow_rays, prof.env_hits), env
    ref, rprof = oracle.create_scene(b).render(rd)
    ps.check_film(base, ref, pbase, rprof)


@pytest.mark.parametrize("scene,L", [("cornell_gem", 2), ("hdri_c4_small", 6), ("test_prism_small", 2), ("test_bokeh_floor_gem_small", 3)])
def test_mesh_shortcuts_change_nothing(engine, pkg, monkeypatch, scene, L):
    """mesh_surely_blocks (a closed mesh's inner balls) and mesh_surely_missed (its 18-DOP slabs) decide a ray at a mesh without the triangle tests the reference runs, on f32
    error budgets (round-5 verdict, "what's weak" 11).  PT_AMD_NO_MESH_SHORTCUTS = PT_TUNE_NO_MESH_SHORTCUTS takes both away: the same film and counters, bit for bit —
    every decision of theirs is checked, not only those that move the film by more than the parity bar; tools/fuzz_soak.py runs the same comparison over random scenes."""
    b = pkg.scene.SCENES[scene]()
    rd = pkg.api.render_desc(192, 160, 8, 8, light_samples=L, seed=17)
    monkeypatch.setenv("PT_AMD_BLOCKS_PER_CU", "2")
    base, pbase = engine.create_scene(b).render(rd)
    for env in ({"PT_AMD_NO_MESH_SHORTCUTS": "1"}, {"PT_AMD_NO_MESH_SHORTCUTS": "1", "PT_AMD_NO_CONVEX": "1"}, {"PT_AMD_NO_MESH_SHORTCUTS": "1", "PT_AMD_NO_SWEEP": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        film, prof = engine.create_scene(b).render(rd)
        for k in env:
            monkeypatch.delenv(k)
        assert np.array_equal(base.view(np.uint32), film.view(np.uint32)), env
        assert (pbase.bounce_rays, pbase.shadow_rays, pbase.env_hits) == (prof.bounce_rays, prof.shadow_rays, prof.env_hits), env
