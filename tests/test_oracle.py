"""Pins the CPU oracle: the numeric contract, the restated `math` crate pieces, and every property the
reference's own tests assert on this path.  CPU only."""
import ctypes as C

import numpy as np
import pytest

from util import film_metrics, fptr, numerics, unit_sphere


# ------------------------------------------------------------------ numeric contract
def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32 10 rounds
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, expect in kat:
        c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
        oracle.lib.ptref_philox(c, k, o)
        assert tuple(o) == expect


def test_uniforms_in_unit_interval(oracle):
    out = (C.c_float * 4)()
    vals = []
    for pixel in range(64):
        for dim in range(8):
            oracle.lib.ptref_draw4(1, pixel, 3, dim, out)
            vals.extend(out)
    vals = np.array(vals)
    assert vals.min() >= 0.0 and vals.max() < 1.0
    assert abs(vals.mean() - 0.5) < 0.03
    assert len(np.unique(vals)) == len(vals)


def test_elementary_functions(oracle):
    rng = np.random.default_rng(0)
    x = rng.uniform(-7, 7, 20000).astype(np.float32)
    assert np.abs(numerics(oracle, 0, x) - np.sin(x.astype(np.float64))).max() < 2.5e-7
    assert np.abs(numerics(oracle, 1, x) - np.cos(x.astype(np.float64))).max() < 2.5e-7
    x = rng.uniform(-80, 80, 20000).astype(np.float32)
    e = numerics(oracle, 2, x)
    assert (np.abs(e / np.exp(x.astype(np.float64)) - 1)).max() < 3e-7
    assert numerics(oracle, 2, np.array([-200.0], np.float32))[0] == 0.0
    assert np.isinf(numerics(oracle, 2, np.array([100.0], np.float32))[0])
    b = rng.uniform(0, 1, 20000).astype(np.float32); p = rng.uniform(1, 500, 20000).astype(np.float32)
    pw = numerics(oracle, 3, b, p)
    ref = np.power(b.astype(np.float64), p.astype(np.float64))
    ok = ref > 1e-30
    assert (np.abs(pw[ok] / ref[ok] - 1)).max() < 2e-7
    x = rng.uniform(-1, 1, 20000).astype(np.float32)
    assert np.abs(numerics(oracle, 4, x) - np.arccos(x.astype(np.float64))).max() < 6e-7
    y = rng.uniform(-1, 1, 20000).astype(np.float32)
    assert np.abs(numerics(oracle, 5, y, x) - np.arctan2(y.astype(np.float64), x.astype(np.float64))).max() < 6e-7
    x = rng.uniform(-30, 30, 5000).astype(np.float32)
    assert (np.abs(numerics(oracle, 6, x).astype(np.float64) / np.exp(x.astype(np.float64)) - 1)).max() < 1.3e-7
    x = rng.uniform(1e-6, 1e6, 5000).astype(np.float32)
    assert np.abs(numerics(oracle, 7, x) - np.log(x.astype(np.float64))).max() < 1e-6


def numerics_pin_inputs():
    rng = np.random.default_rng(20261004)
    bits = rng.integers(0, 2**32, 40000, dtype=np.uint64).astype(np.uint32).view(np.float32)      # any bit pattern: NaNs, infinities, subnormals
    wide = rng.uniform(-110, 95, 40000).astype(np.float32)
    unit = rng.uniform(-1.01, 1.01, 40000).astype(np.float32)
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 88.7, 88.73, -87.3, -103.0, -104.0, 1e-45, 2.4142137, 0.41421357, 0.5, 2.0, 1e30, 709.0, 1e-38], np.float32)
    x = np.concatenate([bits, wide, unit, np.repeat(special, 20)])
    y = np.concatenate([np.roll(bits, 7), rng.uniform(-3, 3, 40000).astype(np.float32), np.roll(unit, 11), np.tile(special, 20)])
    return x, y


def numerics_pin_crc(out):
    import zlib
    u = out.view(np.uint32).copy()
    u[np.isnan(out)] = 0x7fc00000        # (which NaN comes back for a NaN — a signalling one quieted or not — is the compiler's choice of conversions, not the routine's)
    return zlib.crc32(u.tobytes())


# CRC-32 of the output bits of pt_sin, pt_cos, pt_exp, pt_pow, pt_acos, pt_atan2, (f32) pt_exp64, (f32) pt_log64 over numerics_pin_inputs(): taken from the
# routines as they were before round 4 rewrote them as data flow (special cases selected at the end instead of early returns), and unchanged by the rewrite
NUMERICS_PINS = {0: 890435962, 1: 3703979741, 2: 3553140640, 3: 2260303302, 4: 2422965045, 5: 2819024569, 6: 2657817569, 7: 3676543608}


def test_elementary_functions_keep_their_bits(oracle):
    """include/pt_numerics.h is the numeric contract of engine and oracle: a change of any output bit of its routines moves every film."""
    x, y = numerics_pin_inputs()
    for which, crc in NUMERICS_PINS.items():
        assert numerics_pin_crc(numerics(oracle, which, x, y)) == crc, which


def test_cie_fit(oracle):
    """Wyman-Sloan-Shirley fit: y_bar peaks ~1 near 555 nm; x,z lobes in the right places."""
    out = (C.c_float * 3)()
    oracle.lib.ptref_xyz_bar(555.0, out)
    assert abs(out[1] - 1.0) < 0.02 and 0.45 < out[0] < 0.56 and out[2] < 0.02
    oracle.lib.ptref_xyz_bar(450.0, out)
    assert out[2] > 1.5 and out[1] < 0.1
    oracle.lib.ptref_xyz_bar(600.0, out)
    assert out[0] > 1.0 and out[2] < 0.01


# ------------------------------------------------------------------ renderer plumbing
def test_generate_tiles_cover_film(oracle):
    """src/renderer/tiled.rs:677-689: 64x64 tiles over 1920x1080 cover every pixel exactly once."""
    n = C.c_uint32()
    oracle.lib.ptref_generate_tiles(1920, 1080, 64, 64, None, C.byref(n))
    tiles = (C.c_uint32 * (4 * n.value))()
    oracle.lib.ptref_generate_tiles(1920, 1080, 64, 64, tiles, C.byref(n))
    t = np.array(tiles).reshape(-1, 4)
    cover = np.zeros((1080, 1920), np.int32)
    for x0, x1, y0, y1 in t:
        cover[y0:y1, x0:x1] += 1
    assert (cover == 1).all()
    # order: full tiles row-major, right remnants, bottom remnants, corner
    assert tuple(t[0]) == (0, 64, 0, 64) and tuple(t[-1]) == (1920 - 1920 % 64 if 1920 % 64 else 1856, 1920, 1024, 1080)


# ------------------------------------------------------------------ curves
def test_curves(pkg, oracle):
    b = pkg.scene.SceneBuilder()
    tab = b.curve_tabulated("t", [400, 500, 600, 700], [0.0, 8.0, 15.6, 18.4])
    cau = b.curve_cauchy("c", 1.5, 10000.0)
    flat = b.curve_flat("f", 0.78)
    bb = b.curve_blackbody("bb", 5000.0, 2.0)
    spike = b.curve_simple_spike("s", 500.0, 100.0, 50.0, 0.55)
    lin = b.curve_linear("l", 390.0, 10.0, [1.0, 2.0, 4.0], mode=pkg.api.INTERP_LINEAR)
    b.set_environment_constant(flat, 0.0)
    b.add_camera((0, 0, 0), (1, 0, 0), 40.0)
    sc = oracle.create_scene(b)
    lam = np.array([380, 400, 450, 500, 600, 650, 700, 750], np.float32)
    v = sc.curve_eval(tab, lam)
    assert v[0] == 0.0 and v[1] == 0.0 and v[3] == 8.0 and v[4] == np.float32(15.6) and v[6] == np.float32(18.4) and v[7] == np.float32(18.4)
    assert v[2] == 4.0  # zero-tangent Hermite is symmetric at the midpoint
    assert abs(v[5] - 17.0) < 1e-5
    assert np.allclose(sc.curve_eval(cau, lam), 1.5 + 10000.0 / lam.astype(np.float64) ** 2, rtol=1e-6)
    assert (sc.curve_eval(flat, lam) == np.float32(0.78)).all()
    assert (sc.curve_eval(flat, np.array([369.0, 791.0], np.float32)) == 0).all()
    peak = 2.8977721e-3 / (5000.0 * 1e-9)
    assert abs(sc.curve_eval(bb, np.array([peak], np.float32))[0] - 2.0) < 1e-4
    assert (sc.curve_eval(bb, lam) < 2.0 + 1e-5).all() and (sc.curve_eval(bb, lam) > 0.5).all()
    s = sc.curve_eval(spike, np.array([500.0, 400.0, 550.0], np.float32))
    assert abs(s[0] - 0.55) < 1e-7 and abs(s[1] - 0.55 * np.exp(-0.5)) < 1e-6 and abs(s[2] - 0.55 * np.exp(-0.5)) < 1e-6
    l = sc.curve_eval(lin, np.array([389.0, 390.0, 395.0, 400.0, 405.0, 415.0, 425.0], np.float32))
    assert l[0] == 0 and l[1] == 1.0 and l[2] == 1.5 and l[3] == 2.0 and l[4] == 3.0 and l[5] == 4.0 and l[6] == 0.0


# ------------------------------------------------------------------ materials
def _ggx_glass_scene(pkg, roughnesses):
    """ggx_glass(roughness) of the reference's tests: cauchy(1.5, 1e4), eta_o = cie_e(1), kappa = void (ggx.rs:630-635)."""
    b = pkg.scene.SceneBuilder()
    glass = b.curve_cauchy("glass", 1.5, 10000.0)
    one = b.curve_flat("one", 1.0)
    zero = b.curve_flat("zero", 0.0)
    ids = [pkg.api.MATERIAL_NONE and (b.material_ggx("g%d" % i, float(r), glass, one, zero) & 0xFFFF) for i, r in enumerate(roughnesses)]
    b.set_environment_constant(zero, 0.0)
    b.add_camera((0, 0, 0), (1, 0, 0), 40.0)
    return b, ids


def ggx_property_cases(make_scene, n_rough=40, per=50, seed=7):
    rng = np.random.default_rng(seed)
    x = rng.random(n_rough, dtype=np.float32)
    rough = 1.0 / (-np.log(x + np.float32(1.1920929e-7)))   # src/props.rs:10-14
    rough = np.concatenate([rough, [8.736748]]).astype(np.float32)
    return rough, rng, per


def check_ggx_properties(pkg, lib):
    """Replays test_ggx / test_ggx2 (ggx.rs:637-756).  lambda is drawn from [400, 790): the test's eta_o is
    cie_e(1.0), which is 0 outside EXTENDED_VISIBLE_RANGE = [370, 790], so 790..800 divides by zero.
    The strict positivity of test_ggx is asserted where generate_and_evaluate and bsdf agree on the lobe;
    the GGX of the reference lets a steep microfacet "reflect" wo through the macro surface (did_reflect with
    wi.z * wo.z < 0, ggx.rs:428-447), bsdf() then evaluates the transmission lobe for that pair and the
    swapped evaluation can be exactly 0 (total internal reflection) — a property of the reference's model that a
    faithful restatement inherits (measured here: 0.4 % of samples at alpha = 0.2, 7 % at alpha = 1)."""
    rough, rng, per = ggx_property_cases(None)
    b, ids = _ggx_glass_scene(pkg, rough)
    sc = lib.create_scene(b)
    for mid in ids:
        wi = unit_sphere(rng, per)
        wo2 = unit_sphere(rng, per)
        lam = rng.uniform(400, 790, per).astype(np.float32)
        s = rng.random((per, 2), dtype=np.float32)
        f, wo, pdf = sc.bsdf_sample(mid, lam, wi, s)             # test_ggx (ggx.rs:637-683)
        assert np.isfinite(wo).all() and (f > 0).all() and (pdf > 0).all()
        assert np.allclose(np.linalg.norm(wo, axis=1), 1.0, atol=1e-4)
        f0, p0 = sc.bsdf_eval(mid, lam, wi, wo)
        f1, p1 = sc.bsdf_eval(mid, lam, wo, wi)
        assert (f0 >= 0).all() and (p0 >= 0).all() and (f1 >= 0).all() and (p1 >= 0).all()
        same_lobe = np.abs(f - f0) <= 2e-3 * np.maximum(f, f0)
        assert same_lobe.mean() > 0.5
        assert (f0[same_lobe] > 0).all() and (f1[same_lobe] > 0).all()
        g0, q0 = sc.bsdf_eval(mid, lam, wi, wo2)                 # test_ggx2 (ggx.rs:685-756)
        g1, q1 = sc.bsdf_eval(mid, lam, wo2, wi)
        assert (g0 >= 0).all() and (q0 >= 0).all() and (g1 >= 0).all() and (q1 >= 0).all()
    # regression seed, proptest-regressions/materials/ggx.txt:7
    mid = ids[-1]
    wi = np.array([[0.54826164, 0.0, -0.83630687]], np.float32); lam = np.array([400.0], np.float32)
    f, wo, pdf = sc.bsdf_sample(mid, lam, wi, np.zeros((1, 2), np.float32))
    f0, p0 = sc.bsdf_eval(mid, lam, wi, wo); f1, p1 = sc.bsdf_eval(mid, lam, wo, wi)
    # proptest recorded this input because the strict property FAILS on it; the restatement reproduces the
    # mechanism: s.x = 0 <= refl_prob forces did_reflect, alpha = 8.7 tilts wh so far that wo crosses the surface.
    assert wi[0, 2] * wo[0, 2] < 0 and f[0] > 0 and pdf[0] > 0
    assert f0[0] > 0 and p0[0] >= 0 and f1[0] >= 0 and p1[0] >= 0
    # fixed pair of test_ggx_functions (ggx.rs:825-826) at alpha = 0.001
    b2, ids2 = _ggx_glass_scene(pkg, [0.001])
    sc2 = lib.create_scene(b2)
    wi = np.array([[0.9709351, 0.18724124, 0.14908342]], np.float32)
    wo = np.array([[-0.008856451, 0.6295874, -0.7768792]], np.float32)
    f, p = sc2.bsdf_eval(ids2[0], np.array([500.0], np.float32), wi, wo)
    assert f[0] >= 0 and p[0] >= 0
    return sc


def test_ggx_reference_properties(pkg, oracle):
    check_ggx_properties(pkg, oracle)


def test_ggx_sample_matches_eval(pkg, oracle):
    """generate_and_evaluate and bsdf agree on (f, pdf) for a rough dielectric when wh is recoverable."""
    b, ids = _ggx_glass_scene(pkg, [0.3])
    sc = oracle.create_scene(b)
    rng = np.random.default_rng(3)
    n = 20000
    wi = unit_sphere(rng, n); lam = rng.uniform(400, 750, n).astype(np.float32); s = rng.random((n, 2), dtype=np.float32)
    f, wo, pdf = sc.bsdf_sample(ids[0], lam, wi, s)
    f2, pdf2 = sc.bsdf_eval(ids[0], lam, wi, wo)
    # outside reflections only: for wi.z < 0 generate evaluates Fresnel at wi.wh > 0 (flipped wh, ggx.rs:171-180,
    # 457-458) while bsdf evaluates it at wi.wh < 0 (ggx.rs:286-292) — they disagree by construction.
    refl = (wi[:, 2] > 0) & (wo[:, 2] > 0)
    assert refl.sum() > 300
    assert np.allclose(f[refl], f2[refl], rtol=2e-3, atol=1e-6)
    # pdfs are NOT compared: the lobe-mixture weight uses Fresnel(wi.z) in bsdf (ggx.rs:388-391) but Fresnel(wi.wh)
    # in generate_and_evaluate (ggx.rs:557) — reference quirk, SURVEY §8(a) a16.


def test_sharp_light_pdf_integrates_to_one(pkg, oracle):
    """src/materials/sharp_light.rs:229-...: (n+1)|cos|^n / 2pi integrates to 1 over the hemisphere."""
    b = pkg.scene.SceneBuilder()
    one = b.curve_flat("one", 1.0)
    zero = b.curve_flat("zero", 0.0)
    b.set_environment_constant(zero, 0.0)
    for sharp in (0.0, 1.5, 40.0, 400.0):
        mid = b.material_sharp_light("s%g" % sharp, one, zero, sharp, pkg.api.SIDED_DUAL) & 0xFFFF
        b.add_camera((0, 0, 0), (1, 0, 0), 40.0)
        sc = oracle.create_scene(b)
        n = 1 + abs(sharp)
        m = 200000
        z = (np.arange(m) + 0.5) / m
        wi = np.stack([np.sqrt(1 - z * z), np.zeros(m), z], axis=1).astype(np.float32)
        e = sc.emission(mid, np.full(m, 550.0, np.float32), wi).astype(np.float64)
        integral = e.mean() * 2 * np.pi   # dw = 2 pi dz
        assert abs(integral - 1.0) < 2e-3, (sharp, n, integral)


def test_lambertian_and_light(pkg, oracle):
    b = pkg.scene.cornell_box()
    sc = oracle.create_scene(b)
    white = b.material("lambertian_white") & 0xFFFF
    light = b.material("diffuse_light_cornell") & 0xFFFF
    lam = np.array([400.0, 550.0, 700.0], np.float32)
    wi = np.tile(np.array([[0.0, 0.6, 0.8]], np.float32), (3, 1))
    f, wo, pdf = sc.bsdf_sample(white, lam, wi, np.array([[0.1, 0.2], [0.5, 0.5], [0.9, 0.7]], np.float32))
    assert abs(f[0] * np.pi - 0.343) < 1e-6 and abs(f[2] * np.pi - 0.737) < 1e-6  # knots of the cornell white table
    assert 0.70 < f[1] * np.pi < 0.76
    assert np.allclose(pdf, np.abs(wo[:, 2]) / np.pi, rtol=1e-6) and (wo[:, 2] > 0).all()
    f2, pdf2 = sc.bsdf_eval(white, lam, wi, wo)
    assert np.array_equal(f, f2) and np.array_equal(pdf, pdf2)
    f3, pdf3 = sc.bsdf_eval(white, lam, wi, -wo)
    assert (f3 == 0).all() and (pdf3 == 0).all()
    # Reverse-sided: emits only for wi.z < 0 (diffuse_light.rs:123-133); cornell_light(500) = 8.0
    up = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (3, 1))
    assert (sc.emission(light, lam, up) == 0).all()
    e = sc.emission(light, np.array([500.0, 700.0, 390.0], np.float32), -up)
    assert np.allclose(e, np.array([8.0, 18.4, 0.0]) / np.pi, rtol=1e-6)
    assert (sc.emission(white, lam, up) == 0).all()


# ------------------------------------------------------------------ intersection
def test_world_intersection(pkg, oracle):
    """Analogue of src/world/mod.rs:268-292 (a ray aimed at the scene must hit) + closest-hit semantics."""
    b = pkg.scene.cornell_box()
    sc = oracle.create_scene(b)
    o = np.array([[-0.8, 0.1, 0.5], [0.3, 0.278, 0.45], [0.3, 0.278, 0.45], [-0.8, 0.278, 0.273]], np.float32)
    d = np.array([[1, 0, 0], [0, 0, 1], [0, 0, -1], [-1, 0, 0]], np.float32)
    h = sc.intersect(o, d)
    assert h["valid"].tolist() == [1, 1, 1, 0]
    assert abs(h["t"][0] - (0.8 + 0.5592)) < 1e-5       # back wall
    assert h["instance"][1] == 0 and abs(h["t"][1] - 0.0987) < 1e-6 and (h["material"][1] >> 16) == pkg.api.TAG_LIGHT
    assert np.allclose(h["normal"][1], [0, 0, 1])        # one-sided rect keeps +Z
    assert abs(h["t"][2] - 0.12) < 1e-6 and np.allclose(np.abs(h["normal"][2]), [0, 0, 1])
    # uv of the light rect
    assert np.allclose(h["uv"][1], [(0.3 - 0.278 + 0.0525) / 0.105, (0.278 - 0.2795 + 0.065) / 0.13], atol=1e-5)


def test_intersection_against_brute_force(pkg, oracle):
    """BVH traversal == exhaustive closest hit: random rays in the mixed scene vs a numpy brute force over
    the transformed gem triangles (Moller-Trumbore) — checks transforms, mesh BVH and instance dispatch."""
    b = pkg.scene.mixed_primitives()
    sc = oracle.create_scene(b)
    rng = np.random.default_rng(5)
    n = 4000
    o = np.tile(np.array([[-3.0, 0.2, 0.6]], np.float32), (n, 1)) + rng.normal(0, 0.05, (n, 3)).astype(np.float32)
    target = np.array([-0.8, 0.0, -0.5]) + rng.normal(0, 0.35, (n, 3))
    d = (target - o); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    h = sc.intersect(o, d)
    gem_inst = len(b.instances) - 1
    inst = b.instances[gem_inst]
    fwd = np.array(list(inst.forward), np.float64).reshape(4, 4)
    mesh = b.meshes[inst.mesh]
    v = np.array(b.vertices, np.float64).reshape(-1, 3)[mesh.vertex_offset:mesh.vertex_offset + mesh.vertex_count]
    f = np.array(b.indices, np.int64)[mesh.index_offset:mesh.index_offset + 3 * mesh.face_count].reshape(-1, 3)
    vw = v @ fwd[:3, :3].T + fwd[:3, 3]
    p0, p1, p2 = vw[f[:, 0]], vw[f[:, 1]], vw[f[:, 2]]
    tbest = np.full(n, np.inf)
    od, dd = o.astype(np.float64), d.astype(np.float64)
    e1, e2 = p1 - p0, p2 - p0
    for i in range(n):
        pv = np.cross(dd[i], e2); det = (e1 * pv).sum(1)
        ok = np.abs(det) > 1e-12
        inv = np.where(ok, 1.0 / np.where(ok, det, 1), 0)
        tv = od[i] - p0; u = (tv * pv).sum(1) * inv
        qv = np.cross(tv, e1); vv = (qv * dd[i]).sum(1) * inv
        t = (e2 * qv).sum(1) * inv
        m = ok & (u >= 0) & (vv >= 0) & (u + vv <= 1) & (t > 1e-9)
        if m.any():
            tbest[i] = t[m].min()
    hit_gem = h["valid"].astype(bool) & (h["instance"] == gem_inst)
    expect_gem = np.isfinite(tbest)
    # where the oracle says gem, brute force agrees on t; where brute force sees the gem first, oracle's t is not larger
    assert hit_gem.sum() > 200
    assert np.allclose(h["t"][hit_gem], tbest[hit_gem], rtol=1e-4, atol=1e-5)
    closer_other = h["valid"].astype(bool) & ~hit_gem & expect_gem
    assert (h["t"][closer_other] <= tbest[closer_other] * (1 + 1e-4)).all()
    assert not (expect_gem & ~h["valid"].astype(bool)).any()
    assert np.allclose(np.linalg.norm(h["normal"][h["valid"] == 1], axis=1), 1.0, atol=1e-5)


# ------------------------------------------------------------------ whole renders
def test_cornell_render_is_deterministic_and_thread_independent(pkg, oracle):
    import oracle_loader
    sc = oracle.create_scene(pkg.scene.cornell_box())
    rd = pkg.api.render_desc(48, 40, 5, 4, tile=(32, 32))
    a, pa = oracle_loader.render_mt(oracle, sc, rd, 1)
    b, pb = oracle_loader.render_mt(oracle, sc, rd, 4)
    assert np.array_equal(a, b)
    assert pa.camera_rays == 48 * 40 * 5 == pb.camera_rays and pa.bounce_rays == pb.bounce_rays and pa.shadow_rays == pb.shadow_rays
    assert np.isfinite(a).all() and (a[..., 3] == 0).all() and a[..., :3].min() >= 0 and a[..., 1].mean() > 1e-3
    rd2 = pkg.api.render_desc(48, 40, 5, 4, tile=(32, 32), seed=2)
    c, _ = oracle_loader.render_mt(oracle, sc, rd2, 4)
    assert not np.array_equal(a, c)


def test_shards_partition_the_film(pkg, oracle):
    """Film tiles dealt to shards (PT_TILE_SHARD): the shard films are disjoint and sum to the whole film exactly."""
    sc = oracle.create_scene(pkg.scene.cornell_box())
    whole, _ = sc.render(pkg.api.render_desc(70, 50, 3, 3, tile=(16, 16)))
    acc = np.zeros_like(whole)
    for k in range(3):
        part, _ = sc.render(pkg.api.render_desc(70, 50, 3, 3, tile=(16, 16), shard=(k, 3)))
        assert ((part != 0) & (acc != 0)).sum() == 0
        acc += part
    assert np.array_equal(acc, whole)


def test_sample_ranges_compose(pkg, oracle):
    """Rendering [0,10) and [10,20) separately and summing equals the 20-spp render (phases of 10, tiled.rs:347-361)."""
    sc = oracle.create_scene(pkg.scene.cornell_box())
    whole, _ = sc.render(pkg.api.render_desc(32, 32, 20, 4))
    a, _ = sc.render(pkg.api.render_desc(32, 32, 20, 4, first_sample=0, sample_count=10))
    b, _ = sc.render(pkg.api.render_desc(32, 32, 20, 4, first_sample=10, sample_count=10))
    assert np.array_equal((a + b) / np.float32(20), whole)


def test_white_furnace(pkg, oracle, cmf=None):
    """data/scenes/white_furnace.toml + config_test_whitefurnace.toml: camera inside a non-absorbing rough glass
    sphere in a constant environment -> the film is spatially uniform (up to MC noise)."""
    b = pkg.scene.white_furnace("ggx_glass_rough")
    sc = oracle.create_scene(b)
    rd = pkg.api.render_desc(24, 24, 256, 8, light_samples=6)
    film, prof = sc.render(rd)
    y = film[..., 1].astype(np.float64)
    assert np.isfinite(film).all() and y.min() > 0
    blocks = y.reshape(4, 6, 4, 6).mean(axis=(1, 3))
    assert blocks.std() / blocks.mean() < 0.08
    # expected level: env radiance integrated against y_bar over the sampled wavelengths, x k in {1, eta^2}
    lam = np.linspace(380, 750, 2000).astype(np.float32)
    env = sc.curve_eval(b.curve("simple_sky_blue"), lam).astype(np.float64)
    xyz = np.zeros((lam.size, 3), np.float32)
    out = (C.c_float * 3)()
    for i, l in enumerate(lam):
        (cmf or oracle).lib.ptref_xyz_bar(float(l), out); xyz[i] = out[:]
    level = (env * xyz[:, 1]).mean()
    ratio = y.mean() / level
    assert 0.5 < ratio < 3.0, ratio


def test_hero_wavelengths_follow_the_single_wavelength_path(pkg, oracle):
    """C5 as defined in oracle/ptref.cpp: the hero wavelength takes every decision, so ray counters equal the
    single-wavelength render of the same seed, and the film agrees statistically (same expectation)."""
    sc = oracle.create_scene(pkg.scene.cornell_box())
    a, pa = sc.render(pkg.api.render_desc(32, 32, 64, 6))
    b, pb = sc.render(pkg.api.render_desc(32, 32, 64, 6, hero_wavelengths=4))
    assert (pa.bounce_rays, pa.shadow_rays, pa.env_hits) == (pb.bounce_rays, pb.shadow_rays, pb.env_hits)
    assert np.isfinite(b).all() and b[..., :3].min() >= 0
    assert abs(a[..., 1].mean() - b[..., 1].mean()) / a[..., 1].mean() < 0.05
    # colour noise drops: per-pixel chroma variance of the hero render is lower
    def chroma_var(f):
        s = f[..., :3].sum(axis=2) + 1e-9
        return np.var(f[..., 0] / s) + np.var(f[..., 2] / s)
    assert chroma_var(b) < chroma_var(a)


def test_panorama_camera_directions(pkg, oracle):
    """PanoramaCamera::get_ray (src/camera/panorama_camera.rs:71-95): film u is the azimuth about the centre line, v the
    elevation from the horizon (top of the film looks up); no aperture.  Checked through a render of a constant
    environment split by a ground rect: pixels above the horizon see the environment, pixels below see the (black) ground."""
    b = pkg.scene.SceneBuilder()
    pkg.scene.add_library_curves(b, ["flat_one", "flat_zero"])
    b.set_environment_constant(b.curve("flat_one"), 1.0)
    b.env_sampling_probability = 0.5
    black = b.material_lambertian("black", b.texstack_texture1("black", b.curve("flat_zero")))
    b.add_rect((1000.0, 1000.0), (0.0, 0.0, -1.0), "Z", True, black)
    b.add_panorama_camera((0.0, 0.0, 0.0), (1.0, 0.0, 0.0), (360.0, 180.0))
    film, prof = oracle.create_scene(b).render(pkg.api.render_desc(64, 32, 4, 2, light_samples=0))
    y = film[..., 1]
    assert (y[:15] > 0).all() and (y[17:] == 0).all()                 # v < 0.5 looks up (angle_y = span_y * (0.5 - v))
    assert 0.1 < y[:15].mean() < 0.6                                   # spectral noise only (Y of a flat spectrum over random wavelengths)
    # a narrow field of view looking straight down sees only the ground
    b.cameras.clear(); b.add_panorama_camera((0.0, 0.0, 0.0), (0.0, 0.0, -1.0), (10.0, 10.0), v_up=(1.0, 0.0, 0.0))
    film, _ = oracle.create_scene(b).render(pkg.api.render_desc(8, 8, 2, 2, light_samples=0))
    assert (film[..., :3] == 0).all()


def test_reference_instance_case(pkg, oracle):
    """src/geometry/instance.rs test_instance (prints only in the reference): a sphere of radius 2 and a 4x4 Z rect, each
    under from_stack(scale 3, Rx(1 rad) * Ry(1 rad), no translation), hit by the ray (0, 0, 10) -> -Z.  Expected values from
    first principles: the scaled sphere has radius 6 (t = 4, normal +Z); the rect's plane passes through the origin (t = 10)
    with normal +-(Rx Ry) e_z."""
    S = pkg.scene
    deg = 180.0 / np.pi
    tr = S.transform_from_data(scale=(3.0, 3.0, 3.0), rotate=[((0.0, 1.0, 0.0), deg), ((1.0, 0.0, 0.0), deg)])   # list order: Ry first, then Rx
    rx = np.array([[1, 0, 0], [0, np.cos(1), -np.sin(1)], [0, np.sin(1), np.cos(1)]])
    ry = np.array([[np.cos(1), 0, np.sin(1)], [0, 1, 0], [-np.sin(1), 0, np.cos(1)]])
    assert np.allclose(tr[:3, :3], (rx @ ry) * 3.0, atol=1e-6)
    o = np.array([[0.0, 0.0, 10.0]], np.float32); d = np.array([[0.0, 0.0, -1.0]], np.float32)
    for shape in ("sphere", "rect"):
        b = S.SceneBuilder()
        white = S.add_library_material(b, "lambertian_white")
        S.add_library_curves(b, ["flat_zero"])
        b.set_environment_constant(b.curve("flat_zero"), 0.0)
        if shape == "sphere":
            b.add_sphere(2.0, (0.0, 0.0, 0.0), white, tr)
        else:
            b.add_rect((4.0, 4.0), (0.0, 0.0, 0.0), "Z", True, white, tr)
        b.add_camera((0, 0, 10), (0, 0, 0), 30.0)
        h = oracle.create_scene(b).intersect(o, d)[0]
        assert h["valid"] == 1
        if shape == "sphere":
            assert abs(h["t"] - 4.0) < 1e-4 and np.allclose(h["point"], [0, 0, 6], atol=1e-4) and np.allclose(h["normal"], [0, 0, 1], atol=1e-5)
        else:
            n = (rx @ ry) @ np.array([0.0, 0.0, 1.0])
            assert abs(h["t"] - 10.0) < 1e-4 and np.allclose(h["point"], [0, 0, 0], atol=1e-4)
            assert np.allclose(np.abs(np.dot(h["normal"], n)), 1.0, atol=1e-5) and np.dot(h["normal"], [0, 0, 1]) > 0   # two-sided: faces the ray


def test_mediums(pkg, oracle):
    """src/mediums: the Henyey-Greenstein phase function (pbrt's convention: both directions point away from the scattering point, so the
    mean of wi . wo is -g and g > 0 scatters forward) is a normalised pdf, its sampler follows it and returns unit vectors in the frame
    of wi (TangentFrame::from_normal(wi), hg.rs:82-91), g stored + 1 with the 0.001 nudge of hg.rs:69;
    the Rayleigh sampler's cosine solves the cubic of rayleigh.rs:70-75 and reports 3 (1 + cos^2) / 8; a free flight is
    -ln(1 - x) / sigma_s long and weighs exp(-sigma_t d) (hg.rs:96-115)."""
    import ctypes as C
    from util import fptr
    L = oracle.lib
    L.ptref_medium_sample_p.restype = None
    L.ptref_medium_sample_p.argtypes = [C.c_int, C.c_float, C.c_size_t] + [C.POINTER(C.c_float)] * 4
    L.ptref_medium_flight.restype = None
    L.ptref_medium_flight.argtypes = [C.c_float, C.c_float, C.c_size_t] + [C.POINTER(C.c_float)] * 3
    rng = np.random.default_rng(3)
    n = 1 << 16
    wi = rng.normal(size=(n, 3)).astype(np.float32); wi /= np.linalg.norm(wi, axis=1, keepdims=True).astype(np.float32)
    s2 = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    for g_stored in (1.0, 1.6, 0.3):
        wo = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32)
        L.ptref_medium_sample_p(0, g_stored, n, fptr(wi), fptr(s2), fptr(wo), fptr(pdf))
        g = np.float32(g_stored) + np.float32(0.001) - np.float32(1.0)
        cos = (wo * wi).sum(axis=1)
        assert np.allclose(np.linalg.norm(wo, axis=1), 1.0, atol=1e-5)
        assert abs(cos.mean() + g) < 0.01, (g_stored, cos.mean())                      # mean of wi . wo = -g
        denom = 1.0 + g * g + 2.0 * g * cos.astype(np.float64)
        # the pdf the sampler reports is phase_hg evaluated at the sampled cosine, and phase_hg integrates to 1 over the sphere
        assert np.allclose(pdf, (1.0 - g * g) / (denom * np.sqrt(denom) * 4.0 * np.pi), rtol=2e-4)
        mu = np.linspace(-1, 1, 20001)
        d = 1.0 + g * g + 2.0 * g * mu
        assert abs(np.trapezoid((1.0 - g * g) / (d * np.sqrt(d) * 4.0 * np.pi), mu) * 2.0 * np.pi - 1.0) < 1e-3
    wo = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32)
    L.ptref_medium_sample_p(1, 0.0, n, fptr(wi), fptr(s2), fptr(wo), fptr(pdf))
    cos = (wo * wi).sum(axis=1).astype(np.float64)
    assert np.allclose(np.linalg.norm(wo, axis=1), 1.0, atol=1e-4) and (np.abs(cos) <= 1.0 + 1e-5).all()
    assert np.allclose(pdf, 3.0 * (1.0 + cos * cos) / 8.0, rtol=1e-3, atol=1e-4)
    x = np.where(s2[:, 0] < 0.5, s2[:, 0] / 0.5, (s2[:, 0] - 0.5) / 0.5).astype(np.float64)   # Sample1D::choose(0.5)
    z = 2.0 * (2.0 * x - 1.0)
    assert np.allclose(cos ** 3 + 3.0 * cos, 2.0 * z, atol=2e-3)                      # cbrt(z + r) + cbrt(z - r) solves c^3 + 3 c = 2 z
    xs = rng.uniform(0, 0.999, 4096).astype(np.float32); dist = np.zeros_like(xs); w = np.zeros_like(xs)
    L.ptref_medium_flight(0.7, 0.2, xs.size, fptr(xs), fptr(dist), fptr(w))
    assert np.allclose(dist, -np.log((np.float32(1.0) - xs).astype(np.float64)) / np.float64(np.float32(0.7)), rtol=1e-5)   # (1 - x is taken in f32, hg.rs:98)
    assert np.allclose(w, np.exp(-np.float64(np.float32(0.2) + np.float32(0.7)) * dist.astype(np.float64)), rtol=1e-5)
