"""A THIRD reading of the GGX material (round-3 verdict, "what's weak" 1: the oracle's and the engine's material text share names and factoring, so a shared
misreading would pass every engine-vs-oracle test).  This file restates src/materials/ggx.rs once more — vectorised numpy in f64, written against the Rust
source alone (line numbers below), sharing no code, header or structure with oracle/ptref.cpp or csrc/pt_device.h — and the oracle's probes must agree with it:

  * Material::bsdf (ggx.rs:256-400): reflection F D G / (4 |cos_i cos_o|), transmission with the eta-relative half vector, the Jacobian, eta^2 in Importance
    mode, the Fresnel-weighted mixture pdf;
  * the sampling half of generate_and_evaluate (ggx.rs:401-447): sample_vndf / sample_wh (:129-180), lobe choice on the un-rescaled sample.x, reflect / refract /
    total internal reflection.

f64 against the reference's f32: agreement within 2e-3 relative on the values and 2e-3 on the sampled directions, on inputs kept away from the grazing angles where f32
cancellation dominates.  Spectral inputs (eta, eta_o, kappa at lambda) come from the oracle's curve probe: curves have their own tests (test_curves).  CPU tier."""
import numpy as np
import pytest

PI = np.pi


def dot(a, b):
    return (a * b).sum(-1)


def normalized(v):
    return v / np.sqrt(dot(v, v))[..., None]


def fresnel_dielectric(eta_i, eta_t, cos_i):   # ggx.rs:19-48
    cos_i = np.clip(cos_i, -1.0, 1.0)
    swap = cos_i < 0.0
    cos_i = np.where(swap, -cos_i, cos_i)
    eta_i, eta_t = np.where(swap, eta_t, eta_i), np.where(swap, eta_i, eta_t)
    sin_t = eta_i / eta_t * np.sqrt(np.maximum(0.0, 1.0 - cos_i * cos_i))
    cos_t = np.sqrt(np.maximum(0.0, 1.0 - sin_t * sin_t))
    r_par = (eta_t * cos_i - eta_i * cos_t) / (eta_t * cos_i + eta_i * cos_t)
    r_perp = (eta_i * cos_i - eta_t * cos_t) / (eta_i * cos_i + eta_t * cos_t)
    return (r_par * r_par + r_perp * r_perp) / 2.0


def fresnel_conductor(eta_i, eta_t, k_t, cos_i):   # ggx.rs:50-85
    cos_i = np.clip(cos_i, -1.0, 1.0)
    swap = cos_i < 0.0
    cos_i = np.where(swap, -cos_i, cos_i)
    eta_i, eta_t = np.where(swap, eta_t, eta_i), np.where(swap, eta_i, eta_t)
    eta, etak = eta_t / eta_i, k_t / eta_i
    c2 = cos_i * cos_i
    s2 = 1.0 - c2
    eta2, etak2 = eta * eta, etak * etak
    t0 = eta2 - etak2 - s2
    a2plusb2 = np.sqrt(t0 * t0 + eta2 * etak2 * 4.0)
    t1 = a2plusb2 + c2
    a = np.sqrt((a2plusb2 + t0) * 0.5)
    t2 = a * cos_i * 2.0
    rs = (t1 - t2) / (t1 + t2)
    t3 = a2plusb2 * c2 + s2 * s2
    t4 = t2 * s2
    rp = rs * (t3 - t4) / (t3 + t4)
    return (rs + rp) / 2.0


def ggx_d(alpha, wm):     # ggx.rs:87-97
    t = wm[..., 2] ** 2 + (wm[..., 0] / alpha) ** 2 + (wm[..., 1] / alpha) ** 2
    return 1.0 / (PI * alpha * alpha * t * t)


def ggx_lambda(alpha, w):  # ggx.rs:99-107
    a2 = alpha * alpha
    with np.errstate(divide="ignore", invalid="ignore"):
        c = 1.0 + (a2 * w[..., 0] ** 2 + a2 * w[..., 1] ** 2) / w[..., 2] ** 2
    return np.where(w[..., 2] == 0.0, 0.0, np.sqrt(c) * 0.5 - 0.5)


def ggx_g(alpha, wi, wo):  # ggx.rs:109-113
    return 1.0 / (1.0 + ggx_lambda(alpha, wi) + ggx_lambda(alpha, wo))


def ggx_vnpdf(alpha, wi, wh):      # ggx.rs:115-119
    return ggx_d(alpha, wh) * np.abs(dot(wi, wh)) / ((1.0 + ggx_lambda(alpha, wi)) * np.abs(wi[..., 2]))


def ggx_vnpdf_no_d(alpha, wi, wh):  # ggx.rs:121-123
    return np.abs(dot(wi, wh) / ((1.0 + ggx_lambda(alpha, wi)) * wi[..., 2]))


def reflectance(metallic, eo, ei, kappa, cos):         # ggx.rs:222-228
    return fresnel_conductor(eo, ei, kappa, cos) if metallic else fresnel_dielectric(eo, ei, cos)


def reflectance_probability(metallic, eo, ei, kappa, cos):   # ggx.rs:230-243
    return np.ones_like(cos) if metallic else np.clip(fresnel_dielectric(eo, ei, cos), 0.0, 1.0)


def eta_rel(eo, ei, wi):   # ggx.rs:244-253
    return np.where(wi[..., 2] < 0.0, eo / ei, ei / eo)


def bsdf(alpha, metallic, ei, eo, kappa, wi, wo, importance=True):   # ggx.rs:256-400
    wi = normalized(wi)
    same = wi[..., 2] * wo[..., 2] > 0.0
    g = np.abs(wi[..., 2] * wo[..., 2])
    # reflection
    wh = normalized(wo + wi)
    wh = np.where((wh[..., 2] < 0.0)[..., None], -wh, wh)
    ndotv = dot(wi, wh)
    glossy = reflectance(metallic, eo, ei, kappa, ndotv) * (0.25 / g) * ggx_d(alpha, wh) * ggx_g(alpha, wi, wo)
    glossy_pdf = np.where(np.abs(ndotv) == 0.0, 0.0, ggx_vnpdf(alpha, wi, wh) * 0.25 / np.abs(ndotv))
    # transmission
    er = eta_rel(eo, ei, wi)
    wht = normalized(wi + er[..., None] * wo)
    wht = np.where((wht[..., 2] < 0.0)[..., None], -wht, wht)
    nv, nl = dot(wi, wht), dot(wo, wht)
    sq = nv + er * nl
    d1 = nl / (sq * sq)
    d2 = er * er * d1
    d = ggx_d(alpha, wht)
    weight = d * ggx_g(alpha, wi, wo) * nv * (d2 if importance else d1) / g
    trans_pdf = np.abs(d * ggx_vnpdf_no_d(alpha, wi, wht) * d2)
    trans = (1.0 - reflectance(metallic, eo, ei, kappa, nv)) * np.abs(weight)
    if metallic:
        trans, trans_pdf = np.zeros_like(trans), np.zeros_like(trans_pdf)
    glossy, glossy_pdf = np.where(same, glossy, 0.0), np.where(same, glossy_pdf, 0.0)
    trans, trans_pdf = np.where(same, 0.0, trans), np.where(same, 0.0, trans_pdf)
    rp = reflectance_probability(metallic, eo, ei, kappa, wi[..., 2])
    f, pdf = glossy + trans, rp * glossy_pdf + (1.0 - rp) * trans_pdf
    return np.where(g == 0.0, 0.0, f), np.where(g == 0.0, 0.0, pdf)


def sample_wh(alpha, wi, x, y):   # ggx.rs:129-180
    flip = wi[..., 2] < 0.0
    w = np.where(flip[..., None], -wi, wi)
    v = normalized(np.stack([alpha * w[..., 0], alpha * w[..., 1], w[..., 2]], -1))
    z = np.array([0.0, 0.0, 1.0])
    t1 = np.where((v[..., 2] < 0.9999)[..., None], normalized(np.cross(v, z) + 1e-300), np.array([1.0, 0.0, 0.0]))
    t2 = np.cross(t1, v)
    a = 1.0 / (1.0 + v[..., 2])
    r = np.sqrt(x)
    phi = np.where(y < a, y / a * PI, PI + (y - a) / (1.0 - a) * PI)
    p1 = r * np.cos(phi)
    p2 = r * np.sin(phi) * np.where(y < a, 1.0, v[..., 2])
    n = p1[..., None] * t1 + p2[..., None] * t2 + np.sqrt(np.maximum(0.0, 1.0 - p1 * p1 - p2 * p2))[..., None] * v
    wh = normalized(np.stack([alpha * n[..., 0], alpha * n[..., 1], np.maximum(n[..., 2], 0.0)], -1))
    return np.where(flip[..., None], -wh, wh)


def generate(alpha, metallic, ei, eo, kappa, wi, x, y):   # ggx.rs:401-447: direction only
    wh = normalized(sample_wh(alpha, wi, x, y))
    rp = reflectance_probability(metallic, eo, ei, kappa, dot(wh, wi))
    refl = normalized(-wi - 2.0 * dot(-wi, wh)[..., None] * wh)                       # reflect, ggx.rs:3-6
    er = 1.0 / eta_rel(eo, ei, wi)
    cos_i = dot(wi, wh)                                                                # refract, ggx.rs:8-17
    s2t = er * er * np.maximum(0.0, 1.0 - cos_i * cos_i)
    tir = s2t >= 1.0
    refr = normalized(-wi * er[..., None] + wh * (er * cos_i - np.sqrt(np.maximum(0.0, 1.0 - s2t)))[..., None])
    reflect_chosen = (x <= rp) | tir
    return np.where(reflect_chosen[..., None], refl, refr), reflect_chosen, rp


def scene_and_materials(pkg):
    b = pkg.scene.SceneBuilder()
    pkg.scene.add_library_curves(b, ["flat_zero"])
    b.set_environment_constant(b.curve("flat_zero"), 0.0)
    ids = {name: pkg.scene.add_library_material(b, name) & 0xFFFF for name in ("ggx_glass_rough", "ggx_gold", "ggx_moissanite")}
    rough = {"ggx_glass_rough": 0.2, "ggx_gold": 0.004, "ggx_moissanite": 0.0004}
    b.add_sphere(1.0, (0.0, 0.0, 0.0), b.material_ids["ggx_glass_rough"])
    b.add_camera((-5.0, 0.0, 0.0), (0.0, 0.0, 0.0), 30.0)
    return b, ids, rough


def spectral(pkg, b, sc, name, lam):
    """eta, eta_o, kappa of a library GGX material at lambda, through the oracle's curve probe (the curve names of scene.add_library_material)."""
    if name == "ggx_gold":
        return sc.curve_eval(b.curve("gold_n"), lam), sc.curve_eval(b.curve("air_ior"), lam), sc.curve_eval(b.curve("gold_k"), lam), True
    return sc.curve_eval(b.curve(name + ".eta"), lam), sc.curve_eval(b.curve("air_ior"), lam), np.zeros_like(lam), False


def unit(rng, n, zmin):
    """directions with |z| >= zmin (away from grazing, where the f32 reference cancels)"""
    z = rng.uniform(zmin, 1.0, n) * rng.choice([-1.0, 1.0], n)
    phi = rng.uniform(0, 2 * PI, n)
    r = np.sqrt(1 - z * z)
    return np.stack([r * np.cos(phi), r * np.sin(phi), z], -1)


@pytest.mark.parametrize("name", ["ggx_glass_rough", "ggx_gold"])
def test_bsdf_agrees_with_a_third_reading(pkg, oracle, name):
    b, ids, rough = scene_and_materials(pkg)
    sc = oracle.create_scene(b)
    rng = np.random.default_rng(17)
    n = 20000
    lam = rng.uniform(400, 700, n).astype(np.float32)
    wi, wo = unit(rng, n, 0.15).astype(np.float32), unit(rng, n, 0.15).astype(np.float32)
    ei, eo, kappa, metallic = spectral(pkg, b, sc, name, lam)
    alpha = rough[name] if name != "ggx_gold" else 0.004
    f, pdf = sc.bsdf_eval(ids[name], lam, wi, wo)
    f3, pdf3 = bsdf(np.float64(alpha), metallic, ei.astype(np.float64), eo.astype(np.float64), kappa.astype(np.float64), wi.astype(np.float64), wo.astype(np.float64))
    # (a mirror-like conductor: the lobe is a few milliradians wide, most random pairs sit in its far tail where D underflows towards 0 in both readings — compared
    # where the value is representable; the rough glass everywhere)
    ok = np.isfinite(f3) & np.isfinite(pdf3)
    big = ok & ((np.abs(f3) > 1e-12) | (np.abs(pdf3) > 1e-12))
    assert big.sum() > n // 4
    rel_f = np.abs(f[big] - f3[big]) / np.maximum(np.abs(f3[big]), 1e-12)
    rel_p = np.abs(pdf[big] - pdf3[big]) / np.maximum(np.abs(pdf3[big]), 1e-12)
    assert np.percentile(rel_f, 99.5) < 2e-3 and np.percentile(rel_p, 99.5) < 2e-3, (name, float(rel_f.max()), float(rel_p.max()))
    assert np.median(rel_f) < 2e-5 and np.median(rel_p) < 2e-5
    # what the reference asserts (ggx.rs:637-756): non-negative everywhere
    assert (f >= 0).all() and (pdf >= 0).all()
    if metallic:
        assert (f[wi[:, 2] * wo[:, 2] < 0] == 0).all()      # a conductor transmits nothing


@pytest.mark.parametrize("name", ["ggx_glass_rough", "ggx_moissanite", "ggx_gold"])
def test_sampling_agrees_with_a_third_reading(pkg, oracle, name):
    b, ids, rough = scene_and_materials(pkg)
    sc = oracle.create_scene(b)
    rng = np.random.default_rng(23)
    n = 20000
    lam = rng.uniform(400, 700, n).astype(np.float32)
    wi = unit(rng, n, 0.1).astype(np.float32)
    s2 = rng.random((n, 2)).astype(np.float32)
    ei, eo, kappa, metallic = spectral(pkg, b, sc, name, lam)
    f, wo, pdf = sc.bsdf_sample(ids[name], lam, wi, s2)
    wo3, refl3, rp = generate(np.float64(rough[name]), metallic, ei.astype(np.float64), eo.astype(np.float64), kappa.astype(np.float64), wi.astype(np.float64),
                              s2[:, 0].astype(np.float64), s2[:, 1].astype(np.float64))
    # the lobe decision is a comparison of sample.x with a Fresnel value: the two readings may differ where the two are within rounding of each other
    decided = np.abs(s2[:, 0] - rp) > 1e-4
    same_side = (wo[:, 2] * wi[:, 2] > 0) == (wo3[:, 2] * wi[:, 2] > 0)
    assert same_side[decided].mean() > 0.9995, (name, float(same_side[decided].mean()))
    agree = decided & same_side
    err = np.linalg.norm(wo[agree].astype(np.float64) - wo3[agree], axis=1)
    assert np.percentile(err, 99.5) < 2e-3 and np.median(err) < 1e-5, (name, float(err.max()), float(np.median(err)))
    # and the value returned with a reflected sample is the bsdf's at that pair — from OUTSIDE the surface.  (From inside, the reference's two functions disagree by
    # construction: generate_and_evaluate takes the Fresnel term at wi . wh with the sampled wh on wi's side — a positive cosine, "entering" — ggx.rs:422,459-461, while bsdf
    # flips wh to +z and gets a negative one, "leaving", with total internal reflection, ggx.rs:290-298.  The oracle restates both as they are; this test only notes it.)
    if rough[name] >= 0.1:
        f3, pdf3 = bsdf(np.float64(rough[name]), metallic, ei.astype(np.float64), eo.astype(np.float64), kappa.astype(np.float64), wi.astype(np.float64), wo.astype(np.float64))
        refl = agree & (wo[:, 2] * wi[:, 2] > 0) & (np.abs(wo[:, 2]) > 0.05)
        outside = refl & (wi[:, 2] > 0)
        rel = np.abs(f[outside] - f3[outside]) / np.maximum(np.abs(f3[outside]), 1e-9)
        assert outside.sum() > 300 and np.percentile(rel, 99) < 5e-3, float(np.percentile(rel, 99))
        inside = refl & (wi[:, 2] < 0)
        if not metallic:
            assert (np.abs(f[inside] - f3[inside]) / np.maximum(np.abs(f3[inside]), 1e-9) > 0.5).mean() > 0.5    # (the quirk is there, in the reference's own terms)
