"""A THIRD reading of the integrator (round-3 verdict, "what's weak" 1): PathTracingIntegrator::color, random_walk and the light-sample loop restated once
more — scalar Python, written against src/integrator/pt.rs and utils.rs alone (line numbers below), structured as the reference is (a vertex list, then a second
pass over it), sharing no code with oracle/ptref.cpp or csrc/pt_stages.h.  It owns the ESTIMATOR: which vertices are pushed, the roulette, the throughput update,
the MIS weights of light and environment vertices, the light-sample formula with its extra light-side cosine, the balance heuristic, the shadow test that accepts any
light, the division by light_samples, the phases of ten samples.  Everything below the estimator comes from the oracle's PROBES, which have parity tests of their own:
World::hit (intersect), Material::{generate_and_evaluate, bsdf, emission}, the camera stage (camera_samples), the counter-based uniforms (ptref_draw4, laid out as
include/pt_numerics.h says) and the colour-matching fit (ptref_xyz_bar); the tangent frame is the one DESIGN.md section 2 lists (Duff et al.), AARect::sample / psa_pdf
(rect.rs:113-173) are restated here.  Decisions are taken in f32 on the probes' own numbers, so the paths are the oracle's; the film of a small Cornell render must
then agree with the oracle's to rounding.  CPU tier."""
import ctypes as C

import numpy as np

from util import numerics

F = np.float32
NORMAL_OFFSET = F(0.001)   # src/lib.rs:48
TAG_LIGHT = 1


def f3(x):
    return np.asarray(x, dtype=np.float32)


def dot(a, b):
    return F(F(F(a[0] * b[0]) + F(a[1] * b[1])) + F(a[2] * b[2]))


def norm(v):
    return np.sqrt(dot(v, v), dtype=np.float32)


def normalized(v):
    return (v / norm(v)).astype(np.float32)


class Frame:
    """TangentFrame::from_normal as DESIGN.md section 2 reads the math crate (Duff et al. 2017)."""
    def __init__(self, n):
        n = f3(n)
        sign = F(-1.0) if np.signbit(n[2]) else F(1.0)
        a = F(F(-1.0) / F(sign + n[2]))
        b = F(F(n[0] * n[1]) * a)
        self.t = f3([F(F(1.0) + F(F(F(sign * n[0]) * n[0]) * a)), F(sign * b), F(-sign * n[0])])
        self.b = f3([b, F(sign + F(F(n[1] * n[1]) * a)), F(-n[1])])
        self.n = n

    def to_local(self, v):
        return f3([dot(self.t, v), dot(self.b, v), dot(self.n, v)])

    def to_world(self, v):
        return (self.t * v[0] + self.b * v[1] + self.n * v[2]).astype(np.float32)


def signum(x):   # f32::signum: 1.0 for +0.0, -1.0 for -0.0
    return F(-1.0) if np.signbit(x) else F(1.0)


class Probes:
    def __init__(self, pkg, oracle, builder, rd):
        self.pkg, self.oracle, self.b, self.rd = pkg, oracle, builder, rd
        self.sc = oracle.create_scene(builder)

    def draw4(self, pixel, sample, dim):
        out = (C.c_float * 4)()
        self.oracle.lib.ptref_draw4(self.rd.seed, pixel, sample, dim, out)
        return f3(list(out))

    def hit(self, o, d):
        h = self.sc.intersect(f3(o)[None], f3(d)[None])[0]
        return None if h["valid"] == 0 else h

    def generate(self, material, lam, wi, s):
        f, wo, pdf = self.sc.bsdf_sample(material & 0xFFFF, f3([lam]), f3(wi)[None], f3(s)[None])
        return F(f[0]), f3(wo[0]), F(pdf[0])

    def bsdf(self, material, lam, wi, wo):
        f, pdf = self.sc.bsdf_eval(material & 0xFFFF, f3([lam]), f3(wi)[None], f3(wo)[None])
        return F(f[0]), F(pdf[0])

    def emission(self, material, lam, wi):
        return F(self.sc.emission(material & 0xFFFF, f3([lam]), f3(wi)[None])[0])

    def env_emission(self, lam):                                              # environment.rs:60-66: Constant { color, strength }
        e = self.b.environment
        assert e.kind == self.pkg.api.ENV_CONSTANT
        return F(F(self.sc.curve_eval(e.curve, f3([lam]))[0]) * F(e.strength))

    def sincos(self, x):                                                      # the routines of include/pt_numerics.h (the `math` crate's are not vendored)
        return F(numerics(self.oracle, 0, f3([x]))[0]), F(numerics(self.oracle, 1, f3([x]))[0])

    def xyz_bar(self, lam):
        out = (C.c_float * 3)()
        self.oracle.lib.ptref_xyz_bar(C.c_float(lam), out)
        return f3(list(out))


def tag(material):
    return (int(material) >> 16) & 3


class Vertex:
    def __init__(self, kind, local_wi, point, normal, material, instance, throughput, pdf_forward):
        self.kind, self.local_wi, self.point, self.normal = kind, local_wi, f3(point), f3(normal)
        self.material, self.instance, self.throughput, self.pdf_forward = material, instance, F(throughput), F(pdf_forward)


def random_walk(P, o, d, lam, pixel, sample, vertices):   # utils.rs:152-376 (TransportMode::Importance, ignore_backward)
    rd = P.rd
    beta = F(1.0)
    for bounce in range(rd.max_bounces if not rd.only_direct else 1):
        hit = P.hit(o, d)
        if hit is None:                                                       # :347-371: the environment vertex
            vertices.append(Vertex("env", f3([0, 0, 1]), f3(d) * F(1.0), d, 0, 0, beta, F(0.0)))
            break
        frame = Frame(hit["normal"])                                          # :175-176
        wi = normalized(frame.to_local(-f3(d)))
        kind = "light" if tag(hit["material"]) == TAG_LIGHT else "eye"        # :205-207
        r = P.draw4(pixel, sample, 32 + bounce * (1 + rd.light_samples))
        f, wo, pdf = P.generate(hit["material"], lam, wi, r[:2])              # :214-221
        cos_o = F(abs(wo[2]))
        if np.isnan(pdf):                                                     # :261-263: never pushed
            break
        rr = F(min(F(f / pdf), F(1.0))) if bounce >= rd.min_bounces else F(1.0)   # :266-276
        pdf_forward = F(pdf * F(rr / cos_o))                                  # :282
        vertices.append(Vertex(kind, wi, hit["point"], hit["normal"], int(hit["material"]), int(hit["instance"]), beta, pdf_forward))   # :299
        beta = F(beta * F(f / pdf_forward))                                   # :301
        if pdf_forward == 0:
            beta = F(0.0)
        if beta == 0:                                                         # :315
            break
        if r[2] > rr:                                                         # :319-322
            break
        o = (f3(hit["point"]) + f3(hit["normal"]) * F(NORMAL_OFFSET * signum(wo[2]))).astype(np.float32)   # :326-329
        d = normalized(frame.to_world(wo))


def rect_of(builder, instance):
    inst = builder.instances[instance]
    return f3(list(inst.size)), f3(list(inst.origin)), int(inst.axis), bool(inst.two_sided)


def rect_sample(rect, s, frm):    # rect.rs:113-155 (one-sided, normal = +Z: vec_shuffle is the identity)
    size, origin, axis, two_sided = rect
    assert axis == 2 and not two_sided
    point = (origin + f3([F(F(s[0] - F(0.5)) * size[0]), F(F(s[1] - F(0.5)) * size[1]), 0.0])).astype(np.float32)
    normal = f3([0, 0, 1])
    direction = (point - f3(frm)).astype(np.float32)
    cos_i = dot(normal, normalized(direction))
    area_pdf = F(F(1.0) / F(size[0] * size[1]))
    with np.errstate(divide="ignore", invalid="ignore"):
        pdf = F(F(area_pdf * dot(direction, direction)) / F(abs(cos_i)))      # PDF<Area>::convert_to_solid_angle
    if not np.isfinite(pdf):
        pdf = F(0.0)
    return normalized(direction), pdf


def rect_psa_pdf(rect, cos_o, cos_i, frm, to):   # rect.rs:156-173
    size = rect[0]
    direction = (f3(to) - f3(frm)).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        sa = F(F(F(F(1.0) / F(size[0] * size[1])) * dot(direction, direction)) / F(abs(cos_i)))
        return F(sa / F(abs(cos_o)))                                          # convert_to_projected_solid_angle


def power_heuristic(a, b):
    with np.errstate(divide="ignore", invalid="ignore"):
        return F(F(a * a) / F(F(a * a) + F(b * b)))


ENV_PDF = F(F(1.0) / F(4.0 * np.pi))      # environment.rs:124-126 (Constant: the uniform sphere, whatever uv)


def uv_to_direction(P, u, v):    # math::misc::uv_to_direction as DESIGN.md §10 reads it (z up)
    st, ct = P.sincos(F(F(u - F(0.5)) * F(2.0) * F(np.pi)))
    sp, cp = P.sincos(F(v * F(np.pi)))
    return f3([F(sp * ct), F(sp * st), cp])


def direct_illumination_from_world(P, lam, hit_point, hit_normal, frame, wi, material, throughput, s):   # pt.rs:224-331
    direction = uv_to_direction(P, s[0], s[1])                                # Constant: uv = the raw sample, environment.rs:301-305
    wo = frame.to_local(direction)
    if wo[2] <= 0:                                                            # :243-245
        return F(0.0)
    refl, scatter_pdf = P.bsdf(material, lam, wi, wo)
    o = (f3(hit_point) + f3(hit_normal) * F(NORMAL_OFFSET * signum(direction[2]))).astype(np.float32)   # :256 the WORLD z (kept quirk)
    if P.hit(o, direction) is not None:
        return F(0.0)
    weight = F(1.0) if P.rd.only_direct else F(ENV_PDF / F(ENV_PDF + scatter_pdf))
    return F(F(F(F(F(throughput * weight) * refl) * P.env_emission(lam)) * F(abs(wo[2]))) * F(F(1.0) / ENV_PDF))   # :316-323


def direct_illumination(P, lights, lam, hit_point, hit_normal, frame, wi, material, throughput, pixel, sample, bounce):   # pt.rs:333-393 + 146-218
    rd = P.rd
    total = F(0.0)
    env_p = F(P.b.env_sampling_probability) if lights else F(1.0)            # world/mod.rs:126-134
    if not lights and env_p == 0:
        return total
    for l in range(rd.light_samples):
        r = P.draw4(pixel, sample, 32 + bounce * (1 + rd.light_samples) + 1 + l)
        x = r[0]
        if x < env_p:                                                         # math::random::choose: the sample rescaled on either side
            total = F(total + direct_illumination_from_world(P, lam, hit_point, hit_normal, frame, wi, material, throughput, r[1:3]))
            continue
        x = F(F(x - env_p) / F(F(1.0) - env_p))
        n = len(lights)
        idx = int(min(max(F(F(n) * x), F(0.0)), F(n - 1)))                    # world/mod.rs:100-124
        rect = lights[idx][1]
        direction, light_pdf = rect_sample(rect, r[1:3], hit_point)           # pt.rs:147
        light_pdf = F(light_pdf * F(F(1.0) / F(n)))
        if light_pdf == 0:
            continue
        wo = frame.to_local(direction)
        refl, bounce_pdf = P.bsdf(material, lam, wi, wo)                      # :153-160
        weight = F(1.0) if rd.only_direct else F(light_pdf / F(light_pdf + bounce_pdf))   # power_heuristic_generic = balance, lib.rs:114-119
        o = (f3(hit_point) + f3(hit_normal) * F(NORMAL_OFFSET * signum(wo[2]))).astype(np.float32)
        sh = P.hit(o, direction)                                              # :171-176
        if sh is None or tag(sh["material"]) != TAG_LIGHT:                    # :177-178: ANY light
            continue
        lwi = Frame(sh["normal"]).to_local(-direction)
        le = P.emission(sh["material"], lam, lwi)
        cos_i, cos_o = F(abs(lwi[2])), F(abs(wo[2]))
        total = F(total + F(F(F(F(F(F(refl * throughput) * cos_i) * cos_o) * le) * weight) / light_pdf))   # :196-202
    return total


def color(P, lights, pixel, sample):   # pt.rs:397-615
    rd = P.rd
    o, d, lam = P.sc.camera_samples(rd, [pixel], [sample])
    o, d, lam = f3(o[0]), f3(d[0]), F(lam[0])
    path = [Vertex("camera", f3([0, 0, 0]), o, d, 0, 0, F(1.0), F(100.0))]
    random_walk(P, o, d, lam, pixel, sample, path)
    energy = F(0.0)
    for index in range(1, len(path)):
        prev, v = path[index - 1], path[index]
        if v.kind == "env":                                                   # :487-511
            wo = v.normal
            cos_i = F(abs(dot(prev.normal, wo)))
            with np.errstate(divide="ignore", invalid="ignore"):
                w = power_heuristic(F(prev.pdf_forward / cos_i), F(ENV_PDF / cos_i))
            energy = F(energy + F(F(w * v.throughput) * P.env_emission(lam)))
        elif v.kind == "light":                                               # :512-561
            em = P.emission(v.material, lam, v.local_wi)
            if em > 0:
                if rd.light_samples == 0 or prev.kind == "camera":
                    energy = F(energy + F(v.throughput * em))
                elif not rd.only_direct:
                    nd = normalized((v.point - prev.point).astype(np.float32))
                    hyp = rect_psa_pdf(dict(lights)[v.instance], dot(prev.normal, nd), dot(v.normal, nd), prev.point, v.point)
                    w = power_heuristic(prev.pdf_forward, hyp)
                    energy = F(energy + F(F(w * v.throughput) * em))
        else:                                                                 # :562-604
            n = normalized(v.normal)                                          # HitRecord::from(vertex) normalises again
            frame = Frame(n)
            wi = frame.to_local(normalized((prev.point - v.point).astype(np.float32)))
            if rd.light_samples > 0:
                lc = direct_illumination(P, lights, lam, v.point, n, frame, wi, v.material, v.throughput, pixel, sample, index - 1)
                energy = F(energy + F(lc / F(rd.light_samples)))
    return (P.xyz_bar(lam) * energy).astype(np.float32)                       # XYZColor::from(SingleWavelength)


def render(P, lights):
    rd = P.rd
    film = np.zeros((rd.height, rd.width, 4), np.float32)
    for y in range(rd.height):
        for x in range(rd.width):
            px, temp = np.zeros(3, np.float32), np.zeros(3, np.float32)
            for s in range(rd.spp):                                           # tiled.rs:347-398: phases of ten samples, then / spp
                temp = (temp + color(P, lights, y * rd.width + x, s)).astype(np.float32)
                if (s + 1) % 10 == 0 or s + 1 == rd.spp:
                    px = (px + temp).astype(np.float32)
                    temp = np.zeros(3, np.float32)
            film[y, x, :3] = px / F(rd.spp)
    return film


def agree(P, lights, lit):
    mine = render(P, lights)
    ref, _ = P.sc.render(P.rd)
    assert (np.isfinite(ref) == np.isfinite(mine)).all()
    ok = np.isfinite(ref) & np.isfinite(mine)
    d = np.abs(np.where(ok, mine - ref, 0.0))[..., :3]
    assert (ref[..., :3] > 0).mean() > lit, float((ref[..., :3] > 0).mean())
    assert (d / np.maximum(np.abs(ref[..., :3]), 1e-3)).max() < 2e-5, float((d / np.maximum(np.abs(ref[..., :3]), 1e-3)).max())
    return ref


def sky_and_lamp(pkg):
    """A rect lamp on the ground, a white and a rough-glass sphere over a floor, under a constant sky sampled half of the time (env_sampling_probability 0.5)."""
    b = pkg.scene.SceneBuilder()
    pkg.scene.add_library_curves(b, ["simple_sky_blue"])
    b.set_environment_constant(b.curve("simple_sky_blue"), 0.5)
    b.env_sampling_probability = 0.5
    lamp = pkg.scene.add_library_material(b, "diffuse_light_cornell")
    white = pkg.scene.add_library_material(b, "lambertian_white")
    glass = pkg.scene.add_library_material(b, "ggx_glass_rough")
    b.add_rect((0.5, 0.5), (0.0, 0.0, 0.0), "Z", False, lamp)
    b.add_rect((6.0, 6.0), (0.0, 0.0, -0.01), "Z", True, white)
    b.add_sphere(0.35, (0.1, 0.5, 0.6), white)
    b.add_sphere(0.3, (0.0, -0.45, 0.5), glass)
    b.add_camera((-3.0, 0.0, 1.2), (0.0, 0.0, 0.4), 35.0, focal_distance=3.0, aperture_diameter=0.01)
    return b


def test_environment_light_samples_agree_with_a_third_reading(pkg, oracle):
    """estimate_direct_illumination_from_world (pt.rs:224-331), the choice between the sky and the lights (pt.rs:346-358, world/mod.rs:126-134) and the MIS weight of
    an environment vertex (pt.rs:487-511), for a Constant environment: the white furnace (no light: every sample to the sky) and a scene with both."""
    b = pkg.scene.white_furnace()
    for kw in ({}, {"only_direct": True}, {"light_samples": 2, "seed": 4}):
        agree(Probes(pkg, oracle, b, pkg.api.render_desc(10, 10, 3, 8, **kw)), [], 0.6)
    b = sky_and_lamp(pkg)
    lights = [(i, rect_of(b, i)) for i, inst in enumerate(b.instances) if inst.kind == pkg.api.SHAPE_RECT and tag(inst.material) == TAG_LIGHT]
    assert len(lights) == 1
    for kw in ({}, {"light_samples": 3, "seed": 11}):
        agree(Probes(pkg, oracle, b, pkg.api.render_desc(16, 12, 3, 6, **kw)), lights, 0.9)


def test_gem_film_agrees_with_a_third_reading(pkg, oracle):
    """The same estimator over C3's scene: rough-free moissanite (GGX dielectric with dispersion) on a transformed mesh, a SharpLight rect, depth 12."""
    b = pkg.scene.cornell_gem()
    lights = [(i, rect_of(b, i)) for i, inst in enumerate(b.instances) if inst.kind == pkg.api.SHAPE_RECT and tag(inst.material) == TAG_LIGHT]
    assert len(lights) == 1
    rd = pkg.api.render_desc(14, 14, 4, 12, seed=2)
    P = Probes(pkg, oracle, b, rd)
    mine = render(P, lights)
    ref, _ = P.sc.render(rd)
    assert (np.isfinite(ref) == np.isfinite(mine)).all()
    ok = np.isfinite(ref) & np.isfinite(mine)
    d = np.abs(np.where(ok, mine - ref, 0.0))[..., :3]
    assert (ref[..., :3] > 0).mean() > 0.3
    assert (d / np.maximum(np.abs(ref[..., :3]), 1e-3)).max() < 2e-5


def test_cornell_film_agrees_with_a_third_reading(pkg, oracle):
    b = pkg.scene.cornell_box()
    lights = [(i, rect_of(b, i)) for i, inst in enumerate(b.instances) if inst.kind == pkg.api.SHAPE_RECT and tag(inst.material) == TAG_LIGHT]
    assert len(lights) == 1
    for kw, lit in (({}, 0.5), ({"light_samples": 0}, 0.02), ({"only_direct": True}, 0.3), ({"min_bounces": 3, "seed": 5}, 0.5), ({"light_samples": 3, "seed": 9}, 0.5)):
        rd = pkg.api.render_desc(12, 10, 3, 6, **kw)
        P = Probes(pkg, oracle, b, rd)
        mine = render(P, lights)
        ref, _ = P.sc.render(rd)
        ok = np.isfinite(ref) & np.isfinite(mine)
        assert (np.isfinite(ref) == np.isfinite(mine)).all()
        d = np.abs(np.where(ok, mine - ref, 0.0))[..., :3]
        scale = np.maximum(np.abs(ref[..., :3]), 1e-3)
        assert (ref[..., :3] > 0).mean() > lit, (kw, float((ref[..., :3] > 0).mean()))   # a lit image
        assert (d / scale).max() < 2e-5, (kw, float((d / scale).max()))      # same paths, the same sums: rounding only
