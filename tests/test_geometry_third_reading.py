"""A THIRD reading of the layers under the integrator (round-3 verdict, "what's weak" 1): the camera and the primitives restated once more in numpy f64 from the
Rust source alone — projective_camera.rs, sphere.rs, rect.rs, disk.rs, instance.rs, aabb.rs, mesh.rs — sharing no code with oracle/ptref.cpp or csrc/pt_device.h, and
compared with the oracle's probes (camera_samples, intersect) on scenes of one primitive.  f64 against the oracle's f32: distances, points and normals agree to
rounding; the DECISIONS (hit or not, which root, which side) are compared where f64 decides them with a margin.  CPU tier."""
import ctypes as C

import numpy as np
import pytest


def unit(v):
    v = np.asarray(v, np.float64)
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def draws(oracle, seed, pixel, sample, dim):
    out = (C.c_float * 4)()
    oracle.lib.ptref_draw4(seed, pixel, sample, dim, out)
    return np.array(list(out), np.float64)


# ------------------------------------------------------------------------------------------------ camera
def camera_rays(oracle, cam, rd, pixels, samples):
    """ProjectiveCamera::new (projective_camera.rs:27-95), with_aspect_ratio (:121-133), get_ray (:101-120); the film point and the wavelength as
    tiled.rs:369-375 and pt.rs:406-417 form them.  The uniforms and the circular aperture's rejection are this repo's (include/pt_numerics.h)."""
    look_from, look_at, v_up = (np.array(list(x), np.float64) for x in (cam.look_from, cam.look_at, cam.v_up))
    direction = unit(look_at - look_from)
    half_height = np.tan(np.radians(np.float64(cam.vfov)) / 2.0)
    half_width = (rd.width / rd.height) * half_height
    w = -direction
    u = -unit(np.cross(unit(v_up), w))
    v = unit(np.cross(w, u))
    fd = np.float64(cam.focal_distance)
    llc = look_from - u * half_width * fd - v * half_height * fd - w * fd
    horizontal, vertical = u * 2.0 * half_width * fd, v * 2.0 * half_height * fd
    O, D, L = [], [], []
    for p, s in zip(pixels, samples):
        f = draws(oracle, rd.seed, p, s, 0)
        x, y = p % rd.width, p // rd.width
        cu, cv = (x + f[0]) / rd.width, (y + f[1]) / rd.height
        L.append(rd.wavelength_lo + f[2] * (rd.wavelength_hi - rd.wavelength_lo))
        eps = np.float64(np.finfo(np.float32).eps)
        cu, cv = min(max(cu, 0.0), 1.0 - eps), min(max(cv, 0.0), 1.0 - eps)
        a = None
        for blk in range(16):
            r = draws(oracle, rd.seed, p, s, 1 + blk) * 2.0 - 1.0
            for q in (r[:2], r[2:]):
                if a is None and q @ q <= 1.0:
                    a = q
        rdv = np.float64(cam.aperture_diameter) * a
        origin = look_from + u * rdv[0] + v * rdv[1]
        O.append(origin)
        D.append(unit(llc + cu * horizontal + cv * vertical - origin))
    return np.array(O), np.array(D), np.array(L)


def test_camera_agrees_with_a_third_reading(pkg, oracle):
    for b, (W, H) in ((pkg.scene.cornell_box(), (64, 48)), (pkg.scene.cornell_gem(), (96, 54)), (pkg.scene.white_furnace(), (20, 20))):
        rd = pkg.api.render_desc(W, H, 4, 4, seed=3)
        sc = oracle.create_scene(b)
        rng = np.random.default_rng(1)
        pixels = rng.integers(0, W * H, 300).astype(np.uint32)
        samples = rng.integers(0, 4, 300).astype(np.uint32)
        o, d, lam = sc.camera_samples(rd, pixels, samples)
        O, D, L = camera_rays(oracle, b.cameras[rd.camera_index], rd, pixels.tolist(), samples.tolist())
        assert np.abs(o - O).max() < 2e-6
        assert np.abs(d - D).max() < 2e-6        # a direction on the unit sphere: a few ulp of 1
        assert np.abs(lam - L).max() < 1e-4       # nanometres
        assert (np.abs(np.linalg.norm(d, axis=1) - 1.0) < 1e-6).all()


# ------------------------------------------------------------------------------------------------ primitives
def sphere_hit(o, d, centre, radius):    # sphere.rs:34-87 (t0 = 0, t1 = inf)
    oc = o - centre
    a, b, c = (d * d).sum(-1), (oc * d).sum(-1), (oc * oc).sum(-1) - radius * radius
    disc = b * b - a * c
    with np.errstate(invalid="ignore"):
        root = np.sqrt(disc)
        t_near, t_far = (-b - root) / a, (-b + root) / a
    t = np.where(disc > 0, np.where(t_near > 0, t_near, np.where(t_far > 0, t_far, np.inf)), np.inf)
    p = o + t[:, None] * d
    return t, p, (p - centre) / radius, disc


def plane_hit(o, d, origin, axis):       # rect.rs:69-112 / disk.rs:31-62: the shuffled ray meets z = 0
    to, td = o - origin, d
    order = {0: [2, 1, 0], 1: [0, 2, 1], 2: [0, 1, 2]}[axis]   # vec_shuffle: X <-> Z, Y <-> Z, identity
    to, td = to[:, order], td[:, order]
    with np.errstate(divide="ignore", invalid="ignore"):
        t = -to[:, 2] / td[:, 2]
    ok = (td[:, 2] != 0) & (t > 0) & np.isfinite(t)
    return np.where(ok, t, np.inf), to[:, 0] + t * td[:, 0], to[:, 1] + t * td[:, 1]


def aabb_hit(o, d, lo, hi):                 # aabb.rs:37-65 with t0 = 0, t1 = inf: the decision and how far it is from the other one
    with np.errstate(divide="ignore", invalid="ignore"):
        a = np.where(d == 0, 0.0, (lo - o) / d)
        b = np.where(d == 0, np.inf, (hi - o) / d)
    tmin, tmax = np.minimum(a, b), np.maximum(a, b)
    enter, leave = np.maximum(tmin.max(-1), 0.0), tmax.min(-1)          # the w lane: min 0, max inf; tmax < scaled_t0 = 0 on any axis rejects
    return ~(enter > leave), np.abs(enter - leave)


def rays_at(rng, n, centre, spread):
    o = np.array([-3.0, 0.4, 1.1]) + rng.normal(0, 0.2, (n, 3))
    d = unit(np.asarray(centre) + rng.normal(0, spread, (n, 3)) - o)
    return o.astype(np.float32), d.astype(np.float32)


def one_primitive(pkg, add):
    b = pkg.scene.SceneBuilder()
    pkg.scene.add_library_curves(b, ["flat_zero"])
    b.set_environment_constant(b.curve("flat_zero"), 0.0)
    m = pkg.scene.add_library_material(b, "lambertian_white")
    add(b, m)
    b.add_camera((-3.0, 0.0, 1.0), (0.0, 0.0, 0.0), 35.0)
    return b


def test_sphere_agrees_with_a_third_reading(pkg, oracle):
    centre, radius = np.array([0.2, -0.1, 0.3]), 0.7
    sc = oracle.create_scene(one_primitive(pkg, lambda b, m: b.add_sphere(radius, tuple(centre), m)))
    rng = np.random.default_rng(2)
    o, d = rays_at(rng, 6000, centre, 0.6)
    inside = (centre + rng.normal(0, 0.2, (500, 3))).astype(np.float32)          # rays from inside: the far root
    o, d = np.vstack([o, inside]), np.vstack([d, unit(rng.normal(0, 1, (500, 3))).astype(np.float32)])
    h = sc.intersect(o, d)
    t, p, n, disc = sphere_hit(o.astype(np.float64), d.astype(np.float64), centre.astype(np.float32).astype(np.float64), np.float64(np.float32(radius)))
    sure = np.abs(disc) > 1e-4                                                     # (a grazing ray: f32 and f64 may differ on disc > 0)
    assert ((h["valid"] == 1) == np.isfinite(t))[sure].all()
    both = (h["valid"] == 1) & np.isfinite(t) & sure
    assert both.sum() > 3000 and (both[-500:]).sum() > 400
    cond = 1.0 + np.abs((o.astype(np.float64) - centre) * d.astype(np.float64)).sum(-1) / np.sqrt(np.maximum(disc, 1e-12))   # -b - sqrt(disc): the root's cancellation
    assert (np.abs(h["t"][both] - t[both]) / cond[both]).max() < 2e-6
    assert (np.abs(h["point"][both] - p[both]).max(-1) / cond[both]).max() < 2e-6
    assert (np.abs(h["normal"][both] - n[both]).max(-1) / cond[both]).max() < 3e-6


@pytest.mark.parametrize("axis", ["X", "Y", "Z"])
def test_rect_agrees_with_a_third_reading(pkg, oracle, axis):
    origin, size = np.array([0.1, 0.2, 0.3]), (0.9, 0.5)
    k = "XYZ".index(axis)
    for two_sided in (False, True):
        sc = oracle.create_scene(one_primitive(pkg, lambda b, m: b.add_rect(size, tuple(origin), axis, two_sided, m)))
        rng = np.random.default_rng(3 + k)
        o = (origin + rng.normal(0, 1.5, (6000, 3))).astype(np.float32)
        d = unit(origin + rng.normal(0, 0.5, (6000, 3)) - o).astype(np.float32)
        h = sc.intersect(o, d)
        o64, d64 = o.astype(np.float64), d.astype(np.float64)
        t, xh, yh = plane_hit(o64, d64, origin.astype(np.float32).astype(np.float64), k)
        hx, hy = np.float64(np.float32(size[0])) / 2, np.float64(np.float32(size[1])) / 2
        margin = np.minimum(np.abs(np.abs(xh) - hx), np.abs(np.abs(yh) - hy))
        hit = np.isfinite(t) & (np.abs(xh) <= hx) & (np.abs(yh) <= hy)
        sure = ~np.isfinite(t) | (margin > 1e-5)
        assert ((h["valid"] == 1) == hit)[sure].all()
        both = hit & (h["valid"] == 1) & sure
        assert both.sum() > 500
        assert np.abs(h["t"][both] - t[both]).max() < 1e-5
        assert np.abs(h["point"][both] - (o64 + t[:, None] * d64)[both]).max() < 1e-5
        normal = np.zeros((len(t), 3)); normal[:, k] = 1.0                          # Vec3::from_axis; flipped when two-sided and seen from behind (rect.rs:91-96)
        if two_sided:
            normal[d64[:, k] > 0] *= -1.0
        assert np.array_equal(h["normal"][both], normal[both].astype(np.float32))
        uv = np.stack([(xh + hx) / (2 * hx), (yh + hy) / (2 * hy)], 1)
        assert np.abs(h["uv"][both] - uv[both]).max() < 1e-5


def test_disk_agrees_with_a_third_reading(pkg, oracle):
    origin, radius = np.array([0.0, 0.3, 0.2]), 0.6
    for two_sided in (False, True):
        sc = oracle.create_scene(one_primitive(pkg, lambda b, m: b.add_disk(radius, tuple(origin), two_sided, m)))
        rng = np.random.default_rng(7)
        o = (origin + rng.normal(0, 1.5, (6000, 3))).astype(np.float32)
        d = unit(origin + rng.normal(0, 0.5, (6000, 3)) - o).astype(np.float32)
        h = sc.intersect(o, d)
        o64, d64 = o.astype(np.float64), d.astype(np.float64)
        t, xh, yh = plane_hit(o64, d64, origin.astype(np.float32).astype(np.float64), 2)
        r32 = np.float64(np.float32(radius))
        r2 = r32 ** 2
        # the Disk's box is HALF the radius wide and 0.001 thick (disk.rs:25-28: a kept quirk): only rays that enter it reach Disk::hit
        c = origin.astype(np.float32).astype(np.float64)
        half = np.array([r32 / 2, r32 / 2, np.float64(np.float32(0.001))])
        boxed, box_margin = aabb_hit(o64, d64, (c - half).astype(np.float32).astype(np.float64), (c + half).astype(np.float32).astype(np.float64))
        hit = np.isfinite(t) & (xh * xh + yh * yh <= r2) & boxed
        sure = (~np.isfinite(t) | (np.abs(xh * xh + yh * yh - r2) > 1e-5)) & (box_margin > 1e-5)
        assert ((h["valid"] == 1) == hit)[sure].all()
        both = hit & (h["valid"] == 1) & sure
        outside_box = np.isfinite(t) & (xh * xh + yh * yh <= r2) & ~boxed & sure
        assert both.sum() > 300 and outside_box.sum() > 300 and (h["valid"][outside_box] == 0).all()
        assert np.abs(h["t"][both] - t[both]).max() < 1e-5
        normal = np.zeros((len(t), 3)); normal[:, 2] = 1.0
        if two_sided:
            normal[d64[:, 2] > 0] *= -1.0
        assert np.array_equal(h["normal"][both], normal[both].astype(np.float32))


def test_transformed_instance_agrees_with_a_third_reading(pkg, oracle):
    """Instance::hit (instance.rs:75-116): the ray to local space (origin and direction through the inverse, the direction NOT renormalised, so t is the world's),
    the point back through the forward matrix, the normal through the transpose of the inverse, renormalised."""
    S = pkg.scene
    fwd = S.transform_from_data(scale=(1.5, 0.7, 1.0), rotate=[((0.3, 1.0, 0.2), 40.0)], translate=(0.2, -0.1, 0.4))
    sc = oracle.create_scene(one_primitive(pkg, lambda b, m: b.add_sphere(0.5, (0.0, 0.0, 0.0), m, transform=fwd)))
    M = np.array(fwd, np.float32).astype(np.float64).reshape(4, 4)
    Mi = np.linalg.inv(M)
    rng = np.random.default_rng(11)
    o, d = rays_at(rng, 5000, (0.2, -0.1, 0.4), 0.5)
    h = sc.intersect(o, d)
    o64, d64 = o.astype(np.float64), d.astype(np.float64)
    lo, ld = o64 @ Mi[:3, :3].T + Mi[:3, 3], d64 @ Mi[:3, :3].T
    t, p, n, disc = sphere_hit(lo, ld, np.zeros(3), 0.5)
    sure = np.abs(disc) > 1e-4
    assert ((h["valid"] == 1) == np.isfinite(t))[sure].all()
    both = (h["valid"] == 1) & np.isfinite(t) & sure
    assert both.sum() > 1500
    assert np.abs(h["t"][both] - t[both]).max() < 1e-4
    pw = p[both] @ M[:3, :3].T + M[:3, 3]
    nw = unit(n[both] @ Mi[:3, :3])          # transpose(inverse) * n
    assert np.abs(h["point"][both] - pw).max() < 1e-4
    assert np.abs(h["normal"][both] - nw).max() < 1e-4
    assert np.abs((o64 + t[:, None] * d64)[both] - pw).max() < 1e-4   # t is the world's parameter: the direction was not renormalised


# ------------------------------------------------------------------------------------------------ the watertight triangle
def triangle_hits(o, d, p0, p1, p2):
    """MeshTriangleRef::hit (mesh.rs:67-198) for one ray against many triangles, f64: translate, permute so the dominant axis is z, shear, edge functions,
    the mixed-sign rejection, the determinant and the scaled distance; barycentrics e_i / det; the hit point is b0 p0 + b1 p1 + b2 p2 (mesh.rs:166-170)."""
    ad = np.abs(d)
    kz = 0 if ad[0] >= ad.max() else 1   # max_dimension's ties as f32 decides them are avoided by the rays used here
    if ad[1] >= ad.max():
        kz = 1
    if ad[2] >= ad.max():
        kz = 2
    perm = [(kz + 1) % 3, (kz + 2) % 3, kz]
    dp = d[perm]
    sx, sy, sz = -dp[0] / dp[2], -dp[1] / dp[2], 1.0 / dp[2]
    q0, q1, q2 = ((p - o)[:, perm] for p in (p0, p1, p2))
    x0, y0 = q0[:, 0] + sx * q0[:, 2], q0[:, 1] + sy * q0[:, 2]
    x1, y1 = q1[:, 0] + sx * q1[:, 2], q1[:, 1] + sy * q1[:, 2]
    x2, y2 = q2[:, 0] + sx * q2[:, 2], q2[:, 1] + sy * q2[:, 2]
    e0, e1, e2 = x1 * y2 - y1 * x2, x2 * y0 - y2 * x0, x0 * y1 - y0 * x1
    mixed = ((e0 < 0) | (e1 < 0) | (e2 < 0)) & ((e0 > 0) | (e1 > 0) | (e2 > 0))
    det = e0 + e1 + e2
    ts = e0 * sz * q0[:, 2] + e1 * sz * q1[:, 2] + e2 * sz * q2[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        t = ts / det
        b = np.stack([e0, e1, e2], 1) / det[:, None]
    ok = ~mixed & (det != 0) & (t > 0)
    return np.where(ok, t, np.inf), b, np.minimum(np.minimum(np.abs(e0), np.abs(e1)), np.abs(e2))


def test_mesh_agrees_with_a_third_reading(pkg, oracle):
    """The gem (302 triangles, a transformed instance) as its own scene: closest hit over ALL triangles by the watertight test in f64 against the oracle's BVH walk —
    the same winner (the hit point lies on it), the same distance, the normal the mesh's interpolated one."""
    b = pkg.scene.cornell_gem()
    gem = max(i for i, inst in enumerate(b.instances) if inst.kind == pkg.api.SHAPE_MESH)
    inst = b.instances[gem]
    sc = oracle.create_scene(b)
    mesh = b.meshes[inst.mesh]
    V = np.array(b.vertices, np.float32).astype(np.float64).reshape(-1, 3)[mesh.vertex_offset:mesh.vertex_offset + mesh.vertex_count]
    Fc = np.array(b.indices, np.int64)[mesh.index_offset:mesh.index_offset + 3 * mesh.face_count].reshape(-1, 3)
    M = np.array(list(inst.forward), np.float32).astype(np.float64).reshape(4, 4)
    Mi = np.array(list(inst.reverse), np.float32).astype(np.float64).reshape(4, 4)
    centre = M[:3, 3]
    rng = np.random.default_rng(13)
    n = 1500
    o = (centre + unit(rng.normal(0, 1, (n, 3))) * 0.45).astype(np.float32)
    d = unit(centre + rng.normal(0, 0.03, (n, 3)) - o).astype(np.float32)
    h = sc.intersect(o, d)
    on_gem = (h["valid"] == 1) & (h["instance"] == gem)
    assert on_gem.sum() > 800
    worst_t = worst_p = 0.0
    agree = 0
    for i in np.nonzero(on_gem)[0]:
        lo, ld = o[i].astype(np.float64) @ Mi[:3, :3].T + Mi[:3, 3], d[i].astype(np.float64) @ Mi[:3, :3].T
        t, bary, edge = triangle_hits(lo, ld, V[Fc[:, 0]], V[Fc[:, 1]], V[Fc[:, 2]])
        k = int(np.argmin(t))
        if not np.isfinite(t[k]) or edge[k] < 1e-9:       # on an edge to f64's eye: either neighbour may win in f32
            continue
        agree += 1
        worst_t = max(worst_t, abs(t[k] - h["t"][i]))
        pl = bary[k] @ np.stack([V[Fc[k, 0]], V[Fc[k, 1]], V[Fc[k, 2]]])
        worst_p = max(worst_p, np.abs(pl @ M[:3, :3].T + M[:3, 3] - h["point"][i]).max())
    assert agree > 700
    assert worst_t < 1e-5 and worst_p < 1e-5


# ------------------------------------------------------------------------------------------------ AABB
def test_aabb_rule_agrees_with_a_third_reading(pkg, oracle):
    """AABB::hit (aabb.rs:37-65) decides whether a leaf is entered; a wrong rule would lose hits.  For a lone rect (its box: half the size, 1e-4 thick, rect.rs:60-66)
    and a lone sphere every ray that meets the primitive must be reported — from origins inside the box, on its faces and with zero direction components too."""
    sc = oracle.create_scene(one_primitive(pkg, lambda b, m: b.add_rect((1.0, 1.0), (0.0, 0.0, 0.0), "Z", True, m)))
    o = np.array([[0.2, 0.1, 1.0], [0.2, 0.1, -1.0], [0.5, 0.5, 1.0], [0.2, 0.1, 1e-5], [0.0, 0.0, 2.0], [-0.5, 0.0, 3.0]], np.float32)
    d = np.array([[0, 0, -1], [0, 0, 1], [0, 0, -1], [0, 0, -1], [0, 0, -1], [0, 0, -1]], np.float32)
    h = sc.intersect(o, d)
    assert h["valid"].tolist() == [1, 1, 1, 1, 1, 1]
    assert np.allclose(h["t"], [1.0, 1.0, 1.0, 1e-5, 2.0, 3.0], rtol=1e-6)
    h = sc.intersect(np.array([[0.2, 0.1, 1.0], [2.0, 0.0, 0.0]], np.float32), np.array([[1, 0, 0], [-1, 0, 0]], np.float32))
    assert h["valid"].tolist() == [0, 0]      # parallel to the plane: never (rect.rs:72-75), also inside the box's slab
