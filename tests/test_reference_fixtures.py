"""CPU tier, this container only: the data fixtures and the config / scene files the REFERENCE tree itself holds, read in place from
/root/reference (never copied, never shipped: the whole module is skipped where that tree is absent, i.e. on the GPU box).

SURVEY 8(c) names data/test/{cornell.csv, gold.csv, xenon_lamp.spectra, test.png, test.bmp} — the inputs of the reference's own parse
tests (src/parsing/curves.rs:411-477, src/parsing/texture.rs:302-335).  Those tests only print; what can be pinned here is that the C++
front end (libptscene) reads the very same files the way the reference's loaders are written: every knot of every table comes back
exactly, values between knots stay between them (the interpolant has no overshoot), the images decode to what an independent decoder
written in this file gets.  The second half runs `ptcli --dry-run` over every config and scene file of the reference with the outcome
the reference itself would have: it loads, or it stops at the asset / camera / field the reference tree does not provide.
The oracle stays PARITY UNPINNED: none of this fixes the arithmetic of the un-vendored math crate."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "data", "test")), reason="the reference tree is not on this machine")
HERE = os.path.dirname(os.path.abspath(__file__))
PTCLI = os.path.join(HERE, "..", "rust-pathtracer_amd", "csrc", "ptcli")


@pytest.fixture(scope="module")
def sfmod(pkg):
    m = pkg.scene_file
    m.library()
    return m


def _scene_with_curves(tmp_path, curves_toml, names):
    """A scene whose instances use every curve in `names` (construct_world keeps only the assets its instances reach, mod.rs:216-262)."""
    mats = "".join('[materials.m_%s]\ntype = "DiffuseLight"\nbounce_color = "%s"\nemit_color = "%s"\nsidedness = "Dual"\n' % (n, n, n) for n in names)
    inst = "".join('[[instances]]\nmaterial_name = "m_%s"\n[instances.aggregate]\ntype = "Sphere"\nradius = 0.1\norigin = [%d.0, 0.0, 0.0]\n' % (n, 2 + k) for k, n in enumerate(names))
    text = ('meshes = {}\ntextures = {}\nenv_sampling_probability = 0.5\n[curves]\n' + curves_toml + mats +
            '[environment]\ntype = "Constant"\ncolor = "white"\nstrength = 1.0\n' + inst +
            '[[cameras]]\ntype = "SimpleCamera"\nname = "main"\nlook_from = [0.0, 0.0, 0.0]\nlook_at = [1.0, 0.0, 0.0]\nvfov = 30.0\n')
    p = tmp_path / "scene.toml"
    p.write_text(text)
    return str(p)


def _csv(path):
    rows = [line.split(",") for line in open(path).read().strip().splitlines()[1:]]
    return np.array([[float(v) for v in r] for r in rows], np.float64)


def test_reference_curve_fixtures(sfmod, oracle, tmp_path):
    """curves.rs:411-477: gold.csv through load_ior_and_kappa (micrometres x 1000, Cubic), cornell.csv through load_multiple_csv_rows
    (3 columns, Cubic), xenon_lamp.spectra through load_linear (y x 10, Cubic)."""
    t = os.path.join(REF, "data", "test")
    def csv_curve(name, f, col, extra=""):
        return '%s = { type = "TabulatedCSV", filename = "%s", column = %d, interpolation_mode = "Cubic"%s }\n' % (name, os.path.join(t, f), col, extra)
    lib = (csv_curve("gold_n", "gold.csv", 1, ", domain_mapping = { x_scale = 1000.0 }") + csv_curve("gold_k", "gold.csv", 2, ", domain_mapping = { x_scale = 1000.0 }") +
           csv_curve("cornell_white", "cornell.csv", 1) + csv_curve("cornell_green", "cornell.csv", 2) + csv_curve("cornell_red", "cornell.csv", 3) +
           'xenon = { type = "Linear", filename = "%s", interpolation_mode = "Cubic", domain_mapping = { y_scale = 10.0 } }\n' % os.path.join(t, "xenon_lamp.spectra") +
           'white = { type = "Flat", strength = 1.0 }\n')
    sf = sfmod.SceneFile(_scene_with_curves(tmp_path, lib, ["gold_n", "gold_k", "cornell_white", "cornell_green", "cornell_red", "xenon"]))
    desc = sf.desc
    scene = oracle.create_scene(sf)
    gold, cornell = _csv(os.path.join(t, "gold.csv")), _csv(os.path.join(t, "cornell.csv"))
    xen = open(os.path.join(t, "xenon_lamp.spectra")).read().split()
    x0, step = float(xen[0].rstrip(",")), float(xen[1])
    xenon_y = np.array([float(v) for v in xen[2:]], np.float64)
    cases = {"gold_n": (gold[:, 0].astype(np.float32) * np.float32(1000.0), gold[:, 1]), "gold_k": (gold[:, 0].astype(np.float32) * np.float32(1000.0), gold[:, 2]),
             "cornell_white": (cornell[:, 0], cornell[:, 1]), "cornell_green": (cornell[:, 0], cornell[:, 2]), "cornell_red": (cornell[:, 0], cornell[:, 3])}
    for name, (xs, ys) in cases.items():
        idx = sf.curve(name)
        assert idx >= 0, name
        xs32, ys32 = np.asarray(xs, np.float32), np.asarray(ys, np.float32)
        c = desc.curves[idx]
        assert c.data_count == len(xs32), (name, c.data_count)
        got = scene.curve_eval(idx, xs32)
        assert np.array_equal(got.view(np.uint32), ys32.view(np.uint32)), name          # every knot exactly
        mid = ((xs32[:-1].astype(np.float64) + xs32[1:]) / 2).astype(np.float32)
        gm = scene.curve_eval(idx, mid)
        lo, hi = np.minimum(ys32[:-1], ys32[1:]), np.maximum(ys32[:-1], ys32[1:])
        assert ((gm >= lo - 1e-6) & (gm <= hi + 1e-6)).all(), name                      # no overshoot between knots
    # what the reference's tests print: cornell colours at 520 / 660 nm sit between their neighbours in the table
    for name, col in (("cornell_white", 1), ("cornell_green", 2), ("cornell_red", 3)):
        v = scene.curve_eval(sf.curve(name), np.array([520.0, 660.0], np.float32))
        for lam, got in zip((520.0, 660.0), v):
            k = np.searchsorted(cornell[:, 0], lam)
            assert min(cornell[k - 1, col], cornell[k, col]) - 1e-6 <= got <= max(cornell[k - 1, col], cornell[k, col]) + 1e-6
    xi = sf.curve("xenon")
    assert desc.curves[xi].data_count == len(xenon_y)
    lam = (np.float32(x0) + np.arange(len(xenon_y), dtype=np.float32) * np.float32(step))[:-1]   # (the last sample is the upper bound itself)
    got = scene.curve_eval(xi, lam)
    want = (xenon_y[:-1].astype(np.float32) * np.float32(10.0))
    assert np.allclose(got, want, rtol=2e-6, atol=0), float(np.abs(got - want).max())       # (lambda - x0) / step is not exactly an integer in f32
    assert (scene.curve_eval(xi, np.array([500.0, 500.5], np.float32)) > 0).all()            # curves.rs:474-475 print these two


def _decode_png(path):
    d = open(path, "rb").read()
    assert d[:8] == b"\x89PNG\r\n\x1a\n"
    p, idat, hdr = 8, b"", None
    while p < len(d):
        n, t = struct.unpack(">I", d[p:p + 4])[0], d[p + 4:p + 8]
        body = d[p + 8:p + 8 + n]
        if t == b"IHDR": hdr = struct.unpack(">IIBBBBB", body)
        elif t == b"IDAT": idat += body
        p += 12 + n
    w, h, depth, ctype = hdr[:4]
    assert depth == 8 and ctype in (2, 6)
    ch = 3 if ctype == 2 else 4
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * ch)
    out = np.zeros((h, w * ch), np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        up = out[y - 1] if y else np.zeros(w * ch, np.int32)
        if ft == 0: out[y] = line
        elif ft == 2: out[y] = (line + up) & 255
        else:
            cur = out[y]
            for x in range(w * ch):
                a = cur[x - ch] if x >= ch else 0
                b, c = up[x], (up[x - ch] if x >= ch else 0)
                if ft == 1: v = a
                elif ft == 3: v = (a + b) // 2
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    v = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + v) & 255
    return out.reshape(h, w, ch).astype(np.uint8)


def _decode_bmp8(path):
    d = open(path, "rb").read()
    off, hdr = struct.unpack("<I", d[10:14])[0], struct.unpack("<I", d[14:18])[0]
    w, h = struct.unpack("<ii", d[18:26])
    bits = struct.unpack("<H", d[28:30])[0]
    assert bits == 8 and h > 0
    pal = np.frombuffer(d[14 + hdr:14 + hdr + 1024], np.uint8).reshape(256, 4)[:, 2::-1]   # BGRA -> RGB
    stride = (w + 3) // 4 * 4
    idx = np.frombuffer(d[off:off + stride * h], np.uint8).reshape(h, stride)[::-1, :w]
    return pal[idx]


def test_reference_image_fixtures(sfmod):
    """texture.rs:302-335: parse_bitmap(data/test/test.bmp) (into_luma8 / 255) and parse_rgba(data/test/test.png) (into_rgba8 / 255)."""
    t = os.path.join(REF, "data", "test")
    rgb = _decode_png(os.path.join(t, "test.png"))
    got = sfmod.read_image(os.path.join(t, "test.png"), sfmod.IMAGE_RGBA8)
    assert got.shape == (600, 800, 4)
    want = np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], axis=2).astype(np.float32) / np.float32(255)
    assert np.array_equal(got, want)
    bmp = _decode_bmp8(os.path.join(t, "test.bmp")).astype(np.uint32)
    luma = ((2126 * bmp[..., 0] + 7152 * bmp[..., 1] + 722 * bmp[..., 2]) // 10000).astype(np.float32) / np.float32(255)
    got = sfmod.read_image(os.path.join(t, "test.bmp"), sfmod.IMAGE_GREY8)
    assert got.shape == (600, 800) and np.array_equal(got, luma)
    # the two numbers the reference's tests print
    assert 0.0 < float(got.mean()) < 1.0 and float(got.max()) <= 1.0


# What each of the reference's own config files does when handed to ptcli --dry-run with --root /root/reference, and why that is what
# the reference would do with its own tree (it ships neither OBJ meshes other than the four gems / monkey / prism, nor any HDRI).
CONFIGS = {
    "config.toml": "could not find obj file or mtl file data/meshes/cornell_box.obj",                      # parsing/meshes.rs: the OBJ is not in the tree
    "config_test_cornell_box.toml": "could not find obj file or mtl file data/meshes/cornell_box.obj",
    "config_test_blackbox.toml": "missing field `medium_aware`",                                            # serde: IntegratorType::PT needs it (config.rs:60-75)
    "config_test_candela_calibration.toml": "camera `candela` named by the render settings is not in the scene",   # the scene names its camera "main"
    "config_test_lighting_north.toml": "camera `camera` named by the render settings is not in the scene",
    "config_test_whitefurnace.toml": "camera `camera` named by the render settings is not in the scene",
    "config_test_lighting_hdri.toml": "could not find file at data/hdri/machine_shop_03_4k.hdr",            # no data/hdri directory in the tree
    "raymarch_config.toml": "renderer type Preview is not supported",                                       # out of scope
}
SCENES_THAT_LOAD = ["candela_calibration", "cornell_box_parallel_prism", "cornell_box_single_orb_caustic", "sun_test", "test_blackbox", "test_lighting_north",
                    "test_nee_sphere", "test_rtiow_scene_2", "test_sampling_methods", "white_furnace"]
SCENES_THAT_STOP = {
    "caustic_test_scene": "data/meshes/caustic_test.obj", "cornell_box": "data/meshes/cornell_box.obj", "cornell_box_lenses": "data/meshes/lenses.obj",
    "test_menger": "data/meshes/menger_sponge.obj", "tower": "data/meshes/tower.obj", "trippy_glass_cube": "data/meshes/trippy_glass_cube.obj",
    "cornell_box_diamond_arrangement": "data/hdri/kiara_1_dawn_8k.hdr", "cornell_box_diamond_gem": "data/hdri/kiara_1_dawn_8k.hdr", "cornell_box_medium": "data/hdri/kiara_1_dawn_8k.hdr",
    "cornell_box_metals_and_dielectrics": "data/hdri/kiara_1_dawn_8k.hdr", "cornell_box_textured_walls": "data/hdri/kiara_1_dawn_8k.hdr",
    "cornell_box_hdri_test": "data/hdri/machine_shop_03_4k.hdr", "hdri_test": "data/hdri/machine_shop_03_4k.hdr", "test_veach_mis": "data/hdri/machine_shop_03_4k.hdr",
    "hdri_test_2": "data/hdri/kloofendal_43d_clear_puresky_1k.exr", "test_prism": "data/hdri/kloofendal_43d_clear_puresky_1k.exr",
    "metals_spectral_breakdown": "data/hdri/autumn_park_8k.hdr", "test_bokeh": "data/hdri/autumn_park_8k.hdr", "test_nonuniform_scale": "data/hdri/autumn_park_8k.hdr",
    "raymarch": "data/hdri/sunny_vondelpark_8k.hdr",
}


def _dry_run(*args):
    return subprocess.run([PTCLI, "--root", REF, "--dry-run"] + list(args), capture_output=True, text=True)


def test_every_reference_config_has_its_expected_outcome():
    found = sorted(f for f in os.listdir(os.path.join(REF, "data")) if f.endswith(".toml") and "config" in f)
    assert found == sorted(CONFIGS), found
    for name, why in CONFIGS.items():
        r = _dry_run("--config", os.path.join(REF, "data", name))
        assert r.returncode != 0 and why in r.stderr, (name, r.stderr)
        assert not os.path.exists(os.path.join(REF, why.split()[-1])) or "data/" not in why      # the asset really is not in the reference tree


def test_every_reference_scene_has_its_expected_outcome(pkg):
    config = os.path.join(os.path.dirname(pkg.__file__), "data", "config_cornell_c1.toml")   # a PT config whose camera_id is "main"
    found = sorted(f[:-5] for f in os.listdir(os.path.join(REF, "data", "scenes")) if f.endswith(".toml"))
    assert found == sorted(SCENES_THAT_LOAD + list(SCENES_THAT_STOP)), found
    for name in SCENES_THAT_LOAD:
        r = _dry_run("--config", config, "--scene", os.path.join(REF, "data", "scenes", name + ".toml"))
        assert r.returncode == 0 and "constructing renderer" in r.stdout, (name, r.stderr)
    for name, asset in SCENES_THAT_STOP.items():
        r = _dry_run("--config", config, "--scene", os.path.join(REF, "data", "scenes", name + ".toml"))
        assert r.returncode != 0 and asset in r.stderr, (name, r.stderr)
        assert not os.path.exists(os.path.join(REF, asset)), asset


@pytest.mark.parametrize("name", SCENES_THAT_LOAD)
def test_reference_scenes_that_load_render_on_the_oracle(sfmod, oracle, pkg, name):
    """The reference's own scene files, through the C++ front end into the oracle: a small film comes out finite and, where the scene
    has any light, not black.  (mediums in the file are ignored with a warning: the PT path of this repository has none.)"""
    config = sfmod.Config(os.path.join(os.path.dirname(pkg.__file__), "data", "config_cornell_c1.toml"))
    sfmod.set_root(REF)
    try:
        sf = sfmod.SceneFile(os.path.join(REF, "data", "scenes", name + ".toml"), config)
    finally:
        sfmod.set_root()
    scene = oracle.create_scene(sf)
    film, prof = scene.render(pkg.api.render_desc(24, 16, 4, 4, light_samples=2, seed=2))
    assert prof.camera_rays == 24 * 16 * 4
    finite = np.isfinite(film)
    assert finite.mean() > 0.95
    lit = np.where(finite, film, 0.0)[..., :3].max() > 0.0
    # black for a reason: two caustic set-ups light the camera only through a prism / an orb from a narrow sharp light (nothing at 4 spp
    # and depth 4), and test_lighting_north.toml puts its camera inside the opaque unit sphere
    assert lit or name in ("cornell_box_parallel_prism", "cornell_box_single_orb_caustic", "test_lighting_north"), name


@pytest.mark.parametrize("name", sorted(SCENES_THAT_LOAD))
def test_reference_scene_builders_are_the_reference_scene_files(sfmod, oracle, pkg, name):
    """The GPU tier renders the reference tree's self-contained scenes from scene builders (pkg.scene.REFERENCE_TREE_SCENES — /root/reference does not
    exist on the GPU box).  Here, where it does: the reference's OWN scene file through the C++ front end and the builder give the oracle the same
    film bit for bit and the same counters — the builders are those scenes, value for value."""
    builders = dict(pkg.scene.REFERENCE_TREE_SCENES, white_furnace=pkg.scene.white_furnace)
    assert sorted(builders) == sorted(SCENES_THAT_LOAD)
    config = sfmod.Config(os.path.join(os.path.dirname(pkg.__file__), "data", "config_cornell_c1.toml"))
    sfmod.set_root(REF)
    try:
        sf = sfmod.SceneFile(os.path.join(REF, "data", "scenes", name + ".toml"), config)
    finally:
        sfmod.set_root()
    b = builders[name]()
    assert sf.desc.instance_count == len(b.instances) and sf.desc.env_sampling_probability == np.float32(b.env_sampling_probability)
    kw = {"wavelength": (555.0, 560.0), "only_direct": True, "light_samples": 1} if name == "candela_calibration" else {"light_samples": 2}
    rd = pkg.api.render_desc(40, 32, 6, 6, seed=3, **kw)
    film_f, prof_f = oracle.create_scene(sf).render(rd)
    film_b, prof_b = oracle.create_scene(b).render(rd)
    assert np.array_equal(film_f.view(np.uint32), film_b.view(np.uint32))
    assert (prof_f.camera_rays, prof_f.bounce_rays, prof_f.shadow_rays, prof_f.env_hits) == (prof_b.camera_rays, prof_b.bounce_rays, prof_b.shadow_rays, prof_b.env_hits)


def test_showcase_cornell_box_chromaticities(oracle, pkg):
    """A loose STATISTICAL pin of the oracle against the only pixels in the tree that the reference itself produced:
    showcase/cornell_box_1080p.png.  Its render settings are unknown (exposure, tone mapper, sample count, the author's own cornell_box.obj),
    so nothing absolute can be compared; what survives those unknowns is the chromaticity of large flat patches — the Cornell light's
    spectrum times each wall's reflectance curve, through the colour-matching functions, the XYZ -> sRGB matrix and the OETF.  The oracle
    renders this repository's C2 scene (same camera, light and library curves as data/scenes/cornell_box.toml), the film goes through
    output_film, and the linearised patch colours must have the showcase's chromaticities within a generous tolerance.  This catches a wrong
    unit, matrix, curve table or wall assignment; it does NOT decide between the close alternatives of profiles/r3_oracle_sensitivity.md
    (those move chromaticities by < 0.01).  Parity stays unpinned."""
    import oracle_loader
    show = _decode_png(os.path.join(REF, "showcase", "cornell_box_1080p.png"))[..., :3].astype(np.float64) / 255.0
    assert show.shape[:2] == (1080, 1080)
    sc = oracle.create_scene(pkg.scene.cornell_box())
    film, _ = oracle_loader.render_mt(oracle, sc, pkg.api.render_desc(120, 120, 48, 8, seed=3), os.cpu_count() or 4)
    rgba, _ = oracle.output_film(film, tonemap=pkg.api.TONEMAP_CLAMP, exposure=3.0)   # (bright enough for the 8-bit quantisation not to matter)
    ours = rgba[..., :3].astype(np.float64) / 255.0

    def lin(c):
        return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)

    def chroma(img, cx, cy, r=0.03):
        h, w, _ = img.shape
        p = lin(img[int((cy - r) * h):int((cy + r) * h), int((cx - r) * w):int((cx + r) * w)].reshape(-1, 3).mean(0))
        return p[0] / p.sum(), p[1] / p.sum()
    # patch centres as fractions of the (square) image; tolerance on (r, g) chromaticity
    patches = {"back wall": (0.5, 0.35, 0.05), "right wall (red)": (0.93, 0.5, 0.05), "left wall (green)": (0.08, 0.5, 0.10), "floor": (0.75, 0.95, 0.06),
               "ceiling": (0.3, 0.04, 0.05), "tall block front": (0.33, 0.62, 0.06), "short block top": (0.62, 0.70, 0.07)}
    for name, (cx, cy, tol) in patches.items():
        a, b = chroma(show, cx, cy), chroma(ours, cx, cy)
        assert abs(a[0] - b[0]) <= tol and abs(a[1] - b[1]) <= tol, (name, a, b)
    # the layout itself: green on the left, red on the right, a warm white in between (as the showcase has it)
    gl, rr, bw = chroma(ours, 0.08, 0.5), chroma(ours, 0.93, 0.5), chroma(ours, 0.5, 0.35)
    assert gl[1] > gl[0] and rr[0] > 0.9 and rr[0] > bw[0] > gl[0]


def test_showcase_moissanite_gem_layout(oracle, pkg):
    """A second loose pin against reference-produced pixels (round-3 verdict, item 7): showcase/moissanite_gem_1080p.png, the look of BASELINE's C3 scene
    (data/scenes/cornell_box_diamond_gem.toml).  Its lighting cannot be compared — the image shows sharp caustics all over the room, i.e. it was made with
    the reference's light-tracing / bidirectional integrators and an HDRI behind the open front, neither of which the PT path has — so this test compares
    what does not depend on the integrator: WHERE things are.  The oracle's own camera (pt_camera_samples: jitter + Camera::get_ray) and its World::hit
    give an id buffer of the scene at 216 x 216; the showcase, cut by that id buffer, must show
      * the light's underside at the top centre (x 0.417-0.583, y 0.120-0.153) as the one black quad of a bright ceiling (SharpLight: cos^41, black at
        grazing angles),
      * the green wall (y = -1) on the LEFT and the red one (y = +1) on the right, every interior pixel of each dominated by its colour — camera
        handedness (u = -(v_up x w)), vfov, aspect and the wall materials,
      * the gem's silhouette: the image's fine structure (luminance spread inside 5 x 5 pixel blocks) is concentrated inside the projected mesh.  With
        the gem where the scene file puts it today (translate z = -0.7, the culet below the floor) the silhouette does NOT fit; raised to z = -0.35 —
        culet on the image row 0.815 the showcase shows — it does: the image predates that edit of the file.  Mesh scale 0.5, the 302-triangle
        brilliant_diamond.obj, transform order and the thin-lens camera are what fit.
    Loose and geometric; parity stays unpinned."""
    show = _decode_png(os.path.join(REF, "showcase", "moissanite_gem_1080p.png"))[..., :3].astype(np.float64) / 255.0
    assert show.shape[:2] == (1080, 1080)
    lin = np.where(show <= 0.04045, show / 12.92, ((show + 0.055) / 1.055) ** 2.4)
    N, f = 216, 5
    rgb = lin.reshape(N, f, N, f, 3).mean(axis=(1, 3))
    Y = 0.2126 * lin[..., 0] + 0.7152 * lin[..., 1] + 0.0722 * lin[..., 2]
    Ymean, Yspread = Y.reshape(N, f, N, f).mean(axis=(1, 3)), Y.reshape(N, f, N, f).std(axis=(1, 3))

    def id_buffer(gem_z):
        sc = oracle.create_scene(pkg.scene.cornell_gem(gem_z=gem_z))
        o, d, _ = sc.camera_samples(pkg.api.render_desc(N, N, 1, 1), np.arange(N * N, dtype=np.uint32), np.zeros(N * N, np.uint32))
        hits = sc.intersect(o, d)
        return np.where(hits["valid"] == 1, hits["instance"].astype(np.int64), -1).reshape(N, N)

    def shrink(mask, r):
        for _ in range(r):
            mask = mask & np.roll(mask, 1, 0) & np.roll(mask, -1, 0) & np.roll(mask, 1, 1) & np.roll(mask, -1, 1)
        return mask
    LIGHT, CEILING, FLOOR, RED, GREEN, BACK, GEM = range(7)   # instance order of scene.cornell_gem = the scene file's
    ids = id_buffer(-0.35)
    assert (ids >= 0).all()                                   # the room fills the frame (vfov 27.8 from x = -5)
    # the light
    ys, xs = np.where(ids == LIGHT)
    assert abs(xs.min() / N - 0.417) < 0.01 and abs((xs.max() + 1) / N - 0.583) < 0.01 and abs(ys.min() / N - 0.120) < 0.01 and abs((ys.max() + 1) / N - 0.153) < 0.01
    assert Ymean[shrink(ids == LIGHT, 1)].mean() < 0.05 * Ymean[shrink(ids == CEILING, 2)].mean()
    # the walls
    g, r = rgb[shrink(ids == GREEN, 2)], rgb[shrink(ids == RED, 2)]
    assert ((g[:, 1] > g[:, 0]) & (g[:, 1] > g[:, 2])).mean() > 0.97 and ((r[:, 0] > r[:, 1]) & (r[:, 0] > r[:, 2])).mean() > 0.97
    assert np.where(ids == GREEN)[1].max() < N * 0.17 and np.where(ids == RED)[1].min() > N * 0.83        # green left, red right
    # the gem
    def contrast(ids):
        gem = ids == GEM
        ring = ~gem & ~shrink(~gem, 6) & (ids == BACK)        # the back wall within six cells of the silhouette
        return np.median(Yspread[shrink(gem, 1)]) / np.median(Yspread[ring])
    ys, xs = np.where(ids == GEM)
    assert abs((ys.max() + 1) / N - 0.815) < 0.01 and abs(xs.min() / N - 0.296) < 0.01 and abs((xs.max() + 1) / N - 0.704) < 0.01
    assert contrast(ids) > 2.0, contrast(ids)
    assert contrast(id_buffer(-0.7)) < 1.3                    # (where the scene file puts the gem today: not what the image shows)
