"""Generates tests/golden/*.npz with the CPU oracle (oracle/libptref.so).

The reference cannot run here (no Rust toolchain) and its tests hold no numeric vectors for this path
(SURVEY.md §4), so the committed golden vectors pin the ORACLE: films, closest hits and material samples at
fixed seeds.  The oracle and the HIP engine are both checked against them (tests/test_golden.py).
Run:  python tools/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def golden_rays(scene_name, n, seed):
    """Deterministic ray set aimed at the scene (shared by the generator and the tests)."""
    rng = np.random.default_rng(seed)
    if scene_name.startswith("hdri"):
        o = np.tile(np.array([[-5.0, 0.3, 0.4]], np.float32), (n, 1)) + rng.normal(0, 0.05, (n, 3)).astype(np.float32)
        t = rng.normal(0, 1.0, (n, 3)) + np.array([0.0, 0.6, -0.1])
    elif scene_name == "cornell_box":
        o = np.tile(np.array([[-0.8, 0.278, 0.273]], np.float32), (n, 1)) + rng.normal(0, 0.01, (n, 3)).astype(np.float32)
        t = np.stack([rng.uniform(0.0, 0.56, n), rng.uniform(-0.05, 0.6, n), rng.uniform(-0.05, 0.6, n)], axis=1)
    elif not scene_name.startswith("hdri"):
        o = np.tile(np.array([[-5.0, 0.3, 0.8]], np.float32), (n, 1)) + rng.normal(0, 0.05, (n, 3)).astype(np.float32)
        t = rng.normal(0, 1.2, (n, 3))
    d = t - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    # second half: rays started inside the scene in random directions
    h = n // 2
    if scene_name == "cornell_box":
        o[h:] = np.stack([rng.uniform(0.02, 0.54, n - h), rng.uniform(0.02, 0.54, n - h), rng.uniform(0.02, 0.54, n - h)], axis=1)
    else:
        o[h:] = rng.normal(0, 0.8, (n - h, 3))
    v = rng.normal(0, 1, (n - h, 3)); d[h:] = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    return np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)


def material_inputs(n, seed):
    from util import unit_sphere
    rng = np.random.default_rng(seed)
    wi = unit_sphere(rng, n)
    lam = rng.uniform(380, 750, n).astype(np.float32)
    s2 = rng.random((n, 2), dtype=np.float32)
    wo = unit_sphere(rng, n)
    return lam, wi, s2, wo


GOLDEN_RENDERS = {
    # name: (scene, width, height, spp, max_bounces, light_samples, seed[, hero_wavelengths[, medium_aware]])
    "cornell_64x64_4spp": ("cornell_box", 64, 64, 4, 4, 2, 1),
    "gem_48x32_6spp": ("cornell_gem", 48, 32, 6, 12, 2, 1),
    "mixed_40x40_12spp": ("mixed_primitives", 40, 40, 12, 6, 3, 7),
    "furnace_24x24_16spp": ("white_furnace", 24, 24, 16, 8, 6, 3),
    "hdri_32x32_8spp": ("hdri_small", 32, 32, 8, 4, 6, 2),
    "cornell_hero_48x48_6spp": ("cornell_box", 48, 48, 6, 8, 2, 4, 4),
    "fog_48x32_8spp_medium": ("fog_ball", 48, 32, 8, 8, 2, 3, 1, True),   # the medium-aware walk
}


def main():
    pkg = importlib.import_module("rust-pathtracer_amd")
    import oracle_loader
    ora = oracle_loader.load(pkg)
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    only = sys.argv[1:]   # names to (re)generate; default all
    for name, cfg in GOLDEN_RENDERS.items():
        if only and name not in only:
            continue
        scene, w, h, spp, mb, ls, seed = cfg[:7]
        hero = cfg[7] if len(cfg) > 7 else 1
        medium = cfg[8] if len(cfg) > 8 else False
        sc = ora.create_scene(pkg.scene.SCENES[scene]())
        film, prof = sc.render(pkg.api.render_desc(w, h, spp, mb, light_samples=ls, seed=seed, hero_wavelengths=hero, medium_aware=medium))
        np.savez_compressed(os.path.join(out, name + ".npz"), film=film,
                            counters=np.array([prof.camera_rays, prof.bounce_rays, prof.shadow_rays, prof.env_hits], np.uint64))
        print(name, film[..., :3].mean(axis=(0, 1)), prof.bounce_rays, prof.shadow_rays, prof.env_hits)
    if only:
        return
    for scene in ("cornell_box", "mixed_primitives", "cornell_gem"):
        b = pkg.scene.SCENES[scene]()
        sc = ora.create_scene(b)
        o, d = golden_rays(scene, 4096, 11)
        hits = sc.intersect(o, d)
        np.savez_compressed(os.path.join(out, "hits_%s.npz" % scene), hits=hits)
        print(scene, "hit fraction", hits["valid"].mean())
        lam, wi, s2, wo = material_inputs(1024, 5)
        mats = {}
        for mname, mid in b.material_ids.items():
            idx = mid & 0xFFFF
            f, wo_s, pdf = sc.bsdf_sample(idx, lam, wi, s2)
            f2, pdf2 = sc.bsdf_eval(idx, lam, wi, wo)
            e = sc.emission(idx, lam, wi)
            mats[mname] = np.concatenate([f[:, None], wo_s, pdf[:, None], f2[:, None], pdf2[:, None], e[:, None]], axis=1).astype(np.float32)
        np.savez_compressed(os.path.join(out, "materials_%s.npz" % scene), **mats)


if __name__ == "__main__":
    main()
