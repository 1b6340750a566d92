#!/usr/bin/env python3
"""Differential soak on the GPU over seeds of the BASELINE scenes (C2 / C3 / C4 / C5 shapes at 320x240): engine vs oracle at matched seeds,
films within the parity bars and ray counters equal.  usage: tools/seed_soak.py <first seed> <count>"""
import importlib
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_loader  # noqa: E402
import parity_suite as ps  # noqa: E402

pkg = importlib.import_module("rust-pathtracer_amd")
engine, oracle = pkg.load(), oracle_loader.load(pkg)
first, count = int(sys.argv[1]), int(sys.argv[2])
cases = [("cornell_box", 8, dict(light_samples=2)), ("cornell_gem", 12, dict(light_samples=2)), ("hdri_test", 4, dict(light_samples=6)),
         ("cornell_box", 8, dict(light_samples=2, hero_wavelengths=4)), ("mixed_primitives", 6, dict(light_samples=3)), ("sun_test", 5, dict(light_samples=2))]
scenes = {name: (engine.create_scene(pkg.scene.SCENES[name]()), oracle.create_scene(pkg.scene.SCENES[name]())) for name in {c[0] for c in cases}}
bad = []
for seed in range(first, first + count):
    for name, mb, kw in cases:
        try:
            rd = pkg.api.render_desc(320, 240, 4, mb, seed=seed, **kw)
            film, prof = scenes[name][0].render(rd)
            ref, rprof = scenes[name][1].render(rd)
            ps.check_film(film, ref, prof, rprof)
        except Exception as e:  # noqa: BLE001
            bad.append((seed, name, repr(e)[:200]))
print("seeds", first, "..", first + count - 1, "x", len(cases), "cases; failures:", len(bad), bad[:10])
sys.exit(1 if bad else 0)
