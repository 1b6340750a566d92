#!/usr/bin/env python3
"""Exploration for the second loose pin (round-3 verdict, item 7): the engine's render of the C3 scene — bit-level the oracle's — at the
showcase's framing, written as PNGs next to showcase/moissanite_gem_1080p.png's layout.  GPU box; writes gpurun_out/showcase_gem_*.png/.npy."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("rust-pathtracer_amd")
engine = pkg.load()
out = os.path.join(ROOT, "gpurun_out")
N, spp = int(sys.argv[1]) if len(sys.argv) > 1 else 540, int(sys.argv[2]) if len(sys.argv) > 2 else 2048
for tag, z in (("file", -0.7), ("raised", -0.35)):
    b = pkg.scene.cornell_gem(gem_z=z)
    film, prof = engine.create_scene(b).render(pkg.api.render_desc(N, N, spp, 12, seed=3))
    np.save(os.path.join(out, "showcase_gem_%s.npy" % tag), film.astype(np.float16))
    for ex in (0.0, 3.0):
        rgba, _ = engine.output_film(film, tonemap=pkg.api.TONEMAP_CLAMP, exposure=ex)
        engine.write_png(os.path.join(out, "showcase_gem_%s_ev%d.png" % (tag, int(ex))), rgba)
    print(tag, prof.camera_rays, prof.seconds)
