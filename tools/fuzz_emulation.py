#!/usr/bin/env python3
"""Differential soak on the CPU: the emulated lane logic (tests/host_emulation) vs the oracle over seeded random scenes (tests/fuzz_scenes.py) — what tools/fuzz_soak.py does
on the GPU, for the build container.  usage: tools/fuzz_emulation.py <first seed> <count> [width height spp] [PTEMU_FLAGS]   (PT_FUZZ_SWITCH=PTEMU_NO_CONVEX: each scene also with that switch, bit for bit)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_scenes  # noqa: E402
import oracle_loader  # noqa: E402
import parity_suite as ps  # noqa: E402

pkg = importlib.import_module("rust-pathtracer_amd")
first, count = int(sys.argv[1]), int(sys.argv[2])
W, H, SPP = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (40, 32, 3)
if len(sys.argv) > 6:
    os.environ["PTEMU_FLAGS"] = sys.argv[6]
emu = pkg.api.Library(os.path.join(ROOT, "tests", "host_emulation", "libptemu.so"), "ptemu_", optional=("render_device", "device_info"))
oracle = oracle_loader.load(pkg)
SWITCH = os.environ.get("PT_FUZZ_SWITCH", "")   # (several: comma-separated) e.g. PT_AMD_NO_CONVEX (GPU) / PTEMU_NO_CONVEX, PTEMU_NO_INSIDE (emulation)
bad = []
for seed in range(first, first + count):
    try:
        b = fuzz_scenes.random_scene(seed)
        o, d = fuzz_scenes.random_rays(seed, 1 << 11)
        se, so = emu.create_scene(b), oracle.create_scene(b)
        ps.assert_hits_equal(se.intersect(o, d), so.intersect(o, d))
        rd = pkg.api.render_desc(W, H, SPP, 6, light_samples=int(1 + seed % 3), seed=seed, hero_wavelengths=4 if seed % 5 == 0 and not fuzz_scenes.medium_aware(seed) else 1, medium_aware=fuzz_scenes.medium_aware(seed))
        film, prof = se.render(rd)
        ref, rprof = so.render(rd)
        ps.check_film(film, ref, prof, rprof)
        for switch in [w for w in SWITCH.split(",") if w]:
            # the same render with a switch of the library set (read when the scene is created): the same BITS — a check of every decision the switch governs, not only of
            # those that move the film by more than the parity bar
            os.environ[switch] = "1"
            try:
                film2, prof2 = emu.create_scene(b).render(rd)
            finally:
                del os.environ[switch]
            assert np.array_equal(film.view(np.uint32), film2.view(np.uint32)), ("bits differ under " + switch, float(np.abs(film - film2).max()))
            assert (prof.bounce_rays, prof.shadow_rays, prof.env_hits) == (prof2.bounce_rays, prof2.shadow_rays, prof2.env_hits), "counters differ under " + switch
    except Exception as e:  # noqa: BLE001
        bad.append((seed, repr(e)[:200]))
stops = int(se.library._debug_scene_info(se.handle, 18)) if count else 0   # (sweeps ended by mesh_walk's `inside` rule in this process: that the soak reached the rule at all)
print("seeds", first, "..", first + count - 1, "inside stops:", stops, "failures:", len(bad), bad[:5])
sys.exit(1 if bad else 0)
