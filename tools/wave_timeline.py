#!/usr/bin/env python3
"""Runs on the GPU box with a measurement build (tools/build_variant.sh tl -DPT_TIMELINE; PT_AMD_LIBRARY=variants/tl.so): renders one frame of a
workload and prints, for every k_shadow_parked launch, when its waves began and ended (100 MHz clock): the launch's span, the waves resident on
average (sum of lifetimes / span / wave slots at the kernel's occupancy), the share of waves that had anything to do, the longest waves with
their segment's item count and resumed rays, and how the resident waves thin out over the span (ten slices).
usage: tools/wave_timeline.py <scene> <width> <height> <spp> <max_bounces> <light_samples>"""
import ctypes
import importlib
import os
import sys

import numpy as np

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
pkg = importlib.import_module("rust-pathtracer_amd")
name, w, h, spp, mb, ls = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
lib = pkg.load()
scene = lib.create_scene(pkg.scene.SCENES[name]())
rd = pkg.api.render_desc(w, h, spp, mb, light_samples=ls)
raw = ctypes.CDLL(pkg.LIBRARY_PATH)
raw.pt_debug_timeline.restype = ctypes.c_int
raw.pt_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32)]
L, B = 32, 16384
n = ctypes.c_uint32(0)
scene.render(rd)
assert raw.pt_debug_timeline(None, 0, ctypes.byref(n)) == 0   # (warm-up frame dropped)
scene.render(rd)
RAYS = (1 + 4096 * 12) * 4
whole = np.zeros(L * B * 16 * 8 + RAYS + 8, dtype=np.uint8)
assert raw.pt_debug_timeline(whole.ctypes.data, whole.nbytes, ctypes.byref(n)) == 0
buf = whole[: L * B * 16 * 8].view(np.uint64).reshape(L, B, 4, 4)
rays = whole[L * B * 16 * 8: L * B * 16 * 8 + RAYS]
nr = int(rays[:4].view(np.uint32)[0])
rr = rays[4:].view(np.float32).reshape(-1, 12)[: min(nr, 4096)]
print("walks of more than 1500 box tests:", nr)
for r in rr[np.argsort(-rr[:, 8])][:40]:
    print("   steps %6d stop %d bound %g  o (%g %g %g) d (%g %g %g)  local o.x %g d.x %g  closest %g" % (r[8], r[7], r[6], r[0], r[1], r[2], r[3], r[4], r[5], r[9], r[10], r[11]))
print("launches", n.value)
for l in range(min(n.value, L)):
    t = buf[l].reshape(-1, 4)
    t = t[t[:, 0] != 0]
    if len(t) == 0:
        continue
    t0, t1 = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
    lo, hi = t0.min(), t1.max()
    span = (hi - lo) / 100.0   # us
    life = (t1 - t0) / 100.0
    busy = t[:, 3] > 0
    print("launch %2d: span %8.1f us, waves %6d, resident on average %.2f of 4096 slots (%.2f per SIMD); waves that resumed rays %5.1f %%, their lifetime %.1f us on average, max %.1f; "
          "items per segment avg %.0f max %d; resumed rays %d" % (l, span, len(t), life.sum() / span, life.sum() / span / 1024.0, 100.0 * busy.mean(), life[busy].mean() if busy.any() else 0.0, life.max(),
                                                          t[:, 2].mean(), t[:, 2].max(), t[:, 3].sum()))
    slices = np.linspace(lo, hi, 11)
    res = [int(((t0 < b) & (t1 > a)).sum() * 0 + np.minimum(t1, b).clip(a).sum() - np.maximum(t0, a).clip(None, b).sum()) / max(1.0, (b - a)) for a, b in zip(slices[:-1], slices[1:])]
    print("           waves resident per tenth of the span:", " ".join("%.0f" % r for r in res))
    order = np.argsort(-life)[:5]
    print("           longest waves (us, items, resumed):", " ".join("(%.0f, %d, %d)" % (life[i], t[i, 2], t[i, 3]) for i in order))
    # how much of the resumed-ray work sits in the heaviest tenth of the waves
    rr = np.sort(t[:, 3].astype(np.int64))[::-1]
    print("           resumed rays in the heaviest 1 %% / 10 %% of the waves: %.1f %% / %.1f %%" % (100.0 * rr[: max(1, len(rr) // 100)].sum() / max(1, rr.sum()), 100.0 * rr[: max(1, len(rr) // 10)].sum() / max(1, rr.sum())))
