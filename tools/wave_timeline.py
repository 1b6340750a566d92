#!/usr/bin/env python3
"""Runs on the GPU box with a measurement build (tools/build_variant.sh tl -DPT_TIMELINE; PT_AMD_LIBRARY=variants/tl.so): renders one frame of a
workload and prints, for every launch, when its waves began and ended (100 MHz clock): the launch's span, the waves resident on
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
L, B = 32, 16384
FAMILIES = {"shadow": "parked rays resumed", "extend": "parked rays resumed", "shade": "surface vertices shaded"}
for f in FAMILIES:
    fn = getattr(raw, "pt_debug_timeline_" + f)
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32)]
raw.pt_debug_long_walks.restype = ctypes.c_int
raw.pt_debug_long_walks.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
n = ctypes.c_uint32(0)
rays = np.zeros(1 + 4096 * 12, dtype=np.float32)
scene.render(rd)   # (warm-up frame: its records are dropped)
for f in FAMILIES:
    assert getattr(raw, "pt_debug_timeline_" + f)(None, 0, ctypes.byref(n)) == 0
assert raw.pt_debug_long_walks(rays.ctypes.data, rays.nbytes) == 0
scene.render(rd)
assert raw.pt_debug_long_walks(rays.ctypes.data, rays.nbytes) == 0
nr = int(rays[:1].view(np.uint32)[0])
rr = rays[1:].reshape(-1, 12)[: min(nr, 4096)]
print("light-sample walks of more than 1500 box tests:", nr)
for r in rr[np.argsort(-rr[:, 8])][:10]:
    print("   steps %6d stop %d bound %g  o (%g %g %g) d (%g %g %g)  local o.x %g d.x %g  closest %g" % (r[8], r[7], r[6], r[0], r[1], r[2], r[3], r[4], r[5], r[9], r[10], r[11]))
for f, unit in FAMILIES.items():
    buf = np.zeros((L, B, 4, 4), dtype=np.uint64)
    assert getattr(raw, "pt_debug_timeline_" + f)(buf.ctypes.data, buf.nbytes, ctypes.byref(n)) == 0
    print("== %s kernels: %d launches (work = %s)" % (f, n.value, unit))
    for l in range(min(n.value, L)):
        t = buf[l].reshape(-1, 4)
        t = t[t[:, 0] != 0]
        if len(t) == 0:
            continue
        t0, t1 = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
        lo, hi = t0.min(), t1.max()
        span = max(1.0, (hi - lo) / 100.0)   # us
        life = (t1 - t0) / 100.0
        busy = t[:, 3] > 0
        print("launch %2d: span %8.1f us, waves %6d, resident on average %.2f per SIMD; waves with work %5.1f %%, their lifetime %.1f us on average, max %.1f; items per segment avg %.0f max %d; work %d" % (
            l, span, len(t), life.sum() / span / 1024.0, 100.0 * busy.mean(), life[busy].mean() if busy.any() else 0.0, life.max(), t[:, 2].mean(), t[:, 2].max(), t[:, 3].sum()))
        slices = np.linspace(lo, hi, 11)
        res = [(np.minimum(t1, b).clip(a).sum() - np.maximum(t0, a).clip(None, b).sum()) / max(1.0, (b - a)) for a, b in zip(slices[:-1], slices[1:])]
        print("           waves resident per tenth of the span:", " ".join("%.0f" % r for r in res))
        order = np.argsort(-life)[:4]
        print("           longest waves (us, items, work):", " ".join("(%.0f, %d, %d)" % (life[i], t[i, 2], t[i, 3]) for i in order))
