#!/usr/bin/env python3
"""Host-side experiment: which light-sample rays of C3 still WALK the gem's BVH (after the inner balls and the face-diagonal slabs), and what would a convex-mesh
certificate decide for them?  Renders a small frame on the emulated lane logic with a record of every mesh walk (tools/light_walks.cpp).
usage: tools/light_walks.py [scene] [width] [spp]"""
import ctypes
import importlib
import os
import subprocess
import sys

import numpy as np

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
pkg = importlib.import_module("rust-pathtracer_amd")
lib = os.path.join(R, "tests", "host_emulation", "libptlightwalks.so")
srcs = [os.path.join(R, "tools", "light_walks.cpp"), os.path.join(R, "rust-pathtracer_amd", "csrc", "pt_scene_host.cpp"), os.path.join(R, "rust-pathtracer_amd", "csrc", "pt_plan.cpp")]
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function", "-o", lib] + srcs)
emu = pkg.api.Library(lib, "ptemu_", optional=("render_device", "device_info", "render_multi", "device_count"))
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_gem"
width = int(sys.argv[2]) if len(sys.argv) > 2 else 96
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 4
scene = emu.create_scene(pkg.scene.SCENES[name]())
ls, mb = (6, 4) if name.startswith("hdri") else (2, 12)
film, prof = scene.render(pkg.api.render_desc(width, width * 9 // 16 if name == "cornell_gem" else width, spp, mb, light_samples=ls))
raw = ctypes.CDLL(lib)
raw.ptemu_walks_dump.restype = ctypes.c_size_t
buf = np.zeros(200_000_000, np.float32)
n = raw.ptemu_walks_dump(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), ctypes.c_size_t(buf.size))
w = buf[:n].reshape(-1, 12).astype(np.float64)
print("%d walks recorded; shadow rays %d, segments %d" % (len(w), prof.shadow_rays, prof.bounce_rays))
CB, CT = 45.0, 130.0
mesh = {"cornell_gem": "brilliant_diamond", "hdri_test": "monkey", "hdri_c4_small": "monkey", "test_prism": "prism", "test_prism_small": "prism"}[name]
z = np.load(os.path.join(R, "rust-pathtracer_amd", "data", "meshes", mesh + ".npz"))
P = z["positions"].astype(np.float64); F = z["faces"]
a, b, c = P[F[:, 0]], P[F[:, 1]], P[F[:, 2]]
nrm = np.cross(b - a, c - a); nrm /= np.linalg.norm(nrm, axis=1)[:, None]
dd = (nrm * a).sum(1)
convex = (P @ nrm.T - dd[None, :]).max() < 1e-3 * np.ptp(P, axis=0).max()
print("mesh %s: %d triangles, convex: %s" % (mesh, len(F), convex))
for kind, label in ((0, "closest-hit walks (stop NONE)"), (1, "light rays (stop NONLIGHT)"), (2, "environment rays (stop ANY)")):
    k = w[w[:, 8] == kind]
    if not len(k):
        continue
    cost = k[:, 9] * CB + k[:, 10] * CT
    print("%s: %d walks, %.1f boxes + %.2f triangles each, blocked/over %.3f" % (label, len(k), k[:, 9].mean(), k[:, 10].mean(), k[:, 11].mean()))
    if not convex:
        continue
    o, d = k[:, 0:3], k[:, 3:6]
    s = o @ nrm.T - dd[None, :]                  # signed distances of the origin to every face plane
    depth = s.max(axis=1)                        # < 0: inside the convex body, by that much
    inside = depth < 0
    # the convex body's exact answer for the segment [0, limit]: t_in / t_out over all planes
    limit = np.minimum(k[:, 6], k[:, 7])
    vd = d @ nrm.T
    with np.errstate(all="ignore"):
        t = -s / vd
    t_in = np.where(vd < 0, t, -np.inf).max(axis=1)
    t_out = np.where(vd > 0, t, np.inf).min(axis=1)
    crosses = (t_in < t_out) & (t_out > 0) & (t_in < limit)
    for lab, m in (("origin inside the body", inside), ("origin within 3e-3 outside", ~inside & (depth < 3e-3)), ("origin farther outside", depth >= 3e-3)):
        if m.any():
            print("    %-28s %6.1f %% of the walks, %5.1f %% of their cost; over %.3f; geometric crossing %.3f; %.1f boxes %.2f tris" % (
                lab, 100 * m.mean(), 100 * cost[m].sum() / cost.sum(), k[m, 11].mean(), crosses[m].mean(), k[m, 9].mean(), k[m, 10].mean()))
    far = depth >= 3e-3
    for lab, m in (("far outside, blocked", far & (k[:, 11] == 1)), ("far outside, not blocked", far & (k[:, 11] == 0))):
        if m.any():
            print("    %-28s %6.1f %% of the walks, %5.1f %% of their cost; %.1f boxes %.2f tris" % (lab, 100 * m.mean(), 100 * cost[m].sum() / cost.sum(), k[m, 9].mean(), k[m, 10].mean()))
    np.save("/tmp/light_walks_%s_%d.npy" % (name, kind), k)
