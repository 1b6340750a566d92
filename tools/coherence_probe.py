#!/usr/bin/env python3
"""Experiment (GPU box, under rocprofv3 --kernel-trace): what would sorted rays buy the closest-hit kernel?  Second-bounce rays of the
Cornell box (surface points reached by random rays, cosine-distributed directions) are traced through pt_intersect in different orders;
the kernel trace holds one k_probe_intersect launch per order, in the order printed here.  usage: tools/coherence_probe.py [n_rays]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pkg = importlib.import_module("rust-pathtracer_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 23
rng = np.random.default_rng(3)
sc = pkg.load().create_scene(pkg.scene.SCENES["cornell_box"]())
# surface points: rays from a point cloud in the middle of the room
o0 = rng.uniform(0.12, 0.43, (n, 3)).astype(np.float32)   # (the room spans about [0, 0.55]^3)
d0 = rng.normal(size=(n, 3)); d0 = (d0 / np.linalg.norm(d0, axis=1, keepdims=True)).astype(np.float32)
h0 = sc.intersect(o0, d0)
ok = h0["valid"] != 0
p, nn, inst = h0["point"][ok], h0["normal"][ok], h0["instance"][ok]
d0 = d0[ok]
nn = np.where((np.sum(nn * d0, axis=1) > 0)[:, None], -nn, nn)       # the side the ray came from
m = p.shape[0]
# cosine-distributed directions around nn
u1, u2 = rng.random(m), rng.random(m)
r, phi = np.sqrt(u1), 2 * np.pi * u2
lx, ly, lz = r * np.cos(phi), r * np.sin(phi), np.sqrt(1 - u1)
a = np.where(np.abs(nn[:, 0:1]) > 0.9, np.array([[0, 1, 0]]), np.array([[1, 0, 0]]))
t1 = np.cross(nn, a); t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
t2 = np.cross(nn, t1)
d1 = (t1 * lx[:, None] + t2 * ly[:, None] + nn * lz[:, None]).astype(np.float32)
o1 = (p + nn * 0.001).astype(np.float32)
octant = (d1[:, 0] > 0).astype(np.int64) | (d1[:, 1] > 0).astype(np.int64) << 1 | (d1[:, 2] > 0).astype(np.int64) << 2
fine = (np.clip(((d1 + 1) * 2).astype(np.int64), 0, 3) * np.array([1, 4, 16])).sum(axis=1)   # 4 x 4 x 4 direction cells
res = sc.intersect(o1, d1)                                                                    # (launch 2: as generated = random order)
target = np.where(res["valid"] != 0, res["instance"], 255).astype(np.int64)
orders = [("by origin surface", np.argsort(inst, kind="stable")),
          ("by origin surface, direction octant", np.lexsort((octant, inst))),
          ("by origin surface, 64 direction cells", np.lexsort((fine, inst))),
          ("by origin surface and the surface hit (needs the answer)", np.lexsort((target, inst))),
          ("by 64 direction cells only", np.argsort(fine, kind="stable"))]
print("launches of k_probe_intersect, in order: 1 = %d primary rays (random), 2 = %d second-bounce rays in random order" % (n, m))
for k, (name, idx) in enumerate(orders):
    sc.intersect(o1[idx], d1[idx])
    print("%d = %s" % (k + 3, name))
# light-sample rays from the vertices the second-bounce rays reached, towards random points of the ceiling light
ok2 = res["valid"] != 0
p2, n2, inst2 = res["point"][ok2], res["normal"][ok2], res["instance"][ok2]
n2 = np.where((np.sum(n2 * d1[ok2], axis=1) > 0)[:, None], -n2, n2)
k2 = p2.shape[0]
lp = np.stack([0.278 + (rng.random(k2) - 0.5) * 0.105, 0.2795 + (rng.random(k2) - 0.5) * 0.13, np.full(k2, 0.5487)], axis=1)
o2 = (p2 + n2 * 0.001).astype(np.float32)
d2 = lp - o2; d2 = (d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32)
fine2 = (np.clip(((d2 + 1) * 2).astype(np.int64), 0, 3) * np.array([1, 4, 16])).sum(axis=1)
fine3 = (np.clip(((d2 + 1) * 4).astype(np.int64), 0, 7) * np.array([1, 8, 64])).sum(axis=1)
base = len(orders) + 3
for k, (name, idx) in enumerate([("light-sample rays, %d, random order" % k2, np.arange(k2)), ("... by origin surface", np.argsort(inst2, kind="stable")),
                                 ("... by 64 direction cells", np.argsort(fine2, kind="stable")), ("... by 512 direction cells", np.argsort(fine3, kind="stable")),
                                 ("... by origin surface, 64 direction cells", np.lexsort((fine2, inst2)))]):
    sc.intersect(o2[idx], d2[idx])
    print("%d = %s" % (base + k, name))
