#!/usr/bin/env python3
"""Host-side experiment: how many leaves does a ray test in phase 3 of the leaf sweep, and what would a wave of 64 such rays execute
under different loop policies?  Renders a small frame on the emulated lane logic (tools/candidate_stats.cpp), groups the traced rays 64
at a time at random (the incoherent limit; real waves are more coherent) and prints: the histogram of leaves per ray, the loop steps of
the lane-by-lane loop (= the wave's maximum), of a loop whose rays park after `T` leaves and resume in full waves, of a perfect sort,
and of a loop that runs one kind of test per step.  usage: tools/candidate_stats.py [scene] (default cornell_box)"""
import ctypes
import importlib
import os
import subprocess
import sys

import numpy as np

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
pkg = importlib.import_module("rust-pathtracer_amd")
lib = os.path.join(R, "tests", "host_emulation", "libptcandstats.so")
srcs = [os.path.join(R, "tools", "candidate_stats.cpp"), os.path.join(R, "rust-pathtracer_amd", "csrc", "pt_scene_host.cpp"), os.path.join(R, "rust-pathtracer_amd", "csrc", "pt_plan.cpp")]
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function", "-o", lib] + srcs)
emu = pkg.api.Library(lib, "ptemu_", optional=("render_device", "device_info", "render_multi", "device_count"))
scene = emu.create_scene(pkg.scene.SCENES[sys.argv[1] if len(sys.argv) > 1 else "cornell_box"]())
scene.render(pkg.api.render_desc(64, 64, 4, 8, light_samples=2))
raw = ctypes.CDLL(lib)
raw.ptemu_seq_dump.restype = ctypes.c_size_t
buf = ctypes.create_string_buffer(50_000_000)
filled = raw.ptemu_seq_dump(buf, ctypes.c_size_t(len(buf)))
seqs = buf.raw[:filled].decode().split("\n")[:-1]
c = np.array([len(s) for s in seqs])
n = len(c)
print("rays %d, leaves per ray %.2f (triangles %.2f)" % (n, c.mean(), np.mean([s.count("T") for s in seqs])))
print("histogram", np.round(np.bincount(c)[:14] / n, 3))
rng = np.random.default_rng(1)
order = rng.permutation(n)[: n // 64 * 64].reshape(-1, 64)
print("lane-by-lane loop: %.2f steps per wave (lane utilisation %.2f); perfect sort: %.2f" % (c[order].max(axis=1).mean(), c.mean() / c[order].max(axis=1).mean(),
                                                                                              np.sort(c)[: n // 64 * 64].reshape(-1, 64).max(axis=1).mean()))
for T in (1, 2, 3, 4):   # every pass of a wave runs at most T steps; rays with leaves left are parked and resumed 64 at a time
    left, steps = c.copy(), 0.0
    while (left > 0).any():
        live = left[left > 0]
        steps += np.minimum(live, T).max() * len(live) / 64.0 / (n / 64.0) if len(live) < 64 else T * len(live) / n
        left = np.maximum(left - T, 0)
    print("leaf budget %d: %.2f steps per 64 rays" % (T, steps))
mixed = greedy = 0.0
for w in order:
    lanes = [seqs[i] for i in w]
    for k in range(max(len(s) for s in lanes)):
        kinds = {s[k] for s in lanes if k < len(s)}
        mixed += len(kinds)
    pos = [0] * 64
    while True:
        nxt = [s[p] if p < len(s) else None for s, p in zip(lanes, pos)]
        if all(x is None for x in nxt):
            break
        kind = "T" if "T" in nxt else "A"
        greedy += 1.0
        pos = [p + 1 if x == kind else p for p, x in zip(pos, nxt)]
print("tests per wave (a step with both kinds runs both): %.2f; one kind per step, triangles first: %.2f" % (mixed / len(order), greedy / len(order)))
