#!/usr/bin/env python3
"""Host-side experiment: what does a wave of 64 mesh walks execute under different loop policies?  Renders a small frame on the emulated
lane logic with a trace of every mesh walk (tools/walk_stats.cpp), groups the walks of a kernel (closest-hit walks = k_extend_parked, light
and environment rays = k_shadow_parked) 64 at a time — in trace order, which keeps neighbouring pixels together, and at random — and prices,
in wave instructions (a box step CB, a triangle pass CT, a resumed wave's prologue CR):
  while-while           the product before eviction: the inner loop runs until every lane has a leaf or is done
  evict below E         the same, a wave's last walks leave when fewer than E lanes are still walking and are pooled 64 at a time
  inner exit below X    ... and the inner loop ends once fewer than X lanes are still searching (the others test their triangles first)
  speculative           ... and a lane that holds a leaf goes on with box tests until it finds its next leaf
usage: tools/walk_stats.py [scene] [width] (default hdri_test 96)"""
import ctypes
import importlib
import os
import subprocess
import sys

import numpy as np

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
pkg = importlib.import_module("rust-pathtracer_amd")
lib = os.path.join(R, "tests", "host_emulation", "libptwalkstats.so")
srcs = [os.path.join(R, "tools", "walk_stats.cpp"), os.path.join(R, "rust-pathtracer_amd", "csrc", "pt_scene_host.cpp"), os.path.join(R, "rust-pathtracer_amd", "csrc", "pt_plan.cpp")]
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function", "-o", lib] + srcs)
emu = pkg.api.Library(lib, "ptemu_", optional=("render_device", "device_info", "render_multi", "device_count"))
name = sys.argv[1] if len(sys.argv) > 1 else "hdri_test"
width = int(sys.argv[2]) if len(sys.argv) > 2 else 96
scene = emu.create_scene(pkg.scene.SCENES[name]())
ls, mb = (6, 4) if name.startswith("hdri") else (2, 12)
scene.render(pkg.api.render_desc(width, width, 2, mb, light_samples=ls))
raw = ctypes.CDLL(lib)
raw.ptemu_seq_dump.restype = ctypes.c_size_t
buf = ctypes.create_string_buffer(400_000_000)
filled = raw.ptemu_seq_dump(buf, ctypes.c_size_t(len(buf)))
seqs = buf.raw[:filled].decode().split("\n")[:-1]
CB, CT, CR = 45.0, 130.0, 350.0


def segments(s):
    """a walk as its inner loops: [(box tests, ends with a triangle test)]"""
    out, n = [], 0
    for c in s[1:]:
        if c == "b":
            n += 1
        else:
            out.append((n, True)); n = 0
    if n or not out:
        out.append((n, False))
    return out


def run_wave(lanes, evict=0, inner_exit=0, speculative=False, can_evict=True):
    """lanes: list of segment lists.  Returns (wave instructions, evicted remainders)."""
    lanes = [list(l) for l in lanes if l]
    cost = 0.0
    # per lane: remaining boxes of the current inner loop, whether a leaf is held, boxes done ahead (speculation)
    cur = [[l[0][0], False] for l in lanes]
    idx = [0] * len(lanes)
    alive = list(range(len(lanes)))
    while alive:
        # inner loop: lanes that are searching (remaining boxes > 0 or segment ends without a leaf -> they finish)
        while True:
            searching = [k for k in alive if not cur[k][1]]
            if not searching or (inner_exit and len(searching) < inner_exit and len(searching) < len(alive)):
                break
            cost += CB
            for k in list(alive):
                if cur[k][1] and not speculative:
                    continue
                if cur[k][1] and speculative:
                    # a lane that holds a leaf works ahead on the next inner loop, up to (not including) taking its leaf
                    nxt = idx[k] + 1
                    if nxt < len(lanes[k]) and lanes[k][nxt][0] > 0:
                        lanes[k][nxt] = (lanes[k][nxt][0] - 1, lanes[k][nxt][1])
                    continue
                if cur[k][0] > 0:
                    cur[k][0] -= 1
                if cur[k][0] == 0:
                    if lanes[k][idx[k]][1]:
                        cur[k][1] = True
                    else:
                        alive.remove(k)   # the walk is over
        holders = [k for k in alive if cur[k][1]]
        if holders:
            cost += CT
            for k in holders:
                idx[k] += 1
                if idx[k] >= len(lanes[k]):
                    alive.remove(k)
                else:
                    cur[k] = [lanes[k][idx[k]][0], False]
                    if cur[k][0] == 0:
                        if lanes[k][idx[k]][1]:
                            cur[k][1] = True
                        else:
                            alive.remove(k)
        if evict and can_evict and alive and len(alive) < evict:
            rest = []
            for k in alive:
                r = [(cur[k][0], lanes[k][idx[k]][1])] + lanes[k][idx[k] + 1:]
                rest.append(r)
            return cost, rest
    return cost, []


def price(walks, order, **policy):
    pool = [walks[i] for i in order]
    total, waves = 0.0, 0
    fresh = True
    while pool:
        take, pool = pool[:64], pool[64:]
        c, rest = run_wave(take, can_evict=len(pool) > 0, **policy)
        total += c + (0 if fresh and not policy.get("evict") else 0)
        waves += 1
        if rest:
            total += CR * len(rest) / 64.0   # a resumed ray's share of a prologue
            pool.extend(rest)
    return total


rng = np.random.default_rng(1)
for kernel, kinds in (("k_extend_parked (closest hit)", "C"), ("k_shadow_parked (light + environment rays)", "LE")):
    walks = [segments(s) for s in seqs if s[0] in kinds]
    if len(walks) < 128:
        print("%s: only %d walks traced (a bigger frame: the second argument)" % (kernel, len(walks)))
        continue
    walks = walks[: min(len(walks), 64 * 400)]
    n = len(walks) // 64 * 64
    walks = walks[:n]
    boxes = np.array([sum(b for b, _ in w) for w in walks]); tris = np.array([sum(1 for _, t in w if t) for w in walks])
    useful = float((boxes * CB + tris * CT).sum())
    print("%s: %d walks, %.1f box tests and %.2f triangle tests per walk (max %d / %d); box tests per inner loop %.2f" % (
        kernel, n, boxes.mean(), tris.mean(), boxes.max(), tris.max(), boxes.sum() / max(1, sum(len(w) for w in walks))))
    for label, order in (("trace order", list(range(n))), ("random", list(rng.permutation(n)))):
        base = price(walks, order)
        print("  %s: while-while %.0f instr per wave, lane utilisation %.2f" % (label, base / (n / 64), useful / 64 / base))
        for pol in ({"evict": 16}, {"evict": 32}, {"evict": 48}, {"evict": 32, "inner_exit": 8}, {"evict": 32, "inner_exit": 16}, {"evict": 32, "inner_exit": 32},
                    {"evict": 32, "speculative": True}, {"evict": 32, "inner_exit": 16, "speculative": True}):
            c = price(walks, order, **pol)
            print("     %-55s %.3f of while-while, utilisation %.2f" % (pol, c / base, useful / 64 / c))
