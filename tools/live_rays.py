#!/usr/bin/env python3
"""Host-side experiment: what share of k_shadow's light-sample rays is traced at all?  Renders a small frame on the emulated lane logic
(tools/live_rays.cpp) and prints the counts.  usage: tools/live_rays.py [scene] [light_samples] [max_bounces]"""
import ctypes
import importlib
import os
import subprocess
import sys

import numpy as np

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
pkg = importlib.import_module("rust-pathtracer_amd")
lib = os.path.join(R, "tests", "host_emulation", "libptliverays.so")
srcs = [os.path.join(R, "tools", "live_rays.cpp"), os.path.join(R, "rust-pathtracer_amd", "csrc", "pt_scene_host.cpp"), os.path.join(R, "rust-pathtracer_amd", "csrc", "pt_plan.cpp")]
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function", "-Wno-unused-value", "-o", lib] + srcs)
emu = pkg.api.Library(lib, "ptemu_", optional=("render_device", "device_info", "render_multi", "device_count"))
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mb = int(sys.argv[3]) if len(sys.argv) > 3 else 8
scene = emu.create_scene(pkg.scene.SCENES[name]())
film, prof = scene.render(pkg.api.render_desc(96, 96, 8, mb, light_samples=L))
raw = ctypes.CDLL(lib)
out = (ctypes.c_ulonglong * 4)()
raw.ptemu_live_stats(out)
live, dead, nolight, calls = out[0], out[1], out[2], out[3]
print("%s L=%d depth %d: light-sample rays %d = traced %d (%.3f) + dead %d (%.3f); of the traced, %d (%.3f) meet no light; profile shadow_rays %d; closest-hit calls %d" % (
    name, L, mb, live + dead, live, live / (live + dead), dead, dead / (live + dead), nolight, nolight / max(1, live), prof.shadow_rays, calls))
