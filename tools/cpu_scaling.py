#!/usr/bin/env python3
"""How the CPU oracle scales with host threads on this box (bench.py's cpu_baseline leg picks its thread count from this picture):
os.cpu_count(), the affinity mask, the cgroup CPU quota, and the oracle's Msamples/s on C2's scene at 1, 2, 4 ... threads.
usage: tools/cpu_scaling.py [max_threads]"""
import importlib
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_loader  # noqa: E402

pkg = importlib.import_module("rust-pathtracer_amd")
oracle = oracle_loader.load(pkg)
print("os.cpu_count()", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(p, open(p).read().strip())
    except OSError:
        pass
try:
    print(os.popen("lscpu | egrep 'Model name|Socket|Core|Thread|NUMA node\\(s\\)'").read())
except OSError:
    pass
scene = oracle.create_scene(pkg.scene.cornell_box())
top = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
threads, base = 1, None
while True:
    spp = max(1, min(64, threads // 2))
    rd = pkg.api.render_desc(1024, 1024, spp, 8, shard=(0, 8) if threads < 16 else (0, 0))
    t = time.perf_counter(); _, prof = oracle_loader.render_mt(oracle, scene, rd, threads); dt = time.perf_counter() - t
    rate = prof.camera_rays / dt / 1e6
    base = base or rate
    print("threads %4d  %8.3f Msamples/s  per thread %.4f  efficiency %.2f  (%d samples, %.1f s)" % (threads, rate, rate / threads, rate / threads / base, prof.camera_rays, dt), flush=True)
    if threads >= top:
        break
    threads = min(top, threads * 2)
