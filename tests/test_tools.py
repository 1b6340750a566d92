"""The measurement tools that share the engine's lane code through trace hooks (PT_STAT_EVENT) keep building and tracing: a change of
pt_device.h that drops a hook would silently blind them."""
import os
import re
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def test_walk_stats_traces_the_mesh_walks():
    """tools/walk_stats.py on the small C4 scene: both kernels' walks are traced (closest-hit walks of the monkey, light-sample walks), with
    box and triangle tests per walk, and the loop policies are priced."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "walk_stats.py"), "hdri_c4_small", "56"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout
    for kernel in ("k_extend_parked", "k_shadow_parked"):
        m = re.search(kernel + r"[^\n]*: (\d+) walks, ([0-9.]+) box tests and ([0-9.]+) triangle tests per walk", out)
        assert m, out[-2000:]
        assert int(m.group(1)) >= 128 and float(m.group(2)) > 3.0
    assert "while-while" in out and "evict" in out
