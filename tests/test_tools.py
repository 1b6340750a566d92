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


def test_light_walks_records_the_certificates_effect():
    """tools/light_walks.py (round 6: the measurement behind the convex-body certificates) on C3's scene: with the certificates the light-sample rays that still walk the gem
    no longer START on it going outward, and few start inside it; with them switched off (PTEMU_NO_CONVEX) most of the walks do."""
    def run(env):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "light_walks.py"), "cornell_gem", "96", "2"], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        m = re.search(r"light rays \(stop NONLIGHT\): (\d+) walks", r.stdout)
        assert m, r.stdout[-2000:]
        out = re.search(r"origin within 3e-3 outside\s+([0-9.]+) % of the walks", r.stdout[r.stdout.index("light rays"):])
        return int(m.group(1)), (float(out.group(1)) if out else 0.0)
    with_cert, outward_with = run({})
    without, outward_without = run({"PTEMU_NO_CONVEX": "1"})
    assert with_cert < 0.25 * without and outward_without > 40.0 and outward_with < 1.0, (with_cert, without, outward_with, outward_without)


def test_profile_summary_writes_the_l2_counts_bench_reads(tmp_path):
    """tools/summarize_profile.py turns a TCC_HIT_sum / TCC_MISS_sum pass into <tag>_tcc.txt, one line per kernel, and bench.py's roofline.l2 parses that line format."""
    import ast
    out = tmp_path / "prof"
    (out / "pmc_tcc" / "x").mkdir(parents=True)
    rows = ["Kernel_Name,Counter_Name,Counter_Value"]
    for launch, (hit, miss) in enumerate(((100.0, 50.0), (300.0, 150.0))):
        rows += ['"void ptk::k_shade<1, 1, 2, 0u, -1>(unsigned int const*)",TCC_HIT_sum,%r' % hit, '"void ptk::k_shade<1, 1, 2, 0u, -1>(unsigned int const*)",TCC_MISS_sum,%r' % miss]
    (out / "pmc_tcc" / "x" / "1_counter_collection.csv").write_text("\n".join(rows) + "\n")
    tag = "zz_test_%d" % os.getpid()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_profile.py"), str(out), tag], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    made = [os.path.join(ROOT, "profiles", "%s_%s" % (tag, n)) for n in ("tcc.txt", "summary.json", "kernel_stats.csv")]
    try:
        line = open(made[0]).read().strip()
        assert line.split(" ", 1)[0] == "k_shade" and line.endswith("launches 2")
        c = ast.literal_eval(line[line.index("{"):line.rindex("}") + 1])          # (what bench.py does with the line)
        assert float(c["TCC_HIT_sum"]) == 200.0 and float(c["TCC_MISS_sum"]) == 100.0
    finally:
        for f in made:
            if os.path.exists(f):
                os.remove(f)


def test_round_table_is_the_committed_records():
    """tools/round_table.py: the table of profiles/r6_experiments.md section 0 is made from the committed bench records of the tag it names — the figures of the experiments
    file are the records' (the default run's and C3's throughput, to the digit)."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "round_table.py"), "r6z", "r5z"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [l for l in r.stdout.split("\n") if l.startswith("| C") or l.startswith("| G")]
    assert len(rows) == 8, r.stdout
    text = open(os.path.join(ROOT, "profiles", "r6_experiments.md")).read()
    for cfg in ("default", "C3", "G1"):
        d = json.loads(open(os.path.join(ROOT, "profiles", "r6z_bench_%s.json" % cfg)).read().strip().split("\n")[-1])
        assert "**%.0f**" % d["value"] in r.stdout and "**%.0f**" % d["value"] in text, cfg


def test_the_soak_tools_take_a_switch(tmp_path):
    """tools/fuzz_emulation.py with PT_FUZZ_SWITCH: every scene is rendered once more under that switch of the emulated library and compared BIT FOR BIT (a wrong decision of a
    certificate that moves the film by less than the parity bar still changes bits); a few scenes of the one-glass-body class, whose segments take the inside rule."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_emulation.py"), "190000", "6"], capture_output=True, text=True, timeout=900, env=dict(os.environ, PT_FUZZ_SWITCH="PTEMU_NO_CONVEX"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    m = re.search(r"inside stops: (\d+) failures: 0", r.stdout)
    assert m and int(m.group(1)) > 50, r.stdout
