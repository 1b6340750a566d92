"""bench.py's entry point (round-3 verdict, item 1): `python bench.py --gpus N` must really run N ranks — as fresh child processes
started by a parent that makes no GPU call — relay rank 0's single record, and refuse to print a record that is not about N GPUs.
CPU tier: the ranks are a stub here (PT_BENCH_LAUNCHER), and once the real `torch.distributed.run`, whose ranks stop at "needs a GPU"."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BENCH = os.path.join(ROOT, "bench.py")

STUB = r'''
import json, os, sys
# what a launcher's rank 0 would leave on stdout: RCCL's banner, a stray brace line, then the record
script, args = sys.argv[1], sys.argv[2:]
assert os.path.basename(script) == "bench.py" and "--gpus" in args, sys.argv
assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0" and os.environ.get("PT_BENCH_PARENT")
n = int(args[args.index("--gpus") + 1])
mode = os.environ.get("STUB_MODE", "ok")
print("RCCL version 2.22.3+hip6.3")
print("{not a record}")
sys.stderr.write("rank noise\n")
if mode == "fail":
    sys.exit(7)
if mode == "silent":
    sys.exit(0)
if mode == "port":   # the launcher lost the race for its master port the first time
    marker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "port_lost_once")
    if not os.path.exists(marker):
        open(marker, "w").close()
        sys.stderr.write("RuntimeError: The server socket has failed to listen on any local network address. port: 29500, useIpv6: false, code: -98, name: EADDRINUSE\n")
        sys.exit(1)
rec = {"metric": "Msamples/s", "value": 123.0, "n_gpus": n if mode != "lies" else 1, "ranks_seen": n if mode != "half" else n - 1, "steps": 2}
print(json.dumps({"metric": "a line of JSON that is not the record", "n_gpus": 1}))
print("PT_BENCH_RECORD " + json.dumps(rec))   # (rank 0 marks its record for a bench.py parent)
print("trailing noise")
'''


def run_parent(tmp_path, mode, gpus=4, extra_env=None):
    stub = tmp_path / "stub_launcher.py"
    stub.write_text(STUB)
    env = dict(os.environ, PT_BENCH_LAUNCHER=json.dumps([sys.executable, str(stub)]), STUB_MODE=mode)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH, "--gpus", str(gpus), "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=120)


def test_parent_relays_one_record(tmp_path):
    r = run_parent(tmp_path, "ok")
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                        # ONE line of stdout: the record
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["ranks_seen"] == 4 and d["value"] == 123.0 and "launched_by" in d
    for noise in ("RCCL version", "{not a record}", "trailing noise", "rank noise", "a line of JSON that is not the record"):
        assert noise in r.stderr and noise not in r.stdout   # everything else the ranks said: stderr


def test_parent_starts_again_when_the_master_port_was_taken(tmp_path):
    """The parent finds a free port by bind-and-close; a launcher that then fails with EADDRINUSE is started again on another port."""
    r = run_parent(tmp_path, "port")
    assert r.returncode == 0, r.stderr
    assert "starting the ranks again" in r.stderr and json.loads(r.stdout.strip())["ranks_seen"] == 4


@pytest.mark.parametrize("mode,code", [("lies", 4), ("half", 4), ("fail", 7), ("silent", 3)])
def test_parent_refuses_a_record_that_is_not_about_n_gpus(tmp_path, mode, code):
    """A record with n_gpus != N (what round 3's bench.py would have produced: one rank, "n_gpus": 1), a rank missing from the
    all-reduced count, a launcher that failed, ranks that printed nothing: no record on stdout, a non-zero exit code."""
    r = run_parent(tmp_path, mode)
    assert r.returncode == code, (r.returncode, r.stderr)
    assert r.stdout.strip() == ""


def test_launcher_that_disagrees_with_gpus_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and r.stdout.strip() == "" and "WORLD_SIZE = 2" in r.stderr


def test_real_launcher_starts_n_ranks(tmp_path):
    """The real thing as far as a box without a GPU goes: the parent starts `torch.distributed.run` with two ranks, each rank passes the
    --gpus == WORLD_SIZE check and stops at the product's "needs a GPU" (no CPU fallback); the parent hands the failure on."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PT_BENCH_LAUNCHER"):
        env.pop(k, None)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_multi_process.py runs the launcher for real")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-seconds", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert r.stderr.count("bench.py needs a GPU") >= 2, r.stderr[-3000:]
