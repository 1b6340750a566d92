"""CPU, world_size 2, backend gloo: the N > 1 path of the film sharding.  Each process renders its shard
with the emulated engine (tests/host_emulation — the product itself has no CPU path), the films are summed with one
`dist.reduce`, and the result must equal the single-process film bit for bit."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

WORKER = r'''
import importlib, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, {here!r})
pkg = importlib.import_module("rust-pathtracer_amd")
rank, local_rank, world = pkg.sharding.rank_world()
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
emu = pkg.api.Library(os.path.join({here!r}, "host_emulation", "libptemu.so"), "ptemu_", optional=("render_device", "device_info"))
scene = emu.create_scene(pkg.scene.cornell_box())
S = pkg.sharding.weak_scaling_samples(3, world)
total = torch.zeros((40, 56, 4), dtype=torch.float32)
for k in range(2):
    rd = pkg.api.render_desc(56, 40, 2 * S, 4, tile=(16, 16), shard=pkg.sharding.shard(rank, world), first_sample=k * S, sample_count=S)
    film, prof = scene.render(rd)
    total += torch.from_numpy(film)
pkg.sharding.reduce_film(total, dst=0)
t = pkg.sharding.max_over_ranks(float(rank + 1), "cpu")
c = pkg.sharding.sum_over_ranks([prof.camera_rays], "cpu")
if rank == 0:
    np.save({out!r}, total.numpy())
    assert t == float(world) and c[0] == 56 * 40 * S, (t, c)
dist.destroy_process_group()
'''


def test_two_process_gloo_reduce_equals_single_process(pkg, tmp_path):
    import test_emulation  # builds libptemu.so on demand
    lib = os.path.join(HERE, "host_emulation", "libptemu.so")
    if not os.path.exists(lib):
        test_emulation.emu.__wrapped__(pkg) if hasattr(test_emulation.emu, "__wrapped__") else None
    assert os.path.exists(lib), "run tests/test_emulation.py first (it builds the emulation harness)"
    out = str(tmp_path / "film.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, here=HERE, out=out))
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    reduced = np.load(out)
    emu = pkg.api.Library(lib, "ptemu_", optional=("render_device", "device_info"))
    scene = emu.create_scene(pkg.scene.cornell_box())
    S = pkg.sharding.weak_scaling_samples(3, 2)
    whole = np.zeros_like(reduced)
    for k in range(2):
        film, _ = scene.render(pkg.api.render_desc(56, 40, 2 * S, 4, tile=(16, 16), first_sample=k * S, sample_count=S))
        whole += film
    assert np.array_equal(reduced, whole)
