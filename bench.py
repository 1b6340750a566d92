#!/usr/bin/env python3
"""bench.py — Msamples/s of the PT hot path on the BASELINE.json headline workload (C2: Cornell box 1024x1024,
max_bounces = 8, PT + NEE, L = 2), one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the wavefront pipeline over this rank's film shard.  `--scaling weak` (the default, what the driver's scaling
curve assumes): `--spp-per-step` x N samples per pixel — tiles are dealt along diagonals to ranks, every rank renders 1/N of the
pixels at N times the samples, so per-GPU work is fixed.  `--scaling strong`: the fixed frame of the BASELINE configurations (C2 = 1024
spp, C4 = 2048 spp of one 1024 x 1024 film) — `--spp-per-step` samples per pixel whatever N is, the tiles of that one frame dealt over
the N ranks, one reduce; total work is fixed and every fixed cost per call shows.  With one GPU the two modes are the same run.  Scene upload, BVH build and buffer allocation happen before the timed region (the window of
src/renderer/tiled.rs:294 -> 536); the film stays in HBM.  After the last step the rank films are summed into rank 0
with one RCCL reduce (disjoint shards, so the sum is a gather) inside the timed region.

Prints ONE JSON line on rank 0 with `roofline` (HIP-event time of the dominant kernel, measured inside the timed
region by the engine) and `cpu_baseline` (the oracle on the host cores, bounded sample).

`--gpus N` is what runs.  Under a launcher (WORLD_SIZE set) it must equal WORLD_SIZE or the run stops with exit code 2.  Without one and
N > 1, this process — which makes no GPU call and does not import torch — starts the N ranks as fresh child processes
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...  bench.py <the same arguments>`), relays rank
0's record as its own single line of stdout (everything else the ranks print goes to stderr) and exits with the launcher's code; a
record whose `n_gpus` / `ranks_seen` is not N is an error (exit code 4), as is a rank without a GPU of its own (`--allow-shared-gpus`
lets ranks share a device: the one-GPU tests).  For N > 1 the record also carries `strong`: the fixed BASELINE frames (C2: Cornell
1024 x 1024 x 1024 spp; C4: hdri_test 1024 x 1024 x 2048 spp, L = 6) dealt over the N ranks, one RCCL reduce each, with ms per frame,
Msamples/s, the reduce's share and the set-up time (`--strong-legs` adds them to a one-GPU run too).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# VALU issue rates measured on MI355X with tools/microbench/valu_issue.hip (profiles/r2a_valu_issue.txt), G wave64-instructions/s for the
# whole chip at 5-8 waves per SIMD: v_fma/add/mul/mov/xor issue in 2 cycles, every other VALU instruction (min/max/cmp/cndmask/integer/
# f64/packed/any with an SGPR operand) in 4.  The kernels here are ~80 % of the second kind.
VALU_PEAK_4CYCLE, VALU_PEAK_2CYCLE = 600.0, 990.0
SALU_PEAK = 540.0   # G scalar instructions/s for the whole chip, measured (profiles/r4_salu_issue.txt)
# the reference's only self-reported figure: 23.9 Mrays/s on 20 threads of an unnamed CPU (/root/reference/data/config.toml:4-8)
REFERENCE_SELF_REPORTED_MRAYS, REFERENCE_SELF_REPORTED_THREADS = 23.9, 20


def usable_cpus():
    """Host threads this process can really run on: the affinity mask capped by the cgroup CPU quota (the GPU boxes show 256 hardware
    threads and a quota of 16 CPUs; 256 oracle threads there run at half the rate of 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


STAGES = ["generate", "extend", "shade", "shadow", "accumulate"]


def algorithmic_bytes(stage, light_samples):
    """HBM bytes one work item of a stage must move (each field read once by its consumer / written once by its
    producer; DESIGN.md 'Algorithmic bytes').  4-byte fields."""
    path, hit = 16 * 4, 11 * 4
    if stage == "generate":
        return path + 4 + 4 + 4                   # write path state + zero energy + the wavelength sample k_accumulate reads, read pixel id (the lean camera record, 7 words for 16, is taken off below where a render used it)
    if stage == "extend":
        return 6 * 4 + hit                        # read ray (o, d), write hit record
    if stage == "shade":
        return path + hit + 4 + path + (3 + 7 * light_samples) * 4   # read state + hit + pixel id, write next state + light-sample item
    if stage == "shadow":
        return (3 + 7 * light_samples) * 4 + 8   # read item, read-modify-write the slot energy
    return 16 * 2                                 # accumulate: film pixel read-modify-write (energy reads are per sample, added below)


RECORD_MARK = "PT_BENCH_RECORD "   # in front of rank 0's record when a bench.py parent launched the ranks (PT_BENCH_PARENT): the parent relays exactly that line


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher, N > 1: start the N ranks as fresh children and relay rank 0's record.  This process
    never touches the GPU (no torch import, no HIP call): the children are the first to initialise it."""
    import socket
    import subprocess
    import threading
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env["PT_BENCH_PARENT"] = str(os.getpid())
    # The master port is found by bind-and-close, which another process can win before the launcher binds it again (round-4 advisor): a launcher that
    # fails on "address already in use" is started again on a new port, three times at most.  Rank 0 marks its record (RECORD_MARK: it sees PT_BENCH_PARENT),
    # so nothing else a rank prints — RCCL's banner, a stray line of JSON — can be taken for it.
    for attempt in range(3):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port)]
        if os.environ.get("PT_BENCH_LAUNCHER"):        # tests: a stand-in for torch.distributed.run (a JSON list; it gets the script and its arguments appended)
            launcher = json.loads(os.environ["PT_BENCH_LAUNCHER"])
        child = subprocess.Popen(launcher + [os.path.abspath(__file__)] + list(argv), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True)
        port_taken = []

        def relay_stderr(stream=child.stderr, seen=port_taken):
            for line in stream:
                if "EADDRINUSE" in line or "ddress already in use" in line:
                    seen.append(line)
                sys.stderr.write(line)
        relay = threading.Thread(target=relay_stderr, daemon=True)
        relay.start()
        record = None
        for line in child.stdout:                      # the ranks' stdout: keep rank 0's record, pass everything else on to stderr
            text = line.strip()
            if text.startswith(RECORD_MARK):
                try:
                    if record is not None:
                        sys.stderr.write(json.dumps(record) + "\n")
                    record = json.loads(text[len(RECORD_MARK):])
                    continue
                except ValueError:
                    pass
            sys.stderr.write(line)
        code = child.wait()
        relay.join(timeout=10)
        if code != 0 and record is None and port_taken and attempt < 2:
            sys.stderr.write("bench.py: port %d was taken before the launcher could bind it; starting the ranks again\n" % port)
            continue
        break
    sys.stderr.flush()
    if code != 0:
        sys.stderr.write("bench.py: the launcher of %d ranks exited with code %d\n" % (args.gpus, code))
        return code
    if record is None:
        sys.stderr.write("bench.py: %d ranks ran and none printed a record\n" % args.gpus)
        return 3
    if record.get("n_gpus") != args.gpus or record.get("ranks_seen") != args.gpus:
        sys.stderr.write("bench.py: asked for %d GPUs, the record says n_gpus = %r, ranks_seen = %r\n" % (args.gpus, record.get("n_gpus"), record.get("ranks_seen")))
        return 4
    record["launched_by"] = "bench.py --gpus %d: parent without GPU calls -> %s" % (args.gpus, " ".join(launcher[1:4] if launcher[0] == sys.executable else launcher[:1]))
    print(json.dumps(record), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp-per-step", type=int, default=1024, help="samples per pixel per step and per GPU (1024 = a step is the whole C2 frame: 1.07 G samples in 128 Mi-slot passes)")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--max-bounces", type=int, default=8)
    ap.add_argument("--light-samples", type=int, default=2)
    ap.add_argument("--scene", default="cornell_box")
    ap.add_argument("--min-bounces", type=int, default=1)
    ap.add_argument("--env-sampling-probability", type=float, default=None, help="override the scene's env_sampling_probability (as the reference's config file can)")
    ap.add_argument("--hero", type=int, default=1, help="wavelengths per path: 1, or 4 for the hero-wavelength variant (C5)")
    ap.add_argument("--medium-aware", action="store_true", help="the medium-aware walk (RenderSettings of the reference's PT integrator with mediums: src/integrator/utils.rs:708-1103; scene fog_ball)")
    ap.add_argument("--workload", default=None, help="label for config.workload (default: derived from the arguments)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the oracle baseline (0 = skip)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every GPU renders 1/N of the pixels at N x --spp-per-step (per-GPU work fixed); strong: the frame is fixed at --spp-per-step, its tiles dealt over the N GPUs")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (nccl = RCCL) even for one process, so that the film reduce runs through RCCL")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="torch.distributed backend: nccl = RCCL over xGMI (the product's); gloo only for tests that put two ranks on one GPU, which RCCL refuses")
    ap.add_argument("--allow-shared-gpus", action="store_true", help="let ranks share a physical GPU (tests on a one-GPU box); by default a rank without a device of its own is an error")
    ap.add_argument("--strong-legs", action="store_true", help="add the fixed-frame (strong scaling) legs of C2 and C4 to the record also with one GPU (they always run with N > 1)")
    ap.add_argument("--no-strong-legs", action="store_true", help="skip them")
    ap.add_argument("--strong-frames", type=int, default=3, help="timed frames per strong-scaling leg (after one warm-up frame)")
    ap.add_argument("--strong-spp-div", type=int, default=1, help="divide the legs' samples per pixel (C2 1024, C4 2048) by this: tests")
    args = ap.parse_args()

    # ---- how many GPUs: --gpus is what runs.  No launcher and N > 1: be the launcher.  A launcher that disagrees: stop.
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE = %s ranks; refusing to report either number\n" % (args.gpus, env_world))
        sys.exit(2)

    import numpy as np
    import torch
    import torch.distributed as dist

    pkg = importlib.import_module("rust-pathtracer_amd")
    sharding = pkg.sharding
    rank, local_rank, world = sharding.rank_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    n_devices = torch.cuda.device_count()
    if local_rank >= n_devices and n_devices != 1 and not args.allow_shared_gpus:
        # (n_devices == 1 with several ranks: a launcher that exposes one GPU per process shows each its own as device 0 — told apart below by the bus ids)
        raise SystemExit("bench.py: rank %d (local rank %d) has no GPU of its own: %d device(s) visible, %d ranks (--allow-shared-gpus to share)" % (rank, local_rank, n_devices, world))
    device_index = local_rank % max(1, n_devices)
    torch.cuda.set_device(device_index)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend="gloo")
    n_gpus = world
    # who is really there: ranks counted by an all-reduce, physical devices by their PCI bus ids (ranks that share one count once)
    ranks_seen = int(round(sharding.sum_over_ranks([1.0], "cuda")[0]))
    try:
        bus_id = int(torch.cuda.get_device_properties(device_index).pci_bus_id) + 256 * int(torch.cuda.get_device_properties(device_index).pci_domain_id)
    except (AttributeError, TypeError, ValueError):
        bus_id = device_index
    if use_dist:
        ids = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(world)]
        dist.all_gather(ids, torch.tensor([bus_id], dtype=torch.int64, device="cuda"))
        physical_gpus = len({int(t.item()) for t in ids})
    else:
        physical_gpus = 1
    if physical_gpus < world and not args.allow_shared_gpus:
        raise SystemExit("bench.py: %d ranks on %d physical GPU(s) (--allow-shared-gpus to share)" % (world, physical_gpus))
    try:
        rccl_version = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        rccl_version = None

    engine = pkg.load()
    t_create = time.perf_counter()
    builder = pkg.scene.SCENES[args.scene]()
    if args.env_sampling_probability is not None:   # (the config file's override of the scene's value: TOMLConfig::env_sampling_probability, src/parsing/config.rs:125-130)
        builder.env_sampling_probability = args.env_sampling_probability
    scene = engine.create_scene(builder)
    scene_create_ms = 1e3 * (time.perf_counter() - t_create)

    W, H, L = args.width, args.height, args.light_samples
    # weak scaling: N x the samples on 1/N of the pixels; strong scaling: the frame's own samples on 1/N of the pixels
    S = sharding.weak_scaling_samples(args.spp_per_step, n_gpus) if args.scaling == "weak" else args.spp_per_step
    total_spp = S * (args.steps + args.warmup)
    assert total_spp <= 65535 * 16
    film_step = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    film_total = torch.zeros_like(film_step)
    stream = torch.cuda.current_stream().cuda_stream

    def step(k):
        rd = pkg.api.render_desc(W, H, total_spp, args.max_bounces, min_bounces=args.min_bounces, light_samples=L, seed=1,
                                 shard=sharding.shard(rank, n_gpus), first_sample=k * S, sample_count=S, hero_wavelengths=args.hero, medium_aware=args.medium_aware)
        prof = scene.render_device(rd, film_step.data_ptr(), stream)
        film_total.add_(film_step)
        return prof

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    sync()
    t0 = time.perf_counter()
    profs = [step(args.warmup + k) for k in range(args.steps)]
    sharding.reduce_film(film_total, dst=0)                      # RCCL over xGMI: the only exchange step
    sync()
    elapsed = sharding.max_over_ranks(time.perf_counter() - t0, "cuda")

    # ---- the fixed BASELINE frames dealt over the N ranks (strong scaling): after the timed region of `value`, with their own clocks
    def strong_leg(name, leg_scene, create_ms, spp, max_bounces, min_bounces, light_samples, hero):
        """One whole frame per step, whatever N is: this rank's tiles of it, then the RCCL reduce.  Per frame: [barrier] render the shard
        [device sync: this rank's render time] [barrier] reduce [device sync, barrier] — so the reduce is timed free of the wait for the
        slowest rank, and the ranks' own render times show the load balance of the tile deal."""
        film = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")

        def frame(k):
            rd = pkg.api.render_desc(W, H, spp, max_bounces, min_bounces=min_bounces, light_samples=light_samples, seed=1 + k,
                                     shard=sharding.shard(rank, n_gpus), hero_wavelengths=hero)
            t = time.perf_counter()
            prof = leg_scene.render_device(rd, film.data_ptr(), stream)
            torch.cuda.synchronize()
            t_render = time.perf_counter() - t
            if use_dist:
                dist.barrier()
            t = time.perf_counter()
            sharding.reduce_film(film, dst=0)
            sync()
            return prof, t_render, time.perf_counter() - t

        sync()
        t = time.perf_counter()
        frame(0)                                   # warm-up: queue buffers of this film / light-sample count, first-use work
        first_ms = 1e3 * sharding.max_over_ranks(time.perf_counter() - t, "cuda")
        sync()
        t = time.perf_counter()
        res = [frame(1 + k) for k in range(args.strong_frames)]
        total = sharding.max_over_ranks(time.perf_counter() - t, "cuda")
        render_max = sharding.max_over_ranks(sum(r[1] for r in res), "cuda")
        render_min = -sharding.max_over_ranks(-sum(r[1] for r in res), "cuda")
        reduce_s = sharding.max_over_ranks(sum(r[2] for r in res), "cuda")
        cam = sharding.sum_over_ranks([sum(r[0].camera_rays for r in res)], "cuda")[0]
        f = float(args.strong_frames)
        # ---- SURVEY 8(e)'s claim, checked by the record itself (round-5 verdict, item 4): "the reduced film of N ranks is bit-identical to the 1-GPU film".  A small frame of
        # the leg's own settings (256 x 256, or the film if smaller): every rank renders its tiles of it, one reduce into rank 0 — the leg's own code path — and rank 0 ALSO
        # renders the whole frame alone; the two films are compared bit for bit on the device.  After the leg's clocks.
        cw, ch = min(W, 256), min(H, 256)
        small = torch.zeros((ch, cw, 4), dtype=torch.float32, device="cuda")
        rd_s = pkg.api.render_desc(cw, ch, spp, max_bounces, min_bounces=min_bounces, light_samples=light_samples, seed=99, shard=sharding.shard(rank, n_gpus), hero_wavelengths=hero)
        prof_s = leg_scene.render_device(rd_s, small.data_ptr(), stream)
        torch.cuda.synchronize()
        sharding.reduce_film(small, dst=0)
        sync()
        rays = sharding.sum_over_ranks([prof_s.camera_rays, prof_s.bounce_rays, prof_s.shadow_rays], "cuda")
        check = None
        if rank == 0:
            alone = torch.zeros_like(small)
            prof_a = leg_scene.render_device(pkg.api.render_desc(cw, ch, spp, max_bounces, min_bounces=min_bounces, light_samples=light_samples, seed=99, hero_wavelengths=hero), alone.data_ptr(), stream)
            torch.cuda.synchronize()
            same = bool(torch.equal(small.view(torch.int32), alone.view(torch.int32)))
            check = {"film_equals_one_rank": same and [int(x) for x in rays] == [prof_a.camera_rays, prof_a.bounce_rays, prof_a.shadow_rays],
                     "width": cw, "height": ch, "spp": spp, "ranks": n_gpus, "films_bitwise_equal": same,
                     "counters_n_ranks": [int(x) for x in rays], "counters_one_rank": [prof_a.camera_rays, prof_a.bounce_rays, prof_a.shadow_rays],
                     "max_abs_difference": float((small - alone).abs().max().item()), "film_max": float(alone.max().item())}
        sync()
        return {"frame": name, "film_equals_one_rank": check["film_equals_one_rank"] if check else None, "one_rank_check": check,
                "rccl_ranks": (dist.get_world_size() if use_dist else 1), "backend": (args.backend if use_dist else None), "width": W, "height": H, "spp": spp, "max_bounces": max_bounces, "light_samples": light_samples, "n_gpus": n_gpus, "frames": args.strong_frames,
                "ms_per_frame": 1e3 * total / f, "value": cam / total / 1e6, "unit": "Msamples/s", "samples_per_frame": cam / f,
                "render_ms_slowest_rank": 1e3 * render_max / f, "render_ms_fastest_rank": 1e3 * render_min / f, "reduce_ms": 1e3 * reduce_s / f,
                "film_bytes_reduced": W * H * 16, "setup_ms": create_ms + first_ms, "setup": {"scene_create_ms": create_ms, "first_frame_ms": first_ms}}

    strong = None
    if (world > 1 or args.strong_legs) and not args.no_strong_legs:
        div = max(1, args.strong_spp_div)
        c2_scene, c2_ms = (scene, scene_create_ms)
        if args.scene != "cornell_box":
            t = time.perf_counter(); c2_scene = engine.create_scene(pkg.scene.SCENES["cornell_box"]()); c2_ms = 1e3 * (time.perf_counter() - t)
        strong = {"C2": strong_leg("C2: Cornell box, max_bounces 8, L = 2", c2_scene, c2_ms, max(1, 1024 // div), 8, 1, 2, 1)}
        t = time.perf_counter(); c4_scene = engine.create_scene(pkg.scene.SCENES["hdri_test"]()); c4_ms = 1e3 * (time.perf_counter() - t)
        strong["C4"] = strong_leg("C4: hdri_test (sphere + monkey mesh under the synthetic HDRI), max_bounces 4, L = 6", c4_scene, c4_ms, max(1, 2048 // div), 4, 1, 6, 1)
        strong["note"] = ("total work fixed: one BASELINE frame per step, its 32x32 tiles dealt along diagonals over the N ranks, one RCCL reduce of the whole XYZ film per frame; "
                          "ms_per_frame = wall time barrier to barrier, max over ranks; setup_ms (scene build + upload + first frame's allocations) is outside it, "
                          "as parsing and BVH build are outside the reference's own window (tiled.rs:294 -> 536); film_equals_one_rank: a 256 x 256 frame of the leg's settings rendered by the N ranks "
                          "and reduced, and by rank 0 alone — films compared bit for bit, counters summed over the ranks (SURVEY 8(e): the shards are disjoint, the sum is a gather)")

    # whole-job units: every rank rendered (pixels / N) x (S = spp_per_step x N) samples per step
    shard_pixels = sum(p.stage_items[4] for p in profs) / max(1, len(profs))
    samples_rank = sum(p.camera_rays for p in profs)
    counts = sharding.sum_over_ranks([samples_rank] + [sum(p.stage_items[i] for p in profs) for i in range(5)] +
                                     [sum(p.bounce_rays for p in profs), sum(p.shadow_rays for p in profs)], "cuda")
    total_samples = counts[0]
    value = total_samples / elapsed / 1e6

    # BASELINE.json's own metric string when this run is its configuration (C2), a descriptive one otherwise
    metric_name = "Msamples/s (paths x spp / s), %s %dx%d, max_bounces=%d" % (args.scene, W, H, args.max_bounces)
    if (args.scene, W, H, args.max_bounces, L, args.hero) == ("cornell_box", 1024, 1024, 8, 2, 1):
        try:
            metric_name = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        except (OSError, ValueError, KeyError):
            pass
    if rank == 0:
        # ---- roofline of the dominant kernel (rank 0's own launches, HIP events inside the timed region)
        ksec = [sum(p.kernel_seconds[i] for p in profs) for i in range(5)]
        klaunch = [sum(p.kernel_launches[i] for p in profs) for i in range(5)]
        kitems = [sum(p.stage_items[i] for p in profs) for i in range(5)]
        dom = max(range(5), key=lambda i: ksec[i])
        per_item = [algorithmic_bytes(s, L) for s in STAGES]
        fused = klaunch[1] == 0 and klaunch[2] > 0      # k_shade traced its own segments: no hit queue, the ray read with the rest of the path record
        if fused:
            per_item[2] -= 11 * 4                       # no hit record to read (and k_extend's 24 + 44 B per segment are not moved at all)
        kbytes = [per_item[i] * kitems[i] for i in range(5)]
        kbytes[4] += 8 * sum(p.camera_rays for p in profs)          # accumulate also reads one energy and one wavelength sample per sample
        camera_record = all(p.stage_items[7] == 1 for p in profs)    # the camera vertex' lean record: 7 of the 16 words written by k_generate, read at bounce 0
        if camera_record:
            lean = 9 * 4 * sum(p.camera_rays for p in profs)
            kbytes[0] -= lean; kbytes[2] -= lean
            per_item[0] -= 9 * 4
        kernels = {}
        for i, s in enumerate(STAGES):
            if klaunch[i]:
                kernels[s] = {"launches": klaunch[i], "avg_us": 1e6 * ksec[i] / klaunch[i], "items_per_launch": kitems[i] / klaunch[i],
                              "bytes_per_item": per_item[i], "achieved_GBs": (kbytes[i] / ksec[i] / 1e9) if ksec[i] > 0 else None}
        achieved = kbytes[dom] / ksec[dom] / 1e9 if ksec[dom] > 0 else 0.0
        cam = sum(p.camera_rays for p in profs)
        segs = kitems[1]
        d_bar = segs / cam if cam else 0.0
        # HBM bytes per launch of that kernel, and its VALU issue rate, from the committed rocprofv3 PMC passes of this same command
        # (profiles/<tag>_summary.json, tools/profile_gpu.sh): (2 x FETCH_SIZE + WRITE_SIZE) KiB — FETCH_SIZE reads 1/2 of a
        # coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM), WRITE_SIZE is exact (calibrated on k_generate's 68 B/item).
        # Only a profile taken on exactly this workload counts (every key below must match); otherwise the fields stay null.
        workload_key = {**({"env_sampling_probability": args.env_sampling_probability} if args.env_sampling_probability is not None else {}), "scene": args.scene, "width": W, "height": H, "max_bounces": args.max_bounces, "min_bounces": args.min_bounces,
                        "light_samples": L, "hero": args.hero, "spp_per_step": S, "n_gpus": n_gpus, **({"medium_aware": True} if args.medium_aware else {})}
        traffic, traffic_src, valu = None, None, None
        try:
            import glob
            # (by tag, newest round and letter first: r3b before r3a before r2s — file times do not survive a checkout)
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json")), key=os.path.basename, reverse=True):
                summ = json.load(open(path))
                if summ.get("workload_key") != workload_key:
                    continue
                # the stage's kernel under whichever of its forms ran (k_shadow, k_shadow_parked, ...): the one with the most time
                names = [k for k in summ.get("kernels", {}) if k.startswith("k_" + STAGES[dom])]
                if not names:
                    continue
                kname = max(names, key=lambda k: summ["kernels"][k]["total_ms"])
                traffic = (2.0 * summ["FETCH_SIZE"][kname]["avg_per_launch"] + summ["WRITE_SIZE"][kname]["avg_per_launch"]) * 1024.0
                traffic_src = os.path.basename(path)
                sq = summ.get("SQ", {}).get(kname, {})
                if sq.get("SQ_INSTS_VALU") and summ["kernels"][kname].get("total_ms"):
                    # (the SQ counters were collected over the same launches the trace pass timed: same command, same step count)
                    rate = sq["SQ_INSTS_VALU"] / (summ["kernels"][kname]["total_ms"] * 1e-3) / 1e9
                    valu = {"kernel": kname, "achieved": rate, "unit": "G wave64-instructions/s", "peak_4cycle_class": VALU_PEAK_4CYCLE,
                            "peak_2cycle_class": VALU_PEAK_2CYCLE, "frac_of_4cycle_peak": rate / VALU_PEAK_4CYCLE, "frac_of_2cycle_peak": rate / VALU_PEAK_2CYCLE,
                            "lane_utilization": (sq["SQ_THREAD_CYCLES_VALU"] / (64.0 * sq["SQ_ACTIVE_INST_VALU"])) if sq.get("SQ_ACTIVE_INST_VALU") else None,
                            "instructions_per_wave": sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"] if sq.get("SQ_WAVES") else None,
                            "scalar_unit": ({"achieved": sq["SQ_INSTS_SALU"] / (summ["kernels"][kname]["total_ms"] * 1e-3) / 1e9, "unit": "G scalar instructions/s",
                                             "peak": SALU_PEAK, "frac": sq["SQ_INSTS_SALU"] / (summ["kernels"][kname]["total_ms"] * 1e-3) / 1e9 / SALU_PEAK,
                                             "per_vector_instruction": sq["SQ_INSTS_SALU"] / sq["SQ_INSTS_VALU"],
                                             "peak_source": "profiles/r4_salu_issue.txt (tools/microbench/salu_issue.hip on MI355X): one scalar unit per CU, shared by its four SIMDs"}
                                            if sq.get("SQ_INSTS_SALU") else None),
                            "peaks_source": "profiles/r2a_valu_issue.txt (tools/microbench/valu_issue.hip on MI355X): fma/add/mul/mov/xor issue in 2 cycles per "
                                            "wave64, min/max/cmp/cndmask/integer/f64/packed/SGPR-operand instructions in 4", "source": traffic_src}
                break
        except Exception:
            traffic, valu = None, None
        # SURVEY.md 8(d)'s a-priori byte model next to the engine's own record sizes: B(sample) = 32 + D (192 + 72 L)
        survey_bytes = 32.0 + d_bar * (192.0 + 72.0 * L)
        # What `traffic` is made of (round-5 verdict, item 5): FETCH_SIZE counts every L2 miss, whether HBM or the 256 MB Infinity Cache (MALL) serves it
        # (MI355X_MICROARCH.md).  A scene whose tables exceed the 4 MB L2 of an XCD but fit the MALL — C4: 20 MB of importance-map rows, guide tables and texels — shows
        # "traffic" well above its algorithmic bytes without reading HBM again: fabric traffic behind L2 misses.  The TCC hit / miss counts of the same workload, where a
        # PMC pass of them is committed (profiles/<tag>_tcc.txt), ride along.
        traffic_note, l2 = None, None
        if traffic is not None:
            table_bytes = 4 * int(engine.lib.pt_debug_scene_info(scene.handle, 8))
            traffic_note = ("(2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch; FETCH_SIZE counts L2 misses served by the Infinity Cache (MALL) as well as by HBM: "
                            "bytes above the algorithmic ones are re-reads of L2-missing scene tables (%.1f MB of texels and importance-map tables in this scene; "
                            "they fit the 256 MB MALL), i.e. fabric traffic, not HBM traffic" % (table_bytes / 1e6))
            try:
                tcc_path = os.path.join(ROOT, "profiles", traffic_src.replace("_summary.json", "_tcc.txt"))
                for line in open(tcc_path):
                    if line.split(" ", 1)[0] == kname:
                        import ast
                        body = line[line.index("{"):line.rindex("}") + 1]
                        c = ast.literal_eval(body)
                        hit, miss = float(c["TCC_HIT_sum"]), float(c["TCC_MISS_sum"])
                        l2 = {"kernel": kname, "hit": hit, "miss": miss, "hit_rate": hit / (hit + miss) if hit + miss > 0 else None, "source": os.path.basename(tcc_path)}
            except (OSError, ValueError, KeyError, SyntaxError):
                l2 = None
        roofline = {"bound": "hbm", "kernel": "k_" + STAGES[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "traffic_note": traffic_note, "l2": l2,
                    "avg_launch_us": 1e6 * ksec[dom] / klaunch[dom], "algorithmic_bytes_per_launch": kbytes[dom] / klaunch[dom],
                    "device_time_share": ksec[dom] / sum(ksec) if sum(ksec) > 0 else None,
                    "whole_pipeline": {"bytes_per_sample": sum(kbytes) / cam if cam else None, "segments_per_sample": d_bar,
                                       "achieved_GBs": sum(kbytes) / elapsed / 1e9, "frac": sum(kbytes) / elapsed / 1e9 / HBM_PEAK_GBS,
                                       "survey_model": {"bytes_per_sample": survey_bytes, "achieved_GBs": value * 1e6 * survey_bytes / 1e9,
                                                        "frac": value * 1e6 * survey_bytes / 1e9 / HBM_PEAK_GBS}},
                    "kernels": kernels, "valu": valu, "fused_extend_shade": fused,
                    "binding_resource": "VALU issue, not HBM: the scene (%d B) is LDS-resident, HBM only carries the queues; `frac` is the fraction of the HBM roof "
                                        "the contract asks for, `valu` (when a matching profile is committed) the fraction of the measured VALU issue rate" % engine.lib.pt_debug_scene_info(scene.handle, 0)}

        # ---- CPU baseline: the oracle on the host cores, bounded sample of the same workload
        cpu = None
        if args.cpu_seconds > 0:
            import oracle_loader
            oracle = oracle_loader.load(pkg)
            oscene = oracle.create_scene(builder)
            cores = usable_cpus()

            def orender(rd):
                t = time.perf_counter(); _, pr = oracle_loader.render_mt(oracle, oscene, rd, cores); return pr, time.perf_counter() - t
            # calibrate on ~1 s of work, then a bounded sample: every 32x32 tile of the film (the thread pool takes them from one
            # queue) at as many samples per pixel as the budget buys — at least 4 tiles per thread of ~1 s each, so the tail is short
            pp, dt = orender(pkg.api.render_desc(W, H, 1, args.max_bounces, min_bounces=args.min_bounces, light_samples=L, seed=1, shard=(0, 4), hero_wavelengths=args.hero, medium_aware=args.medium_aware))
            rate = pp.camera_rays / dt
            pp, dt = orender(pkg.api.render_desc(W, H, max(1, min(16, int(rate * 2.0 / (W * H)))), args.max_bounces, min_bounces=args.min_bounces, light_samples=L, seed=1, hero_wavelengths=args.hero, medium_aware=args.medium_aware))
            rate = pp.camera_rays / dt
            spp = max(1, int(rate * args.cpu_seconds / (W * H)))
            pc, dt = orender(pkg.api.render_desc(W, H, spp, args.max_bounces, min_bounces=args.min_bounces, light_samples=L, seed=1, hero_wavelengths=args.hero, medium_aware=args.medium_aware))
            rays_per_sample = (pc.bounce_rays + pc.shadow_rays) / max(1, pc.camera_rays)
            cpu = {"value": pc.camera_rays / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
                   "sample": "oracle (C++ restatement of the reference PT path, std::thread over a queue of 32x32 tiles) on %d host threads (os.cpu_count() = %d, "
                             "capped by affinity and the cgroup CPU quota): the whole %dx%d film at %d spp = %d samples in %.1f s; Rust reference not buildable here"
                             % (cores, os.cpu_count() or 0, W, H, spp, pc.camera_rays, dt),
                   "rays_per_sample": rays_per_sample, "Mrays_per_s": (pc.bounce_rays + pc.shadow_rays) / dt / 1e6,
                   "per_thread_Msamples_per_s": pc.camera_rays / dt / 1e6 / cores,
                   "reference_self_reported": {"Mrays_per_s": REFERENCE_SELF_REPORTED_MRAYS, "threads": REFERENCE_SELF_REPORTED_THREADS,
                                               "source": "data/config.toml:4-8 of the reference (its author's machine, CPU not named)",
                                               "as_Msamples_per_s_at_this_rays_per_sample": REFERENCE_SELF_REPORTED_MRAYS / rays_per_sample if rays_per_sample else None}}

        # ---- the checker next to the number: a smoke-size render of this scene, engine against oracle at the same seed, under the bars the
        # parity tests use (tests/parity_suite.py: L-inf < 1e-4 flat; a pixel whose own 8 ulp exceed that — values above ~128 — gets those)
        parity = None
        if args.cpu_seconds > 0:
            rd = pkg.api.render_desc(64, 64, 4, args.max_bounces, min_bounces=args.min_bounces, light_samples=L, seed=1, hero_wavelengths=args.hero, medium_aware=args.medium_aware)
            gfilm, gprof = scene.render(rd)
            ofilm, oprof = oscene.render(rd)
            ok = np.isfinite(ofilm[..., :3]) & np.isfinite(gfilm[..., :3])
            d = np.where(ok, np.abs(gfilm[..., :3].astype(np.float64) - ofilm[..., :3].astype(np.float64)), 0.0)
            parity = {"render": "%s 64x64, 4 spp, max_bounces %d, L = %d, seed 1: HIP engine against the CPU oracle (oracle/ptref.cpp; parity unpinned against the Rust reference)" % (args.scene, args.max_bounces, L),
                      "film_linf": float(d.max()), "film_rel": float((d / np.maximum(np.abs(ofilm[..., :3]), 1e-6)).max()), "film_linf_bar": 1e-4,
                      "pixels_over_flat_bar": int((d > 1e-4).any(axis=-1).sum()), "non_finite_mismatch": int((np.isfinite(ofilm) != np.isfinite(gfilm)).sum()),
                      "counters_equal": (gprof.camera_rays, gprof.bounce_rays, gprof.shadow_rays, gprof.env_hits) == (oprof.camera_rays, oprof.bounce_rays, oprof.shadow_rays, oprof.env_hits)}

        out = {
            "metric": metric_name,
            "value": value, "unit": "Msamples/s", "n_gpus": n_gpus, "ranks_seen": ranks_seen, "physical_gpus": physical_gpus, "rccl_version": rccl_version, "backend": (args.backend if use_dist else None),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload or ("C2: Cornell box (authored mesh + reference light/camera), %dx%d, PT+NEE, max_bounces=%d, min_bounces=1, "
                                                     "light_samples=%d, wavelengths 380-750 nm" % (W, H, args.max_bounces, L) if (args.scene, args.hero) == ("cornell_box", 1)
                                                     else "%s %dx%d, PT+NEE, max_bounces=%d, min_bounces=%d, light_samples=%d, %d wavelength(s) per path" %
                                                     (args.scene, W, H, args.max_bounces, args.min_bounces, L, args.hero)),
                       "workload_key": workload_key, "scene": args.scene, "spp_per_step_per_gpu": args.spp_per_step, "spp_per_step": S, "samples_per_step": total_samples / args.steps,
                       "parallelism": "film tiles 32x32 dealt along diagonals over %d GPU(s), one RCCL reduce of the XYZ film" % n_gpus,
                       "scaling_modes": {"weak": "every GPU renders 1/N of the pixels at N x %d spp per step: per-GPU work fixed (default)" % args.spp_per_step,
                                         "strong": "the frame is fixed at %d spp per step, its tiles dealt over the N GPUs: total work fixed (--scaling strong)" % args.spp_per_step,
                                         "this_run": args.scaling},
                       "device": engine.device_info()},
            "rays_per_s": {"segments": counts[2] / elapsed, "shadow": counts[7] / elapsed,
                           "total_Mrays": (counts[2] + counts[7]) / elapsed / 1e6},
            "segments_per_sample": counts[2] / total_samples,
            "roofline": roofline, "cpu_baseline": cpu, "strong": strong, "parity": parity,
        }
    else:
        out = None
    if use_dist:
        dist.barrier()   # (rank 0 has just spent its CPU-baseline seconds: the ranks leave together)
        dist.destroy_process_group()
    if out is not None:
        try:   # RCCL's version banner sits in the C library's stdout buffer until exit: push it out first
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print((RECORD_MARK if os.environ.get("PT_BENCH_PARENT") else "") + json.dumps(out), flush=True)   # the last line of stdout, after anything RCCL had to say


if __name__ == "__main__":
    main()
