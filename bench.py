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
        return path + 4 + 4                       # write path state + zero energy, read pixel id
    if stage == "extend":
        return 6 * 4 + hit                        # read ray (o, d), write hit record
    if stage == "shade":
        return path + hit + 4 + path + (3 + 7 * light_samples) * 4   # read state + hit + pixel id, write next state + light-sample item
    if stage == "shadow":
        return (3 + 7 * light_samples) * 4 + 8   # read item, read-modify-write the slot energy
    return 16 * 2                                 # accumulate: film pixel read-modify-write (energy reads are per sample, added below)


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
    ap.add_argument("--hero", type=int, default=1, help="wavelengths per path: 1, or 4 for the hero-wavelength variant (C5)")
    ap.add_argument("--workload", default=None, help="label for config.workload (default: derived from the arguments)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the oracle baseline (0 = skip)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every GPU renders 1/N of the pixels at N x --spp-per-step (per-GPU work fixed); strong: the frame is fixed at --spp-per-step, its tiles dealt over the N GPUs")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (nccl = RCCL) even for one process, so that the film reduce runs through RCCL")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    pkg = importlib.import_module("rust-pathtracer_amd")
    sharding = pkg.sharding
    rank, local_rank, world = sharding.rank_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    device_index = local_rank % max(1, torch.cuda.device_count())  # (a launcher that exposes one GPU per process shows it as device 0)
    torch.cuda.set_device(device_index)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
    n_gpus = world

    engine = pkg.load()
    builder = pkg.scene.SCENES[args.scene]()
    scene = engine.create_scene(builder)

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
                                 shard=sharding.shard(rank, n_gpus), first_sample=k * S, sample_count=S, hero_wavelengths=args.hero)
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
        kbytes[4] += 4 * sum(p.camera_rays for p in profs)          # accumulate also reads one energy per sample
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
        workload_key = {"scene": args.scene, "width": W, "height": H, "max_bounces": args.max_bounces, "min_bounces": args.min_bounces,
                        "light_samples": L, "hero": args.hero, "spp_per_step": S, "n_gpus": n_gpus}
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
                            "peaks_source": "profiles/r2a_valu_issue.txt (tools/microbench/valu_issue.hip on MI355X): fma/add/mul/mov/xor issue in 2 cycles per "
                                            "wave64, min/max/cmp/cndmask/integer/f64/packed/SGPR-operand instructions in 4", "source": traffic_src}
                break
        except Exception:
            traffic, valu = None, None
        # SURVEY.md 8(d)'s a-priori byte model next to the engine's own record sizes: B(sample) = 32 + D (192 + 72 L)
        survey_bytes = 32.0 + d_bar * (192.0 + 72.0 * L)
        roofline = {"bound": "hbm", "kernel": "k_" + STAGES[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
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
            pp, dt = orender(pkg.api.render_desc(W, H, 1, args.max_bounces, min_bounces=args.min_bounces, light_samples=L, seed=1, shard=(0, 4), hero_wavelengths=args.hero))
            rate = pp.camera_rays / dt
            pp, dt = orender(pkg.api.render_desc(W, H, max(1, min(16, int(rate * 2.0 / (W * H)))), args.max_bounces, min_bounces=args.min_bounces, light_samples=L, seed=1, hero_wavelengths=args.hero))
            rate = pp.camera_rays / dt
            spp = max(1, int(rate * args.cpu_seconds / (W * H)))
            pc, dt = orender(pkg.api.render_desc(W, H, spp, args.max_bounces, min_bounces=args.min_bounces, light_samples=L, seed=1, hero_wavelengths=args.hero))
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

        out = {
            "metric": metric_name,
            "value": value, "unit": "Msamples/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
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
            "roofline": roofline, "cpu_baseline": cpu,
        }
    else:
        out = None
    if use_dist:
        dist.destroy_process_group()
    if out is not None:
        try:   # RCCL's version banner sits in the C library's stdout buffer until exit: push it out first
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)   # the last line of stdout, after anything RCCL had to say


if __name__ == "__main__":
    main()
