"""Runs on one GPU: the per-GPU rate of bench.py's weak-scaling shards (1/N of the film at N x 120 spp, N = 1, 2, 4, 8) — what one rank of an
N-GPU run computes, without the other ranks.  Measured (r1o): 1419 / 1415 / 1471 for N = 1 / 2 / 4 and 1458-1468 for the eight shards of N = 8."""
import importlib, sys, time, torch
sys.path.insert(0, '.')
pkg = importlib.import_module("rust-pathtracer_amd")
engine = pkg.load(); scene = engine.create_scene(pkg.scene.cornell_box())
film = torch.zeros((1024, 1024, 4), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
import itertools
for n, r in itertools.chain(((n, n - 1) for n in (1, 2, 4)), ((8, r) for r in range(8))):
    S = 120 * n
    for k in range(4):
        if k == 1: torch.cuda.synchronize(); t0 = time.perf_counter()
        rd = pkg.api.render_desc(1024, 1024, S * 4, 8, light_samples=2, seed=1, shard=(r, n), first_sample=k * S, sample_count=S)
        prof = scene.render_device(rd, film.data_ptr(), stream)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print("   stage ms of the last step (generate, extend, shade, shadow, accumulate):", [round(1e3 * prof.kernel_seconds[i], 2) for i in range(5)])
    print("shard %d of %d: %d spp on 1/%d of the film: %.1f ms/step, %.1f Msamples/s per GPU" % (r, n, S, n, dt * 1e3, prof.camera_rays / dt / 1e6))
