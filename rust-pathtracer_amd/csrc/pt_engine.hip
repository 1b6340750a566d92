// pt_engine.hip — the host side of the HIP engine behind include/pt_api.h (gfx950 / MI355X only).
//
// Scene upload (one flat blob + texels per device), buffer management, the pass loop — per bounce one launch each of extend -> shade ->
// shadow over segmented SoA queues in HBM, with HIP events around every launch so that per-stage device time is measured inside the
// timed region —, the choice of kernel variant per scene (pt_launch.h: staging mode x traversal form x wavelengths x what the scene can
// need), the probes of the trait surface, and pt_render_multi: one replica, host thread and stream per device and one RCCL reduce.
// The kernels themselves are templates in pt_kernels.h, instantiated per family in pt_kern_*.hip.  No CPU fallback: every entry point
// fails with PT_ERR_NO_DEVICE when HIP has no device.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "pt_kernels.h"   /* first: it switches on the wave-level device code of pt_device.h */
#include "../../include/pt_api.h"
#include "pt_error.h"
#include "pt_plan.h"
#include "pt_scene_host.h"

using namespace ptd;
using namespace ptk;

namespace {

thread_local std::string g_error;
pt_status fail(pt_status st, const std::string& msg) { g_error = msg; return st; }
}  // namespace
void pt_set_error(const std::string& message) { g_error = message; }
namespace {

#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess)                                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? PT_ERR_OUT_OF_MEMORY : (e_ == hipErrorNoDevice ? PT_ERR_NO_DEVICE : PT_ERR_DEVICE), \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                                    \
    } while (0)

// mode 0: generate_and_evaluate(lambda, wi, s2) -> f, wo, pdf ; 1: bsdf(lambda, wi, wo) -> f, pdf ; 2: emission(lambda, wi) ; 3: curve(lambda)
__global__ void __launch_bounds__(kBlock) k_probe_material(const uint32_t* __restrict__ blob, const float* __restrict__ tex, int mode, uint32_t record, uint32_t n,
                                                          const float* __restrict__ lambda, const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ f, float* __restrict__ wo, float* __restrict__ pdf) {
    SceneView s; s.w = blob; s.tex = tex; s.m = blob + blob[PT_HDR_CORE_WORDS];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (mode == 0) {
            F3 w; material_sample(s, record, lambda[i], 0.5f, 0.5f, b[2 * i], b[2 * i + 1], f3(a[3 * i], a[3 * i + 1], a[3 * i + 2]), &f[i], &w, &pdf[i]);
            wo[3 * i] = w.x; wo[3 * i + 1] = w.y; wo[3 * i + 2] = w.z;
        } else if (mode == 1) {
            material_bsdf(s, record, lambda[i], 0.5f, 0.5f, f3(a[3 * i], a[3 * i + 1], a[3 * i + 2]), f3(b[3 * i], b[3 * i + 1], b[3 * i + 2]), &f[i], &pdf[i]);
        } else if (mode == 2) {
            f[i] = material_emission(s, record, lambda[i], f3(a[3 * i], a[3 * i + 1], a[3 * i + 2]));
        } else {
            f[i] = curve_eval(s, record, lambda[i]);
        }
    }
}
__global__ void __launch_bounds__(kBlock) k_probe_numerics(int which, uint32_t n, const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float r;
        switch (which) {
            case 0: r = pt_sin(x[i]); break;
            case 1: r = pt_cos(x[i]); break;
            case 2: r = pt_exp(x[i]); break;
            case 3: r = pt_pow(x[i], y[i]); break;
            case 4: r = pt_acos(x[i]); break;
            case 5: r = pt_atan2(x[i], y[i]); break;
            case 6: r = (float)pt_exp64((double)x[i]); break;
            case 7: r = (float)pt_log64((double)x[i]); break;
            case 8: r = x[i] / y[i]; break;
            case 9: r = pt_sqrt(x[i]); break;
            case 10: r = x[i] * y[i] + x[i]; break;  // must NOT be contracted to an fma
            default: r = 0.0f;
        }
        out[i] = r;
    }
}

// ------------------------------------------------------------------------------------------------ host side
// A device allocation that is freed on every way out of its scope (the probes below return early on any HIP error).
struct DevBuf {
    void* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 4); }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

constexpr uint32_t kUnitCounters = 2 * 64 + 2;   // two parked launches per bounce, max_bounces <= 64
struct DeviceBuffers {
    uint32_t capacity = 0, light_samples = 0, nl = 0;
    uint32_t *paths_a = nullptr, *paths_b = nullptr, *hits = nullptr, *shadow = nullptr, *pixels = nullptr, *counts = nullptr, *park = nullptr;
    uint32_t* unit_counters = nullptr;   // parked kernels with dynamic units: one counter per launch of a pass (kUnitCounters), zeroed per pass
    float* energy = nullptr;
    unsigned long long* block_stats = nullptr;
    size_t pixel_capacity = 0;
    int grid = 0;  // segments per queue == workgroups per launch
    void release() {
        hipFree(paths_a); hipFree(paths_b); hipFree(hits); hipFree(shadow); hipFree(pixels); hipFree(counts); hipFree(energy); hipFree(block_stats); hipFree(park);
        hipFree(unit_counters);
        *this = DeviceBuffers();
    }
};

}  // namespace

struct pt_scene {
    pth::HostScene host;
    uint32_t* d_blob = nullptr;
    float* d_tex = nullptr;
    uint32_t blob_words = 0;
    int lds_mode = 0;  // PT_LDS_*
    uint32_t lacks = 0; // PT_SCENE_* bits: what the scene does not hold (kernel forms without it)
    int device = 0, num_cus = 0;
    DeviceBuffers buf;
    std::vector<hipEvent_t> events;  // pairs (start, stop), grown on demand
    float* film_cache = nullptr;     // pt_render's device film, kept between calls
    size_t film_cache_bytes = 0;
    std::vector<pt_scene*> replicas; // pt_render_multi: this scene on the other devices, by device index (nullptr = not made yet / this one)
};

namespace {

std::string g_device_info;

pt_status ensure_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(PT_ERR_NO_DEVICE, "no HIP device available: the product path has no CPU fallback");
    return PT_OK;
}


// Segment capacity for n items over `grid` segments, rounded up to 64 items so that every segment starts on a
// 256-byte boundary in every field.
uint32_t segment_capacity(uint32_t n, int grid) {
    uint32_t c = (n + (uint32_t)grid - 1) / (uint32_t)grid;
    return (c + 63u) & ~63u;
}

pt_status ensure_buffers(pt_scene* sc, uint32_t capacity, uint32_t light_samples, size_t n_pixels, int grid, uint32_t nl) {
    DeviceBuffers& b = sc->buf;
    uint32_t total = segment_capacity(capacity, grid) * (uint32_t)grid;
    if (b.capacity < total || b.light_samples < light_samples || b.grid != grid || b.nl < nl) {
        hipFree(b.paths_a); hipFree(b.paths_b); hipFree(b.hits); hipFree(b.shadow); hipFree(b.energy); hipFree(b.counts); hipFree(b.block_stats); hipFree(b.park);
        b.paths_a = b.paths_b = b.hits = b.shadow = b.counts = b.park = nullptr; b.energy = nullptr; b.block_stats = nullptr; b.capacity = 0;
        uint32_t ls = light_samples > b.light_samples ? light_samples : b.light_samples;
        uint32_t nlmax = nl > b.nl ? nl : b.nl;
        size_t path_fields = nlmax == 4 ? Layout<4>::path_fields : Layout<1>::path_fields;
        size_t sh_fields = nlmax == 4 ? Layout<4>::shadow_fields(ls ? ls : 1) : Layout<1>::shadow_fields(ls ? ls : 1);
        HIP_TRY(hipMalloc(&b.paths_a, sizeof(uint32_t) * path_fields * total));
        HIP_TRY(hipMalloc(&b.paths_b, sizeof(uint32_t) * path_fields * total));
        HIP_TRY(hipMalloc(&b.hits, sizeof(uint32_t) * (size_t)HS_FIELDS * total));
        HIP_TRY(hipMalloc(&b.shadow, sizeof(uint32_t) * sh_fields * total));
        HIP_TRY(hipMalloc(&b.energy, sizeof(float) * (size_t)nlmax * total));
        b.nl = nlmax;
        HIP_TRY(hipMalloc(&b.counts, sizeof(uint32_t) * 3 * (size_t)grid));
        HIP_TRY(hipMalloc(&b.block_stats, sizeof(unsigned long long) * BS_FIELDS * (size_t)grid));
        if (sc->host.blob[PT_HDR_FLAGS] & PT_FLAG_SWEEP_WALKS) HIP_TRY(hipMalloc(&b.park, sizeof(uint32_t) * kParkFields * kParkCap * (size_t)grid));
        if (!b.unit_counters) HIP_TRY(hipMalloc(&b.unit_counters, sizeof(uint32_t) * kUnitCounters));
        b.capacity = total; b.light_samples = ls; b.grid = grid;
    }
    if (b.pixel_capacity < n_pixels) {
        hipFree(b.pixels); b.pixels = nullptr;
        HIP_TRY(hipMalloc(&b.pixels, sizeof(uint32_t) * n_pixels));
        b.pixel_capacity = n_pixels;
    }
    return PT_OK;
}

template <typename K, typename... Args>
void launch(K kernel, uint32_t lds_bytes, int grid, hipStream_t stream, Args... args) {
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), lds_bytes, stream, args...);
}

uint32_t env_u32(const char* name, uint32_t dflt) {
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    return (uint32_t)strtoul(v, nullptr, 10);
}

pt_status render_impl(pt_scene* sc, const pt_render_desc* rdp, float* d_film, hipStream_t stream, pt_profile* profile) {
    if (!sc || !rdp || !d_film) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    pt_render_desc rd;
    std::string err;
    if (!pth::normalize_render_desc(*rdp, (uint32_t)sc->host.cameras.size(), &rd, &err)) return fail(PT_ERR_INVALID_ARGUMENT, err);
    HIP_TRY(hipSetDevice(sc->device));

    std::vector<uint32_t> pixels = pth::shard_pixels(rd.width, rd.height, rd.tile_width, rd.tile_height, rd.shard_index, rd.shard_count);
    uint32_t capacity = env_u32("PT_AMD_BATCH", 1u << 27);  // path slots per pass (128 Mi ~ 32 GB of queues of the 288 GB; tools/sweep.sh)
    if (capacity < 1024) capacity = 1024;
    uint64_t want = (uint64_t)pixels.size() * rd.sample_count;
    if (want < capacity) capacity = (uint32_t)(want ? want : 1);
    const uint32_t blocks_per_cu = env_u32("PT_AMD_BLOCKS_PER_CU", 64);
    if (blocks_per_cu == 0 || blocks_per_cu > 1024) return fail(PT_ERR_INVALID_ARGUMENT, "PT_AMD_BLOCKS_PER_CU must be in 1..1024");
    const int grid = sc->num_cus * (int)blocks_per_cu;  // queue segments = workgroups per launch
    // a pass holds at least one whole phase of one pixel (pt_plan.cpp): the queues must too (NaiveRenderer settings: phase = spp)
    { const uint32_t phase = rd.sample_count < rd.phase_samples ? rd.sample_count : rd.phase_samples; if (capacity < phase) capacity = phase; }
    const bool hero = rd.hero_wavelengths == 4;
    if (hero && capacity > (1u << 26)) capacity = 1u << 26;  // 4-wavelength queues are ~1.5x wider: 64 Mi slots ~ 24 GB
    // (the medium-aware walk keeps its two extra path fields where the hero layout keeps the passengers' throughputs)
    pt_status st = ensure_buffers(sc, capacity, rd.light_samples, pixels.size() ? pixels.size() : 1, grid, (hero || rd.medium_aware) ? 4u : 1u);
    if (st != PT_OK) return st;
    DeviceBuffers& b = sc->buf;
    if (!pixels.empty()) HIP_TRY(hipMemcpyAsync(b.pixels, pixels.data(), sizeof(uint32_t) * pixels.size(), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemsetAsync(d_film, 0, sizeof(float) * 4 * (size_t)rd.width * rd.height, stream));
    HIP_TRY(hipMemsetAsync(b.block_stats, 0, sizeof(unsigned long long) * BS_FIELDS * (size_t)grid, stream));

    RenderParams rp;
    memset(&rp, 0, sizeof(rp));
    rp.seed = rd.seed; rp.width = rd.width; rp.height = rd.height;
    rp.min_bounces = rd.min_bounces; rp.max_bounces = rd.max_bounces; rp.light_samples = rd.light_samples; rp.only_direct = rd.only_direct;
    rp.wavelength_lo = rd.wavelength_lo; rp.wavelength_span = rd.wavelength_hi - rd.wavelength_lo;
    rp.spp = rd.spp; rp.range_end = rd.first_sample + rd.sample_count;
    rp.normalize = (rd.first_sample == 0 && rd.sample_count == rd.spp) ? 1u : 0u;
    rp.phase = rd.phase_samples;
    rp.camera = pth::camera_params(sc->host.cameras[rd.camera_index], (float)rd.width / (float)rd.height);
    rp.energy_stride = b.capacity;

    const int mode = sc->lds_mode;
    const uint32_t lds_bytes = mode == PT_LDS_ALL ? sc->blob_words * 4u : (mode == PT_LDS_CORE ? sc->host.blob[PT_HDR_CORE_WORDS] * 4u : 0u);
    const bool sweep = sc->host.blob[PT_HDR_SWEEP_OFF] != 0 && !(sc->host.blob[PT_HDR_FLAGS] & PT_FLAG_NO_SWEEP);
    // the sweep table holds walked meshes: rays that reach one are parked and resumed in full waves (PT_AMD_NO_PARK=1: in line)
    const bool walks = (sc->host.blob[PT_HDR_FLAGS] & PT_FLAG_SWEEP_WALKS) != 0;
    const bool parked = sweep && walks && b.park != nullptr && !env_u32("PT_AMD_NO_PARK", 0);
    // PT_AMD_POOL=1: phase 3 of a pure sweep scene pooled per wave (sweep_run_pooled).  Bit-identical, but measured slower than the lane
    // loop on MI355X (C2: k_extend 3155 vs 2475 us, k_shadow 5421 vs 4677 us; DESIGN.md section 5 has the breakdown), so it is not the default.
    const bool pooled = sweep && !walks && mode == PT_LDS_ALL && lds_bytes + pool_lds_bytes() <= kLdsBlobLimitBytes && env_u32("PT_AMD_POOL", 0) != 0;
    // (walked meshes in line under PT_AMD_NO_PARK, and every partly staged or unstaged blob: the run-time choice of PT_FORM_ANY)
    const int trav_form = parked ? PT_FORM_PARKED : (mode != PT_LDS_ALL || (sweep && walks)) ? PT_FORM_ANY : pooled ? PT_FORM_POOLED : sweep ? PT_FORM_SWEEP : PT_FORM_WALK;
    // The parked kernels take units of work from a counter, a few persistent workgroups per CU, when the whole blob is staged in LDS
    // (C3: k_extend 9175 -> 7880 us, k_shadow 8008 -> 7105, 487 -> 543 Msamples/s: park lists that live across units keep the drains
    // full).  With the mesh in HBM/L2 (C4) the static form wins, 1128 vs 1083 Msamples/s: a wave's parked rays then come from one
    // region of the film and walk the same part of the mesh.  PT_AMD_PARK_DYNAMIC=0 / 1 forces either.
    const bool park_dynamic = parked && env_u32("PT_AMD_PARK_DYNAMIC", mode == PT_LDS_ALL ? 1 : 0) != 0;
    const int dyn_grid = sc->num_cus * (int)env_u32("PT_AMD_PARK_BLOCKS_PER_CU", 4);
    LaunchCfg cfg{grid, lds_bytes, stream, mode};
    cfg.dyn_grid = dyn_grid < grid ? dyn_grid : grid;
    cfg.lacks = sc->lacks;
    const SceneArgs sargs{sc->d_blob, sc->blob_words, sc->d_tex};
    // light samples can pick the environment only if env_sampling_probability > 0: otherwise k_shade is the form without that branch
    float env_prob; std::memcpy(&env_prob, &sc->host.blob[PT_HDR_ENV_PROB], sizeof env_prob);
    bool has_ggx = false;
    for (uint32_t i = 0; i < sc->host.blob[PT_HDR_MATERIAL_COUNT]; ++i) {
        const uint32_t kind = sc->host.blob[sc->host.blob[PT_HDR_MATERIAL_OFF] + i * PT_MAT_WORDS + PT_MAT_KIND];
        has_ggx = has_ggx || kind == PT_MATERIAL_GGX || kind == PT_MATERIAL_PASSTHROUGH;
    }
    // (a PassthroughFilter lives in the forms that hold the GGX code)
    const int shade_form = rd.medium_aware ? PT_SHADE_MEDIUM
                         : (env_prob != 0.0f || env_u32("PT_AMD_SHADE_FORM", 0) == 2) ? PT_SHADE_FULL : (has_ggx || env_u32("PT_AMD_SHADE_FORM", 0) == 1) ? PT_SHADE_NO_ENV : PT_SHADE_LEAN;
    // k_shade that traces its own segments (PT_AMD_FUSE): exists for the pure sweep form of a fully staged, transform-free scene shaded by the lean form
    cfg.fuse = env_u32("PT_AMD_FUSE", 0) != 0 && trav_form == PT_FORM_SWEEP && shade_form == PT_SHADE_LEAN && (cfg.lacks & PT_SCENE_NO_XF) != 0;
    const uint32_t bounce_limit = rd.only_direct ? 1u : rd.max_bounces;
    const bool timing = env_u32("PT_AMD_STAGE_TIMING", 1) != 0;
    double stage_ms[ST_COUNT] = {0, 0, 0, 0, 0};
    uint64_t stage_launches[ST_COUNT] = {0, 0, 0, 0, 0};
    // (a queue's tiles are laid out by the number of fields this render uses, pt_stages.h: the buffers are sized for the widest layout seen)
    const uint32_t path_fields = hero ? Layout<4>::path_fields : (rd.medium_aware ? (uint32_t)PS_FIELDS + 2u : Layout<1>::path_fields);
    const uint32_t item_fields = hero ? Layout<4>::shadow_fields(rd.light_samples ? rd.light_samples : 1) : Layout<1>::shadow_fields(rd.light_samples ? rd.light_samples : 1);
    Queue qa{b.paths_a, b.capacity, path_fields}, qb{b.paths_b, b.capacity, path_fields}, qh{b.hits, b.capacity, HS_FIELDS}, qs{b.shadow, b.capacity, item_fields};
    uint32_t* live[2] = {b.counts, b.counts + grid};  // per-segment live-path counts, ping-pong with the path queues
    uint32_t* nshadow = b.counts + 2 * grid;          // per-segment light-sample item counts

    auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipStreamSynchronize(stream));
    t0 = std::chrono::steady_clock::now();
    // HIP events around every launch, recorded on the launch stream and read back after the final sync, so the
    // per-stage device time is measured inside the timed region without stalling it.
    std::vector<int> event_stage;
    auto timed = [&](int stage, auto&& fn) {
        size_t k = event_stage.size();
        if (timing) {
            while (sc->events.size() < 2 * (k + 1)) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) break; sc->events.push_back(e); }
            if (sc->events.size() >= 2 * (k + 1)) hipEventRecord(sc->events[2 * k], stream);
        }
        fn();
        if (timing && sc->events.size() >= 2 * (k + 1)) { hipEventRecord(sc->events[2 * k + 1], stream); event_stage.push_back(stage); }
        stage_launches[stage]++;
    };

    // the planner's capacity is in items; segments round up, so plan with what surely fits
    std::vector<pth::Pass> passes = pth::plan_passes((uint32_t)pixels.size(), rd.first_sample, rd.sample_count, capacity, rd.phase_samples);
    uint64_t camera_rays = 0, accumulated_pixels = 0;
    for (const pth::Pass& pass : passes) {
        accumulated_pixels += pass.pixel_count;
        rp.chunk_pixels = pass.pixel_count; rp.first_sample = pass.first_sample; rp.pass_samples = pass.sample_count;
        uint32_t n = pass.pixel_count * pass.sample_count;
        if (n > b.capacity) return fail(PT_ERR_DEVICE, "internal: a pass of " + std::to_string(n) + " slots exceeds the queue capacity " + std::to_string(b.capacity));
        uint32_t seg_cap = segment_capacity(n, grid);
        camera_rays += n;
        const uint32_t* d_px = b.pixels + pass.pixel_begin;
        if (park_dynamic) HIP_TRY(hipMemsetAsync(b.unit_counters, 0, sizeof(uint32_t) * kUnitCounters, stream));
        timed(ST_GENERATE, [&] {
            if (hero) hipLaunchKernelGGL(k_generate<4>, dim3(grid), dim3(kBlock), 0, stream, rp, d_px, qa, b.energy, n, seg_cap, live[0]);
            else hipLaunchKernelGGL(k_generate<1>, dim3(grid), dim3(kBlock), 0, stream, rp, d_px, qa, b.energy, n, seg_cap, live[0]);
        });
        for (uint32_t bounce = 0; bounce < bounce_limit; ++bounce) {
            Queue qin = (bounce & 1) ? qb : qa, qout = (bounce & 1) ? qa : qb;
            uint32_t *cin = live[bounce & 1], *cout = live[(bounce + 1) & 1];
            // kernel variant = staging mode (PT_LDS_*) x traversal form x wavelengths per path (pt_launch.h)
            if (park_dynamic) cfg.unit_counter = b.unit_counters + 2 * bounce;
            if (!cfg.fuse) timed(ST_EXTEND, [&] { launch_extend(cfg, trav_form, sargs, qin, qh, seg_cap, cin, b.park); });
            if (park_dynamic) cfg.unit_counter = b.unit_counters + 2 * bounce + 1;
            timed(ST_SHADE, [&] { launch_shade(cfg, hero ? 4 : 1, shade_form, sargs, rp, bounce, d_px, qin, qh, qout, qs, b.energy, seg_cap, cin, cout, nshadow, b.block_stats); });
            if (rd.light_samples > 0)   // (shade_form FULL = the scene can produce environment rays)
                timed(ST_SHADOW, [&] { launch_shadow(cfg, trav_form, hero ? 4 : 1, shade_form == PT_SHADE_FULL || shade_form == PT_SHADE_MEDIUM, sargs, rd.light_samples, qs, b.energy, b.capacity, seg_cap, nshadow, b.park, qh); });
        }
        timed(ST_ACCUMULATE, [&] {
            if (hero) hipLaunchKernelGGL(k_accumulate<4>, dim3(grid), dim3(kBlock), 0, stream, rp, d_px, b.energy, d_film);
            else hipLaunchKernelGGL(k_accumulate<1>, dim3(grid), dim3(kBlock), 0, stream, rp, d_px, b.energy, d_film);
        });
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    auto t1 = std::chrono::steady_clock::now();
    for (size_t k = 0; k < event_stage.size(); ++k) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, sc->events[2 * k], sc->events[2 * k + 1]) == hipSuccess) stage_ms[event_stage[k]] += ms;
    }
    if (profile) {
        memset(profile, 0, sizeof(*profile));
        std::vector<unsigned long long> bs((size_t)grid * BS_FIELDS);
        HIP_TRY(hipMemcpy(bs.data(), b.block_stats, sizeof(unsigned long long) * bs.size(), hipMemcpyDeviceToHost));
        unsigned long long c[BS_FIELDS] = {0, 0, 0, 0, 0};
        for (int g = 0; g < grid; ++g) for (int k = 0; k < BS_FIELDS; ++k) c[k] += bs[(size_t)g * BS_FIELDS + k];
        profile->camera_rays = camera_rays;
        profile->bounce_rays = c[BS_VERTICES] + camera_rays;  // vertices.len() counts the camera vertex (utils.rs:375)
        profile->shadow_rays = c[BS_SHADOW_RAYS];
        profile->env_hits = c[BS_ENV_HITS];
        profile->seconds = std::chrono::duration<double>(t1 - t0).count();
        for (int i = 0; i < ST_COUNT; ++i) { profile->kernel_seconds[i] = stage_ms[i] * 1e-3; profile->kernel_launches[i] = stage_launches[i]; }
        profile->stage_items[ST_GENERATE] = camera_rays; profile->stage_items[ST_EXTEND] = c[BS_SEGMENTS]; profile->stage_items[ST_SHADE] = c[BS_SEGMENTS];
        profile->stage_items[ST_SHADOW] = c[BS_ITEMS]; profile->stage_items[ST_ACCUMULATE] = accumulated_pixels;
    }
    return PT_OK;
}

pt_status probe_material(pt_scene* sc, int mode, uint32_t record, size_t n, const float* lambda, const float* a, size_t a_w, const float* b, size_t b_w,
                         float* f, float* wo, float* pdf) {
    HIP_TRY(hipSetDevice(sc->device));
    DevBuf dl, da, db, df, dwo, dp;
    size_t m = n ? n : 1;
    HIP_TRY(dl.alloc(4 * m)); HIP_TRY(da.alloc(4 * m * 3)); HIP_TRY(db.alloc(4 * m * 3));
    HIP_TRY(df.alloc(4 * m)); HIP_TRY(dwo.alloc(4 * m * 3)); HIP_TRY(dp.alloc(4 * m));
    HIP_TRY(hipMemcpy(dl.p, lambda, 4 * n, hipMemcpyHostToDevice));
    if (a) HIP_TRY(hipMemcpy(da.p, a, 4 * n * a_w, hipMemcpyHostToDevice));
    if (b) HIP_TRY(hipMemcpy(db.p, b, 4 * n * b_w, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe_material, dim3(256), dim3(kBlock), 0, 0, sc->d_blob, sc->d_tex, mode, record, (uint32_t)n, dl.as<float>(), da.as<float>(), db.as<float>(),
                       df.as<float>(), dwo.as<float>(), dp.as<float>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (f) HIP_TRY(hipMemcpy(f, df.p, 4 * n, hipMemcpyDeviceToHost));
    if (wo) HIP_TRY(hipMemcpy(wo, dwo.p, 4 * n * 3, hipMemcpyDeviceToHost));
    if (pdf) HIP_TRY(hipMemcpy(pdf, dp.p, 4 * n, hipMemcpyDeviceToHost));
    return PT_OK;
}

}  // namespace

extern "C" {

const char* pt_last_error(void) { return g_error.c_str(); }

const char* pt_device_info(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { g_device_info = "no HIP device"; return g_device_info.c_str(); }
    int dev = 0; hipGetDevice(&dev);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, dev);
    char buf[256];
    snprintf(buf, sizeof(buf), "%s %s %d CUs, %.1f GB, LDS/block %zu KB", p.name, p.gcnArchName, p.multiProcessorCount,
             (double)p.totalGlobalMem / 1e9, p.sharedMemPerBlock / 1024);
    g_device_info = buf;
    return g_device_info.c_str();
}

// Device side of a scene: the blob and the texels on the current device, the staging mode, the kernels' LDS allowance.
static pt_status scene_to_device(pt_scene* sc) {
    hipError_t e = hipGetDevice(&sc->device);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, sc->device);
    if (e != hipSuccess) return fail(PT_ERR_NO_DEVICE, hipGetErrorString(e));
    sc->num_cus = prop.multiProcessorCount;
    if (env_u32("PT_AMD_EXACT_SLAB", 0)) sc->host.blob[PT_HDR_FLAGS] |= PT_FLAG_EXACT_SLAB;
    if (env_u32("PT_AMD_NO_CULL", 0)) sc->host.blob[PT_HDR_FLAGS] |= PT_FLAG_NO_CULL;
    if (env_u32("PT_AMD_NO_SWEEP", 0)) sc->host.blob[PT_HDR_FLAGS] |= PT_FLAG_NO_SWEEP;
    if (env_u32("PT_AMD_NO_MESH_SWEEP", 0)) sc->host.blob[PT_HDR_FLAGS] |= PT_FLAG_NO_MESH_SWEEP;
    if (env_u32("PT_AMD_NO_KNOWN_LIGHT", 0)) sc->host.blob[PT_HDR_FLAGS] |= PT_FLAG_NO_KNOWN_LIGHT;
    sc->blob_words = (uint32_t)sc->host.blob.size();
    {   // no instance carries a transform (the Cornell box): the forms without the matrix paths (PT_AMD_GENERAL_FORMS=1 keeps the general ones)
        bool any_xf = false;
        const std::vector<uint32_t>& bl = sc->host.blob;
        for (uint32_t i = 0; i < bl[PT_HDR_INSTANCE_COUNT]; ++i) any_xf = any_xf || (bl[bl[PT_HDR_INSTANCE_OFF] + i * PT_INST_WORDS + PT_INST_FLAGS] & 1u) != 0u;
        sc->lacks = (!any_xf && !env_u32("PT_AMD_GENERAL_FORMS", 0)) ? PT_SCENE_NO_XF : 0u;
    }
    const bool no_lds = env_u32("PT_AMD_NO_LDS", 0) != 0;
    const uint32_t all_limit = env_u32("PT_AMD_LDS_ALL_LIMIT", kLdsAllLimitBytes);   // (experiments: the largest blob staged whole)
    sc->lds_mode = no_lds ? PT_LDS_NONE : (sc->blob_words * 4 <= (all_limit < kLdsBlobLimitBytes ? all_limit : kLdsBlobLimitBytes) ? PT_LDS_ALL
                 : (sc->host.blob[PT_HDR_CORE_WORDS] * 4 <= kLdsBlobLimitBytes && !env_u32("PT_AMD_NO_CORE_LDS", 0) ? PT_LDS_CORE : PT_LDS_NONE));
    e = hipMalloc(&sc->d_blob, sizeof(uint32_t) * sc->host.blob.size());
    if (e == hipSuccess) e = hipMalloc(&sc->d_tex, sizeof(float) * (sc->host.tex.size() + 4));
    if (e == hipSuccess) e = hipMemcpy(sc->d_blob, sc->host.blob.data(), sizeof(uint32_t) * sc->host.blob.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(sc->d_tex, sc->host.tex.data(), sizeof(float) * sc->host.tex.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? PT_ERR_OUT_OF_MEMORY : PT_ERR_DEVICE, hipGetErrorString(e));
    if (sc->lds_mode != PT_LDS_NONE) {
        e = allow_lds_extend(kLdsBlobLimitBytes);
        if (e == hipSuccess) e = allow_lds_shade(kLdsBlobLimitBytes);
        if (e == hipSuccess) e = allow_lds_shadow(kLdsBlobLimitBytes);
        if (e != hipSuccess) return fail(PT_ERR_DEVICE, std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(e));
    }
    return PT_OK;
}

pt_status pt_scene_create(const pt_scene_desc* desc, pt_scene** out) {
    if (!desc || !out) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    pt_status st = ensure_device();
    if (st != PT_OK) return st;
    pt_scene* sc = new pt_scene();
    std::string err;
    if (!pth::build_host_scene(*desc, &sc->host, &err)) { delete sc; return fail(PT_ERR_INVALID_ARGUMENT, err); }
    st = scene_to_device(sc);
    if (st != PT_OK) { const std::string msg = g_error; pt_scene_destroy(sc); g_error = msg; return st; }
    *out = sc;
    return PT_OK;
}

void pt_scene_destroy(pt_scene* sc) {
    if (!sc) return;
    for (pt_scene* r : sc->replicas) if (r) pt_scene_destroy(r);
    sc->replicas.clear();
    hipSetDevice(sc->device);
    sc->buf.release();
    hipFree(sc->d_blob); hipFree(sc->d_tex); hipFree(sc->film_cache);
    for (auto& e : sc->events) hipEventDestroy(e);
    delete sc;
}

pt_status pt_render_device(pt_scene* sc, const pt_render_desc* rd, void* film_device, void* hip_stream, pt_profile* profile) {
    return render_impl(sc, rd, static_cast<float*>(film_device), static_cast<hipStream_t>(hip_stream), profile);
}

pt_status pt_render(pt_scene* sc, const pt_render_desc* rd, float* film, pt_profile* profile) {
    if (!sc || !rd || !film) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    if (rd->width == 0 || rd->height == 0) return fail(PT_ERR_INVALID_ARGUMENT, "width and height must be positive");
    HIP_TRY(hipSetDevice(sc->device));
    const size_t bytes = sizeof(float) * 4 * (size_t)rd->width * rd->height;
    if (sc->film_cache_bytes < bytes) {   // the device film lives as long as the scene: a sequence of renders allocates it once
        if (sc->film_cache) hipFree(sc->film_cache);
        sc->film_cache = nullptr; sc->film_cache_bytes = 0;
        HIP_TRY(hipMalloc(&sc->film_cache, bytes));
        sc->film_cache_bytes = bytes;
    }
    pt_status st = render_impl(sc, rd, sc->film_cache, nullptr, profile);
    if (st != PT_OK) return st;
    HIP_TRY(hipMemcpy(film, sc->film_cache, bytes, hipMemcpyDeviceToHost));
    return PT_OK;
}

uint32_t pt_device_count(void) {
    int n = 0;
    return (hipGetDeviceCount(&n) == hipSuccess && n > 0) ? (uint32_t)n : 0u;
}

// RCCL, bound on first use: a single-process render pays nothing for it, and the library loads on machines without it.
namespace {
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) comm_init_all = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclReduce) reduce = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    bool ok() const { return comm_init_all && comm_destroy && reduce && group_start && group_end && error_string; }
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (r.handle) break; }
        if (!r.handle) return;
        r.comm_init_all = reinterpret_cast<decltype(r.comm_init_all)>(dlsym(r.handle, "ncclCommInitAll"));
        r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(r.handle, "ncclCommDestroy"));
        r.reduce = reinterpret_cast<decltype(r.reduce)>(dlsym(r.handle, "ncclReduce"));
        r.group_start = reinterpret_cast<decltype(r.group_start)>(dlsym(r.handle, "ncclGroupStart"));
        r.group_end = reinterpret_cast<decltype(r.group_end)>(dlsym(r.handle, "ncclGroupEnd"));
        r.error_string = reinterpret_cast<decltype(r.error_string)>(dlsym(r.handle, "ncclGetErrorString"));
    });
    return r;
}
}  // namespace

pt_status pt_render_multi(pt_scene* sc, const pt_render_desc* rdp, uint64_t device_mask, float* film, pt_profile* profile) {
    if (!sc || !rdp || !film) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    if (rdp->width == 0 || rdp->height == 0) return fail(PT_ERR_INVALID_ARGUMENT, "width and height must be positive");
    if (rdp->shard_count > 1) return fail(PT_ERR_INVALID_ARGUMENT, "pt_render_multi deals the tiles itself: shard_count must be 0");
    const uint32_t visible = pt_device_count();
    if (visible == 0) return fail(PT_ERR_NO_DEVICE, "no HIP device available: the product path has no CPU fallback");
    std::vector<int> devices;
    for (uint32_t d = 0; d < visible && d < 64; ++d) if (device_mask == 0 || ((device_mask >> d) & 1ull)) devices.push_back((int)d);
    if (devices.empty()) return fail(PT_ERR_INVALID_ARGUMENT, "device_mask names no visible HIP device");
    const int n = (int)devices.size();
    // PT_AMD_MULTI_RCCL=1: take the RCCL reduce even for one device (the call path of a multi-device node, exercised on a single GPU)
    const bool use_rccl = n > 1 || env_u32("PT_AMD_MULTI_RCCL", 0) != 0;
    if (!use_rccl && devices[0] == sc->device) return pt_render(sc, rdp, film, profile);

    // one replica per device (this scene itself on its own device), made on first use and kept
    if (sc->replicas.size() < visible) sc->replicas.resize(visible, nullptr);
    std::vector<pt_scene*> scene_of(n, nullptr);
    for (int i = 0; i < n; ++i) {
        const int d = devices[i];
        if (d == sc->device) { scene_of[i] = sc; continue; }
        if (!sc->replicas[d]) {
            HIP_TRY(hipSetDevice(d));
            pt_scene* r = new pt_scene();
            r->host = sc->host;
            pt_status st = scene_to_device(r);
            if (st != PT_OK) { const std::string msg = g_error; pt_scene_destroy(r); hipSetDevice(sc->device); g_error = msg; return st; }
            sc->replicas[d] = r;
        }
        scene_of[i] = sc->replicas[d];
    }
    const size_t bytes = sizeof(float) * 4 * (size_t)rdp->width * rdp->height;
    std::vector<float*> d_film(n, nullptr);
    std::vector<hipStream_t> streams(n, nullptr);
    std::vector<pt_status> status(n, PT_OK);
    std::vector<std::string> errors(n);
    std::vector<pt_profile> profiles(n);
    std::vector<ncclComm_t> comms(n, nullptr);
    auto cleanup = [&] {
        for (int i = 0; i < n; ++i) {
            hipSetDevice(devices[i]);
            if (comms[i]) rccl().comm_destroy(comms[i]);
            if (streams[i]) hipStreamDestroy(streams[i]);
            if (d_film[i]) hipFree(d_film[i]);
        }
        hipSetDevice(sc->device);
    };
    if (use_rccl) {
        if (!rccl().ok()) return fail(PT_ERR_DEVICE, "librccl.so could not be loaded: pt_render_multi needs RCCL for more than one device");
        ncclResult_t rc = rccl().comm_init_all(comms.data(), n, devices.data());
        if (rc != ncclSuccess) { cleanup(); return fail(PT_ERR_DEVICE, std::string("ncclCommInitAll: ") + rccl().error_string(rc)); }
    }
    const auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](int i) {
        auto bad = [&](pt_status st, const std::string& msg) { status[i] = st; errors[i] = msg; };
        if (hipSetDevice(devices[i]) != hipSuccess) return bad(PT_ERR_DEVICE, "hipSetDevice failed");
        if (hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking) != hipSuccess) return bad(PT_ERR_DEVICE, "hipStreamCreate failed");
        if (hipMalloc(&d_film[i], bytes) != hipSuccess) return bad(PT_ERR_OUT_OF_MEMORY, "hipMalloc of the device film failed");
        pt_render_desc rd = *rdp;
        if (n > 1) { rd.shard_index = (uint32_t)i; rd.shard_count = (uint32_t)n; }
        pt_status st = render_impl(scene_of[i], &rd, d_film[i], streams[i], &profiles[i]);
        if (st != PT_OK) bad(st, g_error);   // (g_error is thread-local: carried back to the caller below)
    };
    {
        std::vector<std::thread> pool;
        for (int i = 1; i < n; ++i) pool.emplace_back(worker, i);
        worker(0);
        for (auto& t : pool) t.join();
    }
    for (int i = 0; i < n; ++i) if (status[i] != PT_OK) { const pt_status st = status[i]; const std::string msg = "device " + std::to_string(devices[i]) + ": " + errors[i]; cleanup(); return fail(st, msg); }
    if (use_rccl) {
        // the only exchange step of the path: every device's film (zero outside its own tiles) summed into the first device's
        ncclResult_t rc = rccl().group_start();
        for (int i = 0; i < n && rc == ncclSuccess; ++i) {
            hipSetDevice(devices[i]);
            rc = rccl().reduce(d_film[i], d_film[i], (size_t)4 * rdp->width * rdp->height, ncclFloat, ncclSum, 0, comms[i], streams[i]);
        }
        ncclResult_t rc2 = rccl().group_end();
        if (rc == ncclSuccess) rc = rc2;
        if (rc != ncclSuccess) { cleanup(); return fail(PT_ERR_DEVICE, std::string("ncclReduce: ") + rccl().error_string(rc)); }
        for (int i = 0; i < n; ++i) { hipSetDevice(devices[i]); if (hipStreamSynchronize(streams[i]) != hipSuccess) { cleanup(); return fail(PT_ERR_DEVICE, "stream synchronisation after the film reduce failed"); } }
    }
    const auto t1 = std::chrono::steady_clock::now();
    hipSetDevice(devices[0]);
    hipError_t e = hipMemcpy(film, d_film[0], bytes, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fail(PT_ERR_DEVICE, hipGetErrorString(e));
    if (profile) {
        memset(profile, 0, sizeof(*profile));
        for (const pt_profile& p : profiles) {
            profile->bounce_rays += p.bounce_rays; profile->shadow_rays += p.shadow_rays; profile->light_rays += p.light_rays;
            profile->camera_rays += p.camera_rays; profile->env_hits += p.env_hits;
            for (int k = 0; k < 8; ++k) { profile->kernel_seconds[k] += p.kernel_seconds[k]; profile->kernel_launches[k] += p.kernel_launches[k]; profile->stage_items[k] += p.stage_items[k]; }
        }
        profile->seconds = std::chrono::duration<double>(t1 - t0).count();
    }
    return PT_OK;
}

pt_status pt_intersect(pt_scene* sc, size_t n, const float* origins, const float* directions, pt_hit* hits) {
    if (!sc || !origins || !directions || !hits) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return PT_OK;
    HIP_TRY(hipSetDevice(sc->device));
    DevBuf dor, dd, dh;
    HIP_TRY(dor.alloc(12 * n)); HIP_TRY(dd.alloc(12 * n)); HIP_TRY(dh.alloc(sizeof(pt_hit) * n));
    HIP_TRY(hipMemcpy(dor.p, origins, 12 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dd.p, directions, 12 * n, hipMemcpyHostToDevice));
    int grid = sc->num_cus * 4;
    const uint32_t lds_bytes = sc->lds_mode == PT_LDS_ALL ? sc->blob_words * 4u : (sc->lds_mode == PT_LDS_CORE ? sc->host.blob[PT_HDR_CORE_WORDS] * 4u : 0u);
    launch_probe_intersect(LaunchCfg{grid, lds_bytes, (hipStream_t)0, sc->lds_mode}, SceneArgs{sc->d_blob, sc->blob_words, sc->d_tex}, (uint32_t)n, dor.as<float>(), dd.as<float>(), dh.as<pt_hit>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(hits, dh.p, sizeof(pt_hit) * n, hipMemcpyDeviceToHost));
    return PT_OK;
}

pt_status pt_bsdf_sample(pt_scene* sc, uint32_t material, size_t n, const float* lambda, const float* wi, const float* s2, float* f, float* wo, float* pdf) {
    if (!sc || material >= sc->host.material_count) return fail(PT_ERR_INVALID_ARGUMENT, "bad material");
    return probe_material(sc, 0, sc->host.blob[PT_HDR_MATERIAL_OFF] + material * PT_MAT_WORDS, n, lambda, wi, 3, s2, 2, f, wo, pdf);
}
pt_status pt_bsdf_eval(pt_scene* sc, uint32_t material, size_t n, const float* lambda, const float* wi, const float* wo, float* f, float* pdf) {
    if (!sc || material >= sc->host.material_count) return fail(PT_ERR_INVALID_ARGUMENT, "bad material");
    return probe_material(sc, 1, sc->host.blob[PT_HDR_MATERIAL_OFF] + material * PT_MAT_WORDS, n, lambda, wi, 3, wo, 3, f, nullptr, pdf);
}
pt_status pt_emission(pt_scene* sc, uint32_t material, size_t n, const float* lambda, const float* wi, float* emission) {
    if (!sc || material >= sc->host.material_count) return fail(PT_ERR_INVALID_ARGUMENT, "bad material");
    return probe_material(sc, 2, sc->host.blob[PT_HDR_MATERIAL_OFF] + material * PT_MAT_WORDS, n, lambda, wi, 3, nullptr, 0, emission, nullptr, nullptr);
}
pt_status pt_curve_eval(pt_scene* sc, uint32_t curve, size_t n, const float* lambda, float* value) {
    if (!sc || curve >= sc->host.curve_count) return fail(PT_ERR_INVALID_ARGUMENT, "bad curve");
    return probe_material(sc, 3, sc->host.curve_offsets[curve], n, lambda, nullptr, 0, nullptr, 0, value, nullptr, nullptr);
}

// Not part of pt_api.h: numeric-contract probe used by the GPU parity tests (device arithmetic vs x86).
pt_status pt_debug_numerics(int which, size_t n, const float* x, const float* y, float* out) {
    pt_status st = ensure_device();
    if (st != PT_OK) return st;
    DevBuf dx, dy, dout;
    HIP_TRY(dx.alloc(4 * n)); HIP_TRY(dy.alloc(4 * n)); HIP_TRY(dout.alloc(4 * n));
    HIP_TRY(hipMemcpy(dx.p, x, 4 * n, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(dy.p, y, 4 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe_numerics, dim3(256), dim3(kBlock), 0, 0, which, (uint32_t)n, dx.as<float>(), dy.as<float>(), dout.as<float>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout.p, 4 * n, hipMemcpyDeviceToHost));
    return PT_OK;
}

// Not part of pt_api.h: size of the scene blob and whether kernels read it from LDS (reported by bench.py).
uint32_t pt_debug_scene_info(pt_scene* sc, int what) {
    switch (what) { case 0: return sc->blob_words * 4; case 1: return (uint32_t)sc->lds_mode; case 2: return sc->host.light_count; case 3: return (uint32_t)sc->num_cus;
                    case 4: return sc->host.blob[PT_HDR_SWEEP_OFF] != 0 && !(sc->host.blob[PT_HDR_FLAGS] & PT_FLAG_NO_SWEEP) ? 1u : 0u;
                    case 7: return sc->host.blob[PT_HDR_CORE_WORDS] * 4; default: return 0; }
}

}  // extern "C"
