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
// the first stage of a camera sample as k_generate runs it (stage_generate), for chosen (pixel, sample) pairs: pt_camera_samples
__global__ void __launch_bounds__(kBlock) k_probe_camera(RenderParams rp, uint32_t n, const uint32_t* __restrict__ pixel, const uint32_t* __restrict__ sample,
                                                        float* __restrict__ o, float* __restrict__ d, float* __restrict__ lambda) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const PathVertexT<1> p = stage_generate<1>(rp, sample[i], pixel[i]);
        o[3 * i] = p.o.x; o[3 * i + 1] = p.o.y; o[3 * i + 2] = p.o.z;
        d[3 * i] = p.d.x; d[3 * i + 1] = p.d.y; d[3 * i + 2] = p.d.z;
        lambda[i] = p.lambda;
    }
}
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
            case 11: case 12: case 13: { float c[3]; ptd::xyz_bar(x[i], &c[0], &c[1], &c[2]); r = c[which - 11]; break; }   // x = angstrom
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
    uint32_t capacity = 0, light_samples = 0, nl = 0, park_block = 0;
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

// What pt_render_multi sets up for a set of devices and a film size, kept on the scene: a frame-by-frame caller pays for streams, device
// films and the RCCL communicator once (round-2 advice: they were made and destroyed on every call, outside the timed window).
struct MultiSetup {
    std::vector<int> devices;         // physical devices of the mask
    uint32_t virt = 1;                // virtual devices per physical one (pt_tuning::multi_virtual)
    bool rccl = false;
    size_t film_bytes = 0;
    std::vector<float*> films;        // per virtual device, on its physical device
    std::vector<hipStream_t> streams; // per virtual device
    std::vector<ncclComm_t> comms;    // per physical device (rccl only)
    bool valid = false;
};

struct pt_scene {
    pth::HostScene host;
    pt_tuning tuning;
    MultiSetup multi;
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
    std::vector<pt_scene*> replicas; // pt_render_multi: this scene on the other (virtual) devices, by device index x virtual index (nullptr = not made yet / this one)
};

namespace {
// whether any instance of the flattened scene is a mesh (a scene without one never parks at a mesh: its top-level walks are the ones worth leaving early)
bool scene_has_mesh(const std::vector<uint32_t>& blob) {
    const uint32_t off = blob[PT_HDR_INSTANCE_OFF], n = blob[PT_HDR_INSTANCE_COUNT];
    for (uint32_t i = 0; i < n; ++i) if (blob[off + i * PT_INST_WORDS + PT_INST_KIND] == (uint32_t)PT_SHAPE_MESH) return true;
    return false;
}

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

pt_status ensure_buffers(pt_scene* sc, uint32_t capacity, uint32_t light_samples, size_t n_pixels, int grid, uint32_t nl, uint32_t park_block) {
    DeviceBuffers& b = sc->buf;
    uint32_t total = segment_capacity(capacity, grid) * (uint32_t)grid;
    if (b.capacity < total || b.light_samples < light_samples || b.grid != grid || b.nl < nl || b.park_block < park_block) {
        hipFree(b.paths_a); hipFree(b.paths_b); hipFree(b.hits); hipFree(b.shadow); hipFree(b.energy); hipFree(b.counts); hipFree(b.block_stats); hipFree(b.park);
        b.paths_a = b.paths_b = b.hits = b.shadow = b.counts = b.park = nullptr; b.energy = nullptr; b.block_stats = nullptr; b.capacity = 0;
        uint32_t ls = light_samples > b.light_samples ? light_samples : b.light_samples;
        uint32_t nlmax = nl > b.nl ? nl : b.nl;
        size_t path_fields = nlmax == 4 ? Layout<4>::path_fields : Layout<1>::path_fields;
        size_t sh_fields = nlmax == 4 ? Layout<4>::shadow_queue_fields(ls ? ls : 1) : Layout<1>::shadow_queue_fields(ls ? ls : 1);
        HIP_TRY(hipMalloc(&b.paths_a, sizeof(uint32_t) * path_fields * total));
        HIP_TRY(hipMalloc(&b.paths_b, sizeof(uint32_t) * path_fields * total));
        HIP_TRY(hipMalloc(&b.hits, sizeof(uint32_t) * (size_t)HS_FIELDS * total));
        HIP_TRY(hipMalloc(&b.shadow, sizeof(uint32_t) * sh_fields * total));
        HIP_TRY(hipMalloc(&b.energy, sizeof(float) * (size_t)(nlmax + 1u) * total));   // (+ the plane of wavelength samples, PT_STORED_WAVELENGTH)
        b.nl = nlmax;
        HIP_TRY(hipMalloc(&b.counts, sizeof(uint32_t) * 4 * (size_t)grid));   // (live paths x 2, light-sample items, live light-sample items)
        HIP_TRY(hipMalloc(&b.block_stats, sizeof(unsigned long long) * BS_FIELDS * (size_t)grid));
        // (the parked kernels' scratch: scenes whose sweep table holds walked meshes, and every scene without a sweep table — with or without a mesh: the parked
        // kernels over the top-level tree list their live rays and run at five / four waves per SIMD, and beat the per-lane walk kernels even where no ray ever
        // parks: test_bokeh.toml + a floor, k_extend 1265 -> 693 us, k_shadow 5948 -> 2887 us, profiles/r5_experiments.md section 1)
        const bool no_table = sc->host.blob[PT_HDR_SWEEP_OFF] == 0 || (sc->host.blob[PT_HDR_FLAGS] & PT_FLAG_NO_SWEEP);
        if ((sc->host.blob[PT_HDR_FLAGS] & PT_FLAG_SWEEP_WALKS) || no_table)
            HIP_TRY(hipMalloc(&b.park, sizeof(uint32_t) * kParkFields * (kParkCap / (kBlock / 64)) * (park_block / 64) * (size_t)grid));   // (128 entries per wave)
        b.park_block = park_block;
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
// the value a tuning field stands for (0 = "the default" in the struct)
uint32_t tuned(uint32_t value, uint32_t dflt) { return value ? value : dflt; }

pt_status render_impl(pt_scene* sc, const pt_render_desc* rdp, float* d_film, hipStream_t stream, pt_profile* profile) {
    if (!sc || !rdp || !d_film) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    pt_render_desc rd;
    std::string err;
    if (!pth::normalize_render_desc(*rdp, (uint32_t)sc->host.cameras.size(), &rd, &err)) return fail(PT_ERR_INVALID_ARGUMENT, err);
    HIP_TRY(hipSetDevice(sc->device));

    std::vector<uint32_t> pixels = pth::shard_pixels(rd.width, rd.height, rd.tile_width, rd.tile_height, rd.shard_index, rd.shard_count);
    const pt_tuning& tn = sc->tuning;
    uint32_t capacity = tuned(tn.batch_slots, 1u << 27);  // path slots per pass (128 Mi ~ 32 GB of queues of the 288 GB; tools/sweep.sh)
    if (capacity < 1024) capacity = 1024;
    uint64_t want = (uint64_t)pixels.size() * rd.sample_count;
    if (want < capacity) capacity = (uint32_t)(want ? want : 1);
    const uint32_t blocks_per_cu = tuned(tn.blocks_per_cu, 64);
    if (blocks_per_cu > 1024) return fail(PT_ERR_INVALID_ARGUMENT, "pt_tuning::blocks_per_cu (PT_AMD_BLOCKS_PER_CU) must be in 1..1024");
    const int grid = sc->num_cus * (int)blocks_per_cu;  // queue segments = workgroups per launch
    // a pass holds at least one whole phase of one pixel (pt_plan.cpp): the queues must too (NaiveRenderer settings: phase = spp)
    { const uint32_t phase = rd.sample_count < rd.phase_samples ? rd.sample_count : rd.phase_samples; if (capacity < phase) capacity = phase; }
    const bool hero = rd.hero_wavelengths == 4;
    if (hero && capacity > (1u << 26)) capacity = 1u << 26;  // 4-wavelength queues are ~1.5x wider: 64 Mi slots ~ 24 GB
    // (the medium-aware walk keeps its two extra path fields where the hero layout keeps the passengers' throughputs)
    // the parked kernels in workgroups of 512 / 1024 threads that stage the whole blob (pt_tuning::park_block): static form, one wavelength, a blob that fits
    // 0 = the measured default: the light-sample kernel in workgroups of 512 (C3: -5 %), the closest-hit kernel in its own 256 (at 512 it loses its fifth wave
    // per SIMD: +14 %); 512 / 1024 = both kernels; 256 = neither (profiles/r4_experiments.md section 1)
    const bool park_big_ok = !hero && sc->blob_words * 4u <= kParkBlobLimitBytes && sc->lds_mode == PT_LDS_CORE && !(tn.flags & PT_TUNE_NO_LDS);
    // (round 6: in a scene with a certified convex body — PT_FLAG_CONVEX, the gem of C3 — nine in ten of the light rays that used to walk the mesh no longer reach it, and the
    // light-sample kernel is better off in workgroups of 256 that stage the core alone, twice the workgroups per CU: C3 k_shadow_parked 2339 -> 2083 us, profiles/r6d_ab_park.txt)
    const bool few_walks = (sc->host.blob[PT_HDR_FLAGS] & PT_FLAG_CONVEX) != 0u;
    const uint32_t park_block = !park_big_ok ? (uint32_t)kBlock : tn.park_block == 0u ? (few_walks ? (uint32_t)kBlock : 512u) : tn.park_block;
    const uint32_t park_block_extend = !park_big_ok || tn.park_block == 0u ? (uint32_t)kBlock : tn.park_block;
    pt_status st = ensure_buffers(sc, capacity, rd.light_samples, pixels.size() ? pixels.size() : 1, grid, (hero || rd.medium_aware) ? 4u : 1u, park_block);
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
    const bool parked = sweep && walks && b.park != nullptr && !(tn.flags & PT_TUNE_NO_PARK);
    // no sweep table (more than 64 instances, PT_AMD_NO_SWEEP): the top-level tree per lane, every mesh parked (top_walk_run, pt_device.h)
    const bool parked_walk = !sweep && b.park != nullptr && !(tn.flags & PT_TUNE_NO_PARK);
    // PT_AMD_POOL=1: phase 3 of a pure sweep scene pooled per wave (sweep_run_pooled).  Bit-identical, but measured slower than the lane
    // loop on MI355X (C2: k_extend 3155 vs 2475 us, k_shadow 5421 vs 4677 us; DESIGN.md section 5 has the breakdown), so it is not the default.
#ifdef PT_EXPERIMENTS
#define PT_TUNE_EXPERIMENT_POOL (1u << 31)   /* a pt_tuning::flags bit of its own, unnamed in the public header (round-4 advisor: bit 3 means PT_TUNE_NO_LIVE_LIST in every build) */
    const bool pooled = sweep && !walks && mode == PT_LDS_ALL && lds_bytes + pool_lds_bytes() <= kLdsBlobLimitBytes && (tn.flags & PT_TUNE_EXPERIMENT_POOL) != 0;
#else
    const bool pooled = false;   // (the pooled kernels are not in the product: make EXTRA=-DPT_EXPERIMENTS builds them, PT_AMD_POOL=1 selects them there)
#endif
    // (walked meshes in line under PT_AMD_NO_PARK, and every partly staged or unstaged blob: the run-time choice of PT_FORM_ANY)
    const int trav_form = parked ? PT_FORM_PARKED : parked_walk ? PT_FORM_PARKED_WALK : (mode != PT_LDS_ALL || (sweep && walks)) ? PT_FORM_ANY : pooled ? PT_FORM_POOLED : sweep ? PT_FORM_SWEEP : PT_FORM_WALK;
    // The parked kernels take units of work from a counter, a few persistent workgroups per CU, when the whole blob is staged in LDS
    // (C3: k_extend 9175 -> 7880 us, k_shadow 8008 -> 7105, 487 -> 543 Msamples/s: park lists that live across units keep the drains
    // full).  With the mesh in HBM/L2 (C4) the static form wins, 1128 vs 1083 Msamples/s: a wave's parked rays then come from one
    // region of the film and walk the same part of the mesh.  PT_AMD_PARK_DYNAMIC=0 / 1 forces either.
    const bool park_dynamic = parked && (tn.park_dynamic < 0 ? mode == PT_LDS_ALL : tn.park_dynamic != 0);
    const bool park_big = parked && !park_dynamic && park_block != (uint32_t)kBlock && mode != PT_LDS_ALL;
    const int dyn_grid = sc->num_cus * (int)tuned(tn.park_blocks_per_cu, 4);
    LaunchCfg cfg{grid, lds_bytes, stream, mode};
    cfg.dyn_grid = dyn_grid < grid ? dyn_grid : grid;
    cfg.lacks = sc->lacks;
#ifdef PT_EXPERIMENTS
    cfg.live_lists = env_u32("PT_AMD_LIVE_LISTS", 0) != 0;   // (k_shadow_live: a measurement build's kernel, profiles/r4_experiments.md)
#endif
    if (park_big) { cfg.park_block = (int)park_block; cfg.park_block_extend = (int)park_block_extend; cfg.park_blob_bytes = sc->blob_words * 4u; }
    // The top-level walk is left early by a wave's last lanes (top_walk_run) where that walk is long and nothing else thins the wave out: a tree of more than 64
    // instances — the scenes that have no sweep table by themselves — without a mesh (test_bokeh.toml + a floor: k_shadow_parked 2900 -> 2320 us at 32, 2430 at 16 and
    // at 48).  With a mesh in the scene the lanes of a wave PARK at it, the wave thins out although its rays are not done, and evicting the rest only adds park
    // traffic: the same scene with the gem standing on the floor 1337 -> 1260 Msamples/s at 32 (1341 at 16); a small scene forced off its table (PT_AMD_NO_SWEEP:
    // thirteen instances in the gem scene, ten nodes per walk) lost 15 % (profiles/r5_experiments.md section 2).
    const uint32_t top_evict = tn.top_evict_below ? tn.top_evict_below
                             : (sc->host.blob[PT_HDR_INSTANCE_COUNT] > PT_SWEEP_MAX_BITS && !scene_has_mesh(sc->host.blob) ? kTopEvictBelow : 1u);
    // The grouped mesh sweep's group loop is left by a wave's last lanes (mesh_walk, GROUPS) where every parked ray of the closest-hit kernel stands in the SAME mesh — one
    // walked mesh in the table: a resumed wave is then always one the grouped sweep takes, and a ray that comes back with groups to do is never walked from the top.
    uint32_t group_evict = 1u;
    {
        const std::vector<uint32_t>& bl = sc->host.blob;
        uint32_t walked = 0;
        if (sweep) for (uint32_t j = 0; j < bl[PT_HDR_SWEEP_COUNT]; ++j) walked += (bl[bl[PT_HDR_SWEEP_OFF] + j * PT_SWEEP_INST_WORDS + 1] & PT_SWEEP_WALKED) ? 1u : 0u;
        if (parked && walked == 1u) group_evict = tn.group_evict_below ? tn.group_evict_below : kGroupEvictBelow;
    }
    cfg.walk_policy = (tn.walk_evict_below ? tn.walk_evict_below : kWalkEvictBelow) | (tn.walk_search_below ? tn.walk_search_below : kWalkSearchBelow) << 8
                    | ((tn.flags & PT_TUNE_NO_AXIS_SCAN) ? 0u : PT_WALK_SCAN_AXIS)
                    | (top_evict <= 1u ? 0u : top_evict << 24)   // (1 = never: 0 in the policy word)
                    | (group_evict <= 1u ? 0u : group_evict << 17);
    const SceneArgs sargs{sc->d_blob, sc->blob_words, sc->d_tex, marginal_lds_bytes(sc->host.blob.data(), lds_bytes)};
    // light samples can pick the environment only if env_sampling_probability > 0: otherwise k_shade is the form without that branch
    float env_prob; std::memcpy(&env_prob, &sc->host.blob[PT_HDR_ENV_PROB], sizeof env_prob);
    bool has_ggx = false;
    for (uint32_t i = 0; i < sc->host.blob[PT_HDR_MATERIAL_COUNT]; ++i) {
        const uint32_t kind = sc->host.blob[sc->host.blob[PT_HDR_MATERIAL_OFF] + i * PT_MAT_WORDS + PT_MAT_KIND];
        has_ggx = has_ggx || kind == PT_MATERIAL_GGX || kind == PT_MATERIAL_PASSTHROUGH;
    }
    // (a PassthroughFilter lives in the forms that hold the GGX code)
    // (a scene with a convex-body certificate takes at least the NO_ENV form: the lean and fused forms are compiled without the certificate code, pt_kern_shade.hip)
    const bool certs = !rd.medium_aware && (sc->host.blob[PT_HDR_FLAGS] & PT_FLAG_CONVEX) != 0u;
    const int shade_form = rd.medium_aware ? PT_SHADE_MEDIUM
                         : (env_prob != 0.0f || tn.shade_form == 2) ? PT_SHADE_FULL : (has_ggx || certs || tn.shade_form == 1) ? PT_SHADE_NO_ENV : PT_SHADE_LEAN;
    cfg.certs = certs;
    // k_shade that traces its own segments: exists for the pure sweep form of a fully staged, transform-free scene shaded by the lean form.
    // Measured (profiles/r3_experiments.md): C2 +4..5 % (4370 us against 2095 + 2490..2640 per bounce); with four wavelengths per path it
    // lost in round 3 (C5 996 against 1093 Msamples/s: the traversal then ran at the three waves per SIMD the wide vertex code left) and wins since round 4 (below).
#ifndef PT_FUSE_HERO
#define PT_FUSE_HERO 1   /* round 4: built without machine LICM the hero fused form needs 111 VGPRs — four waves per SIMD, not three — and wins: C5 1153 -> 1179 (3625 us against 1160 + 2578) */
#endif
    // (the plain light-sample kernel walks the list of live items; the parked forms list their live RAYS themselves, the measurement forms read every item)
    rp.live_list = ((trav_form == PT_FORM_SWEEP || trav_form == PT_FORM_WALK || trav_form == PT_FORM_ANY) && shade_form == PT_SHADE_LEAN && !(tn.flags & PT_TUNE_NO_LIVE_LIST)) ? 1u : 0u;
#ifdef PT_EXPERIMENTS
    // a measurement build runs the shipped kernel pair unless one of its own forms is asked for: those read every item of a segment
    if (cfg.live_lists || getenv("PT_AMD_EXP_SHADOW")) rp.live_list = 0u;
#endif
    rp.camera_record = (shade_form == PT_SHADE_LEAN || shade_form == PT_SHADE_FULL || shade_form == PT_SHADE_MEDIUM) ? 1u : 0u;   // (the forms whose bounce-0 launch rebuilds the camera vertex: k_shade, pt_kernels.h)
    cfg.fuse = !(tn.flags & PT_TUNE_NO_FUSE) && (!hero || PT_FUSE_HERO) && trav_form == PT_FORM_SWEEP && shade_form == PT_SHADE_LEAN && (cfg.lacks & PT_SCENE_NO_XF) != 0;
    const uint32_t bounce_limit = rd.only_direct ? 1u : rd.max_bounces;
    const bool timing = !(tn.flags & PT_TUNE_NO_STAGE_TIMING);
    double stage_ms[ST_COUNT] = {0, 0, 0, 0, 0};
    uint64_t stage_launches[ST_COUNT] = {0, 0, 0, 0, 0};
    // (a queue's tiles are laid out by the number of fields this render uses, pt_stages.h: the buffers are sized for the widest layout seen)
    const uint32_t path_fields = hero ? Layout<4>::path_fields : (rd.medium_aware ? (uint32_t)PS_FIELDS + 2u : Layout<1>::path_fields);
    const uint32_t item_fields = hero ? Layout<4>::shadow_queue_fields(rd.light_samples ? rd.light_samples : 1) : Layout<1>::shadow_queue_fields(rd.light_samples ? rd.light_samples : 1);
    Queue qa{b.paths_a, b.capacity, path_fields}, qb{b.paths_b, b.capacity, path_fields}, qh{b.hits, b.capacity, HS_FIELDS}, qs{b.shadow, b.capacity, item_fields};
    uint32_t* live[2] = {b.counts, b.counts + grid};  // per-segment live-path counts, ping-pong with the path queues
    uint32_t* nshadow = b.counts + 2 * grid;          // per-segment light-sample item counts; behind them (nshadow + grid) the counts of the live ones (Layout::shadow_live_field)

    auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipStreamSynchronize(stream));
    t0 = std::chrono::steady_clock::now();
    // HIP events around every launch, recorded on the launch stream and read back after the final sync, so the
    // per-stage device time is measured inside the timed region without stalling it.
    // (round 5: ONE event between two launches — the end of one is the start of the next on the stream — not two: the markers cost a short frame 5 % of its time,
    // G2 9.7 -> 9.2 ms per step without any; a launch's time now includes the gap in front of it, a few microseconds)
    std::vector<int> event_stage;
    auto event_at = [&](size_t k) -> bool {
        while (sc->events.size() <= k) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return false; sc->events.push_back(e); }
        return hipEventRecord(sc->events[k], stream) == hipSuccess;
    };
    bool events_ok = timing && event_at(0);
    auto timed = [&](int stage, auto&& fn) {
        fn();
        if (events_ok) { events_ok = event_at(event_stage.size() + 1); if (events_ok) event_stage.push_back(stage); }
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
        if (park_dynamic) {
            HIP_TRY(hipMemsetAsync(b.unit_counters, 0, sizeof(uint32_t) * kUnitCounters, stream));
            // (round-5 advisor) an event of its own behind the stream's non-kernel work, charged to no stage: k_generate's time begins here, not at the end of the previous pass
            if (events_ok) { events_ok = event_at(event_stage.size() + 1); if (events_ok) event_stage.push_back(-1); }
        }
        timed(ST_GENERATE, [&] {
            if (hero) hipLaunchKernelGGL(k_generate<4>, dim3(grid), dim3(kBlock), 0, stream, rp, d_px, qa, b.energy, n, seg_cap, live[0]);
            else hipLaunchKernelGGL(k_generate<1>, dim3(grid), dim3(kBlock), 0, stream, rp, d_px, qa, b.energy, n, seg_cap, live[0]);
        });
        for (uint32_t bounce = 0; bounce < bounce_limit; ++bounce) {
            Queue qin = (bounce & 1) ? qb : qa, qout = (bounce & 1) ? qa : qb;
            uint32_t *cin = live[bounce & 1], *cout = live[(bounce + 1) & 1];
            // kernel variant = staging mode (PT_LDS_*) x traversal form x wavelengths per path (pt_launch.h)
            if (park_dynamic) cfg.unit_counter = b.unit_counters + 2 * bounce;
            // (a marked path segment — it left the scene's one certified convex body outward — skips that instance: records the vertex kernel wrote, so from bounce 1 on; never
            // in the medium-aware walk, whose vertex code makes no marks)
            cfg.path_marks = (bounce > 0 && !rd.medium_aware && (sc->host.blob[PT_HDR_FLAGS] & PT_FLAG_CONVEX)) ? sc->host.blob[PT_HDR_CONVEX_INST] : 0u;
            if (!cfg.fuse) timed(ST_EXTEND, [&] { launch_extend(cfg, trav_form, sargs, qin, qh, seg_cap, cin, b.park); });
            if (park_dynamic) cfg.unit_counter = b.unit_counters + 2 * bounce + 1;
            timed(ST_SHADE, [&] { launch_shade(cfg, hero ? 4 : 1, shade_form, sargs, rp, bounce, d_px, qin, qh, qout, qs, b.energy, seg_cap, cin, cout, nshadow, b.block_stats); });
            if (rd.light_samples > 0)   // (shade_form FULL = the scene can produce environment rays)
                timed(ST_SHADOW, [&] { launch_shadow(cfg, trav_form, hero ? 4 : 1, shade_form == PT_SHADE_FULL || shade_form == PT_SHADE_MEDIUM, sargs, rd.light_samples, qs, b.energy, b.capacity, rp.live_list ? (seg_cap | kShadowListed) : seg_cap, nshadow, b.park, qh); });
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
        if (event_stage[k] >= 0 && hipEventElapsedTime(&ms, sc->events[k], sc->events[k + 1]) == hipSuccess) stage_ms[event_stage[k]] += ms;
    }
    if (profile) {
        memset(profile, 0, sizeof(*profile));
        std::vector<unsigned long long> bs((size_t)grid * BS_FIELDS);
        HIP_TRY(hipMemcpy(bs.data(), b.block_stats, sizeof(unsigned long long) * bs.size(), hipMemcpyDeviceToHost));
        unsigned long long c[BS_FIELDS] = {0, 0, 0, 0, 0, 0};
        for (int g = 0; g < grid; ++g) for (int k = 0; k < BS_FIELDS; ++k) c[k] += bs[(size_t)g * BS_FIELDS + k];
        profile->camera_rays = camera_rays;
        profile->bounce_rays = c[BS_VERTICES] + camera_rays;  // vertices.len() counts the camera vertex (utils.rs:375)
        profile->shadow_rays = c[BS_SHADOW_RAYS];
        profile->env_hits = c[BS_ENV_HITS];
        profile->seconds = std::chrono::duration<double>(t1 - t0).count();
        for (int i = 0; i < ST_COUNT; ++i) { profile->kernel_seconds[i] = stage_ms[i] * 1e-3; profile->kernel_launches[i] = stage_launches[i]; }
        profile->stage_items[ST_GENERATE] = camera_rays; profile->stage_items[ST_EXTEND] = c[BS_SEGMENTS]; profile->stage_items[ST_SHADE] = c[BS_SEGMENTS];
        profile->stage_items[ST_SHADOW] = c[BS_ITEMS]; profile->stage_items[ST_ACCUMULATE] = accumulated_pixels;
        profile->stage_items[6] = (uint64_t)cfg.park_block;   // threads per workgroup of the parked kernels when they ran in their big-workgroup form (pt_tuning::park_block), else 0
        profile->stage_items[7] = rp.camera_record;     // 1: k_generate wrote the camera vertex' lean record (7 of 16 words) and the first bounce's vertex kernel rebuilt the rest
        profile->stage_items[5] = c[BS_MEDIUM_DROPS];   // the medium-aware walk tracks four nested mediums: what a fifth level lost (0 = the walk is the reference's)
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
    const pt_tuning& tn = sc->tuning;
    if (tn.flags & PT_TUNE_EXACT_SLAB) sc->host.blob[PT_HDR_FLAGS] |= PT_FLAG_EXACT_SLAB;
    if (tn.flags & PT_TUNE_NO_CULL) sc->host.blob[PT_HDR_FLAGS] |= PT_FLAG_NO_CULL;
    if (tn.flags & PT_TUNE_NO_SWEEP) sc->host.blob[PT_HDR_FLAGS] |= PT_FLAG_NO_SWEEP;
    if (tn.flags & PT_TUNE_NO_MESH_SWEEP) sc->host.blob[PT_HDR_FLAGS] |= PT_FLAG_NO_MESH_SWEEP;
    if (tn.flags & PT_TUNE_NO_KNOWN_LIGHT) sc->host.blob[PT_HDR_FLAGS] |= PT_FLAG_NO_KNOWN_LIGHT;
    if (tn.flags & PT_TUNE_NO_ONE_LIGHT) sc->host.blob[PT_HDR_FLAGS] |= PT_FLAG_NO_ONE_LIGHT;
    if (tn.flags & PT_TUNE_NO_MESH_SHORTCUTS) {   // (the mesh records' inner ball and slab table: mesh_surely_blocks / mesh_surely_missed claim nothing without them)
        std::vector<uint32_t>& bl = sc->host.blob;
        for (uint32_t i = 0; i < bl[PT_HDR_INSTANCE_COUNT]; ++i) {
            const uint32_t inst = bl[PT_HDR_INSTANCE_OFF] + i * PT_INST_WORDS;
            if (bl[inst + PT_INST_KIND] != (uint32_t)PT_SHAPE_MESH) continue;
            bl[bl[inst + PT_INST_MESH] + PT_MESH_INNER_R] = 0u; bl[bl[inst + PT_INST_MESH] + PT_MESH_DOP_OFF] = 0u;
        }
    }
    if (tn.flags & PT_TUNE_NO_CONVEX) sc->host.blob[PT_HDR_FLAGS] &= ~PT_FLAG_CONVEX;   // (the vertex code looks at an instance's certificate only under this flag, and only it makes marks)
    if (sc->host.blob[PT_HDR_LIGHT_COUNT] > tuned(tn.light_prepass_max, kLightPrepassMax)) sc->host.blob[PT_HDR_FLAGS] |= PT_FLAG_NO_LIGHT_PREPASS;
    sc->blob_words = (uint32_t)sc->host.blob.size();
    {   // no instance carries a transform (the Cornell box): the forms without the matrix paths (PT_AMD_GENERAL_FORMS=1 keeps the general ones)
        bool any_xf = false;
        const std::vector<uint32_t>& bl = sc->host.blob;
        for (uint32_t i = 0; i < bl[PT_HDR_INSTANCE_COUNT]; ++i) any_xf = any_xf || (bl[bl[PT_HDR_INSTANCE_OFF] + i * PT_INST_WORDS + PT_INST_FLAGS] & 1u) != 0u;
        // "no lights" = no hit can carry a Light tag: the light list is empty AND no mesh instance overrides its material with a light
        // (such a mesh is not in the light list, world/mod.rs:45-54, but its hits emit and take no light samples: PT_FLAG_NO_SHADOW_BOUND)
        const bool no_light_hits = bl[PT_HDR_LIGHT_COUNT] == 0u && !(bl[PT_HDR_FLAGS] & PT_FLAG_NO_SHADOW_BOUND);
        sc->lacks = (tn.flags & PT_TUNE_GENERAL_FORMS) ? 0u : ((any_xf ? 0u : PT_SCENE_NO_XF) | (no_light_hits ? PT_SCENE_NO_LIGHTS : 0u));
    }
    const bool no_lds = (tn.flags & PT_TUNE_NO_LDS) != 0;
    const uint32_t all_limit = tuned(tn.lds_all_limit, kLdsAllLimitBytes);   // (experiments: the largest blob staged whole)
    sc->lds_mode = no_lds ? PT_LDS_NONE : (sc->blob_words * 4 <= (all_limit < kLdsBlobLimitBytes ? all_limit : kLdsBlobLimitBytes) ? PT_LDS_ALL
                 : (sc->host.blob[PT_HDR_CORE_WORDS] * 4 <= kLdsBlobLimitBytes && !(tn.flags & PT_TUNE_NO_CORE_LDS) ? PT_LDS_CORE : PT_LDS_NONE));
    e = hipMalloc(&sc->d_blob, sizeof(uint32_t) * sc->host.blob.size());
    if (e == hipSuccess) e = hipMalloc(&sc->d_tex, sizeof(float) * (sc->host.tex.size() + 4));
    if (e == hipSuccess) e = hipMemcpy(sc->d_blob, sc->host.blob.data(), sizeof(uint32_t) * sc->host.blob.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(sc->d_tex, sc->host.tex.data(), sizeof(float) * sc->host.tex.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? PT_ERR_OUT_OF_MEMORY : PT_ERR_DEVICE, hipGetErrorString(e));
    if (sc->lds_mode != PT_LDS_NONE) {
        e = allow_lds_extend(kParkBlobLimitBytes);
        if (e == hipSuccess) e = allow_lds_shade(kLdsBlobLimitBytes > PT_SHADE_LDS_BUDGET ? kLdsBlobLimitBytes : PT_SHADE_LDS_BUDGET);   // (the FULL form's marginal tables ride behind the blob only within PT_SHADE_LDS_BUDGET)
        // (the parked light-sample kernels keep their waves' lists of live rays behind the blob: up to 16 waves x (64 L + 64) words)
        if (e == hipSuccess) e = allow_lds_shadow(kParkBlobLimitBytes + 16u * (64u * PT_MAX_LIGHT_SAMPLES + 64u) * 4u);
        if (e != hipSuccess) return fail(PT_ERR_DEVICE, std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(e));
    }
    return PT_OK;
}

void pt_tuning_default(pt_tuning* t) {
    if (!t) return;
    memset(t, 0, sizeof(*t));
    const struct { const char* name; uint32_t bit; } flags[] = {
        {"PT_AMD_NO_LDS", PT_TUNE_NO_LDS}, {"PT_AMD_NO_CORE_LDS", PT_TUNE_NO_CORE_LDS}, {"PT_AMD_NO_PARK", PT_TUNE_NO_PARK},
#ifdef PT_EXPERIMENTS
        {"PT_AMD_POOL", PT_TUNE_EXPERIMENT_POOL},
#endif

        {"PT_AMD_EXACT_SLAB", PT_TUNE_EXACT_SLAB}, {"PT_AMD_NO_CULL", PT_TUNE_NO_CULL}, {"PT_AMD_NO_SWEEP", PT_TUNE_NO_SWEEP}, {"PT_AMD_NO_MESH_SWEEP", PT_TUNE_NO_MESH_SWEEP},
        {"PT_AMD_NO_KNOWN_LIGHT", PT_TUNE_NO_KNOWN_LIGHT}, {"PT_AMD_GENERAL_FORMS", PT_TUNE_GENERAL_FORMS}, {"PT_AMD_NO_FUSE", PT_TUNE_NO_FUSE}, {"PT_AMD_MULTI_RCCL", PT_TUNE_MULTI_RCCL}, {"PT_AMD_NO_AXIS_SCAN", PT_TUNE_NO_AXIS_SCAN}, {"PT_AMD_NO_ONE_LIGHT", PT_TUNE_NO_ONE_LIGHT}, {"PT_AMD_NO_CONVEX", PT_TUNE_NO_CONVEX}, {"PT_AMD_NO_MESH_SHORTCUTS", PT_TUNE_NO_MESH_SHORTCUTS},
        {"PT_AMD_NO_LIVE_LIST", PT_TUNE_NO_LIVE_LIST},
    };
    for (const auto& f : flags) if (env_u32(f.name, 0)) t->flags |= f.bit;
    if (env_u32("PT_AMD_STAGE_TIMING", 1) == 0) t->flags |= PT_TUNE_NO_STAGE_TIMING;
    t->batch_slots = env_u32("PT_AMD_BATCH", 0);
    t->blocks_per_cu = env_u32("PT_AMD_BLOCKS_PER_CU", 0);
    t->park_blocks_per_cu = env_u32("PT_AMD_PARK_BLOCKS_PER_CU", 0);
    t->park_dynamic = getenv("PT_AMD_PARK_DYNAMIC") && *getenv("PT_AMD_PARK_DYNAMIC") ? (env_u32("PT_AMD_PARK_DYNAMIC", 0) != 0 ? 1 : 0) : -1;
    t->shade_form = env_u32("PT_AMD_SHADE_FORM", 0);
    t->lds_all_limit = env_u32("PT_AMD_LDS_ALL_LIMIT", 0);
    t->multi_virtual = env_u32("PT_AMD_MULTI_VIRTUAL", 0);
    t->walk_evict_below = env_u32("PT_AMD_WALK_EVICT_BELOW", 0);
    t->walk_search_below = env_u32("PT_AMD_WALK_SEARCH_BELOW", 0);
    t->park_block = env_u32("PT_AMD_PARK_BLOCK", 0);
    t->light_prepass_max = env_u32("PT_AMD_LIGHT_PREPASS_MAX", 0);
    t->top_evict_below = env_u32("PT_AMD_TOP_EVICT_BELOW", 0);
    t->group_evict_below = env_u32("PT_AMD_GROUP_EVICT_BELOW", 0);
}

pt_status pt_scene_create(const pt_scene_desc* desc, pt_scene** out) {
    pt_tuning t;
    pt_tuning_default(&t);   // the only place the PT_AMD_* environment is read
    return pt_scene_create_tuned(desc, &t, out);
}

pt_status pt_scene_create_tuned(const pt_scene_desc* desc, const pt_tuning* tuning, pt_scene** out) {
    if (!desc || !out || !tuning) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    for (uint32_t r : tuning->reserved) if (r != 0) return fail(PT_ERR_INVALID_ARGUMENT, "pt_tuning::reserved must be 0");
    if (tuning->park_block != 0 && tuning->park_block != 256 && tuning->park_block != 512 && tuning->park_block != 1024)
        return fail(PT_ERR_INVALID_ARGUMENT, "pt_tuning::park_block (PT_AMD_PARK_BLOCK) must be 0, 256, 512 or 1024");
    if (tuning->shade_form > 2 || tuning->park_dynamic < -1 || tuning->park_dynamic > 1 || tuning->multi_virtual > 64 || tuning->walk_evict_below > 64 || tuning->walk_search_below > 64 || tuning->top_evict_below > 64 || tuning->group_evict_below > 64)
        return fail(PT_ERR_INVALID_ARGUMENT, "pt_tuning: shade_form in 0..2, park_dynamic in -1..1, multi_virtual, walk_evict_below, walk_search_below, top_evict_below, group_evict_below <= 64");
    // the tiled queue index (pt_stages.h qtile) multiplies in 32 bits: capacity <= 2^30; the grids are num_cus * blocks in an int
    if (tuning->batch_slots > (1u << 30) || tuning->blocks_per_cu > 1024u || tuning->park_blocks_per_cu > 1024u)
        return fail(PT_ERR_INVALID_ARGUMENT, "pt_tuning: batch_slots (PT_AMD_BATCH) <= 2^30, blocks_per_cu and park_blocks_per_cu <= 1024");
    pt_status st = ensure_device();
    if (st != PT_OK) return st;
    pt_scene* sc = new pt_scene();
    sc->tuning = *tuning;
    std::string err;
    if (!pth::build_host_scene(*desc, &sc->host, &err)) { delete sc; return fail(PT_ERR_INVALID_ARGUMENT, err); }
    st = scene_to_device(sc);
    if (st != PT_OK) { const std::string msg = g_error; pt_scene_destroy(sc); g_error = msg; return st; }
    *out = sc;
    return PT_OK;
}

static void multi_release(MultiSetup& m);
void pt_scene_destroy(pt_scene* sc) {
    if (!sc) return;
    for (pt_scene* r : sc->replicas) if (r) pt_scene_destroy(r);
    sc->replicas.clear();
    if (sc->multi.valid) multi_release(sc->multi);
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

extern "C++" {   // (C++ helpers inside the extern "C" region of the API functions)
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
}  // extern "C++"

// dst += src over n float4 pixels: the films of the virtual devices that share one physical device (pt_tuning::multi_virtual)
__global__ void __launch_bounds__(kBlock) k_film_add(float4* __restrict__ dst, const float4* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 a = dst[i], b = src[i];
        dst[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

static void multi_release(MultiSetup& m) {
    for (size_t v = 0; v < m.films.size(); ++v) {
        if (m.devices.empty()) break;
        hipSetDevice(m.devices[v / m.virt]);
        if (v < m.streams.size() && m.streams[v]) hipStreamDestroy(m.streams[v]);
        if (m.films[v]) hipFree(m.films[v]);
    }
    for (size_t p = 0; p < m.comms.size(); ++p) if (m.comms[p]) { hipSetDevice(m.devices[p]); rccl().comm_destroy(m.comms[p]); }
    m = MultiSetup();
}

// Streams, device films and (for more than one physical device) the RCCL communicator of a device set, made once per scene and film size.
static pt_status multi_setup(pt_scene* sc, const std::vector<int>& devices, uint32_t virt, bool use_rccl, size_t film_bytes) {
    MultiSetup& m = sc->multi;
    if (m.valid && m.devices == devices && m.virt == virt && m.rccl == use_rccl && m.film_bytes >= film_bytes) return PT_OK;
    multi_release(m);
    m.devices = devices; m.virt = virt; m.rccl = use_rccl; m.film_bytes = film_bytes;
    const size_t nv = devices.size() * virt;
    m.films.assign(nv, nullptr); m.streams.assign(nv, nullptr);
    for (size_t v = 0; v < nv; ++v) {
        hipError_t e = hipSetDevice(devices[v / virt]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m.streams[v], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc(&m.films[v], film_bytes);
        if (e != hipSuccess) { multi_release(m); return fail(e == hipErrorOutOfMemory ? PT_ERR_OUT_OF_MEMORY : PT_ERR_DEVICE, std::string("pt_render_multi set-up: ") + hipGetErrorString(e)); }
    }
    if (use_rccl) {
        if (!rccl().ok()) { multi_release(m); return fail(PT_ERR_DEVICE, "librccl.so could not be loaded: pt_render_multi needs RCCL for more than one device"); }
        m.comms.assign(devices.size(), nullptr);
        ncclResult_t rc = rccl().comm_init_all(m.comms.data(), (int)devices.size(), devices.data());
        if (rc != ncclSuccess) { multi_release(m); return fail(PT_ERR_DEVICE, std::string("ncclCommInitAll: ") + rccl().error_string(rc)); }
    }
    m.valid = true;
    return PT_OK;
}

pt_status pt_render_multi(pt_scene* sc, const pt_render_desc* rdp, uint64_t device_mask, float* film, pt_profile* profile) {
    if (!sc || !rdp || !film) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    if (rdp->width == 0 || rdp->height == 0) return fail(PT_ERR_INVALID_ARGUMENT, "width and height must be positive");
    if (rdp->shard_count > 1) return fail(PT_ERR_INVALID_ARGUMENT, "pt_render_multi deals the tiles itself: shard_count must be 0");
    const uint32_t visible = pt_device_count();
    if (visible == 0) return fail(PT_ERR_NO_DEVICE, "no HIP device available: the product path has no CPU fallback");
    std::vector<int> devices;
    for (uint32_t d = 0; d < visible && d < 64; ++d) if (device_mask == 0 || ((device_mask >> d) & 1ull)) devices.push_back((int)d);
    if (devices.empty()) return fail(PT_ERR_INVALID_ARGUMENT, "device_mask names no visible HIP device");
    const int np = (int)devices.size();                                         // physical devices
    const uint32_t virt = sc->tuning.multi_virtual > 1 ? sc->tuning.multi_virtual : 1u;   // virtual devices per physical one (test mode)
    const int n = np * (int)virt;                                               // shards = host threads = streams = replicas
    // the RCCL reduce for more than one physical device; PT_TUNE_MULTI_RCCL takes it even for one (the call path of a node, on a single GPU)
    const bool use_rccl = np > 1 || (sc->tuning.flags & PT_TUNE_MULTI_RCCL) != 0;
    if (!use_rccl && n == 1 && devices[0] == sc->device) return pt_render(sc, rdp, film, profile);

    // whatever happens below, the caller gets its current device back
    struct DeviceGuard { int d = 0; bool ok = false; DeviceGuard() { ok = hipGetDevice(&d) == hipSuccess; } ~DeviceGuard() { if (ok) hipSetDevice(d); } } guard;
    const auto t_entry = std::chrono::steady_clock::now();

    // one replica per (virtual) device — this scene itself for the first one on its own device — made on first use and kept
    if (sc->replicas.size() < (size_t)visible * virt) sc->replicas.resize((size_t)visible * virt, nullptr);
    std::vector<pt_scene*> scene_of(n, nullptr);
    for (int v = 0; v < n; ++v) {
        const int d = devices[v / virt];
        const size_t slot = (size_t)d * virt + (size_t)v % virt;
        if (d == sc->device && v % (int)virt == 0) { scene_of[v] = sc; continue; }
        if (!sc->replicas[slot]) {
            HIP_TRY(hipSetDevice(d));
            pt_scene* r = new pt_scene();
            r->host = sc->host;
            r->tuning = sc->tuning;
            pt_status st = scene_to_device(r);
            if (st != PT_OK) { const std::string msg = g_error; pt_scene_destroy(r); g_error = msg; return st; }
            sc->replicas[slot] = r;
        }
        scene_of[v] = sc->replicas[slot];
    }
    const size_t bytes = sizeof(float) * 4 * (size_t)rdp->width * rdp->height;
    pt_status st = multi_setup(sc, devices, virt, use_rccl, bytes);
    if (st != PT_OK) return st;
    MultiSetup& m = sc->multi;
    std::vector<pt_status> status(n, PT_OK);
    std::vector<std::string> errors(n);
    std::vector<pt_profile> profiles(n);
    const auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](int v) {
        auto bad = [&](pt_status s2, const std::string& msg) { status[v] = s2; errors[v] = msg; };
        if (hipSetDevice(devices[v / virt]) != hipSuccess) return bad(PT_ERR_DEVICE, "hipSetDevice failed");
        pt_render_desc rd = *rdp;
        if (n > 1) { rd.shard_index = (uint32_t)v; rd.shard_count = (uint32_t)n; }
        pt_status s2 = render_impl(scene_of[v], &rd, m.films[v], m.streams[v], &profiles[v]);
        if (s2 != PT_OK) bad(s2, g_error);   // (g_error is thread-local: carried back to the caller below)
    };
    {
        std::vector<std::thread> pool;
        for (int v = 1; v < n; ++v) pool.emplace_back(worker, v);
        worker(0);
        for (auto& t : pool) t.join();
    }
    for (int v = 0; v < n; ++v) if (status[v] != PT_OK) return fail(status[v], "device " + std::to_string(devices[v / virt]) + (virt > 1 ? "." + std::to_string(v % virt) : "") + ": " + errors[v]);
    // the only exchange step of the path.  Every film is zero outside its own tiles, so the sums are gathers and keep every bit.
    const auto t_reduce = std::chrono::steady_clock::now();
    const size_t pixels = (size_t)rdp->width * rdp->height;
    for (int p = 0; p < np && virt > 1; ++p) {   // (1) the virtual devices of one physical device: added on that device (render_impl has synchronised their streams)
        HIP_TRY(hipSetDevice(devices[p]));
        for (uint32_t j = 1; j < virt; ++j)
            hipLaunchKernelGGL(k_film_add, dim3(1024), dim3(kBlock), 0, m.streams[(size_t)p * virt], reinterpret_cast<float4*>(m.films[(size_t)p * virt]), reinterpret_cast<const float4*>(m.films[(size_t)p * virt + j]), pixels);
        HIP_TRY(hipGetLastError());
        if (!use_rccl) HIP_TRY(hipStreamSynchronize(m.streams[(size_t)p * virt]));
    }
    if (use_rccl) {   // (2) the physical devices: one ncclReduce(sum) into the first device of the mask, over xGMI
        ncclResult_t rc = rccl().group_start();
        for (int p = 0; p < np && rc == ncclSuccess; ++p) {
            hipSetDevice(devices[p]);
            float* f = m.films[(size_t)p * virt];
            rc = rccl().reduce(f, f, pixels * 4, ncclFloat, ncclSum, 0, m.comms[p], m.streams[(size_t)p * virt]);
        }
        ncclResult_t rc2 = rccl().group_end();
        if (rc == ncclSuccess) rc = rc2;
        if (rc != ncclSuccess) { multi_release(m); return fail(PT_ERR_DEVICE, std::string("ncclReduce: ") + rccl().error_string(rc)); }
        for (int p = 0; p < np; ++p) { hipSetDevice(devices[p]); if (hipStreamSynchronize(m.streams[(size_t)p * virt]) != hipSuccess) { multi_release(m); return fail(PT_ERR_DEVICE, "stream synchronisation after the film reduce failed"); } }
    }
    const auto t1 = std::chrono::steady_clock::now();
    HIP_TRY(hipSetDevice(devices[0]));
    HIP_TRY(hipMemcpy(film, m.films[0], bytes, hipMemcpyDeviceToHost));
    if (profile) {
        memset(profile, 0, sizeof(*profile));
        for (const pt_profile& p : profiles) {
            profile->bounce_rays += p.bounce_rays; profile->shadow_rays += p.shadow_rays; profile->light_rays += p.light_rays;
            profile->camera_rays += p.camera_rays; profile->env_hits += p.env_hits;
            for (int k = 0; k < 5; ++k) { profile->kernel_seconds[k] += p.kernel_seconds[k]; profile->kernel_launches[k] += p.kernel_launches[k]; profile->stage_items[k] += p.stage_items[k]; }
            profile->stage_items[5] += p.stage_items[5];
        }
        profile->seconds = std::chrono::duration<double>(t1 - t0).count();
        profile->kernel_seconds[5] = std::chrono::duration<double>(t0 - t_entry).count();    // set-up (replicas, streams, films, communicator): ~0 on a repeated call
        profile->kernel_seconds[6] = std::chrono::duration<double>(t1 - t_reduce).count();   // the film reduce
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

pt_status pt_camera_samples(pt_scene* sc, const pt_render_desc* rdp, size_t n, const uint32_t* pixel, const uint32_t* sample, float* origins, float* directions, float* lambda) {
    if (!sc || !rdp || !pixel || !sample || !origins || !directions || !lambda) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    if (rdp->width == 0 || rdp->height == 0 || rdp->camera_index >= sc->host.cameras.size()) return fail(PT_ERR_INVALID_ARGUMENT, "width, height must be positive, camera_index in range");
    if (!(rdp->wavelength_hi >= rdp->wavelength_lo)) return fail(PT_ERR_INVALID_ARGUMENT, "wavelength_hi must not be below wavelength_lo");
    if (n > 0xffffffffull) return fail(PT_ERR_INVALID_ARGUMENT, "at most 2^32 - 1 samples per call");
    const uint64_t n_pixels = (uint64_t)rdp->width * (uint64_t)rdp->height;   // (64-bit: a 32-bit product wraps for big films and lets ids through)
    if (n_pixels > 0xffffffffull) return fail(PT_ERR_INVALID_ARGUMENT, "width x height must fit a 32-bit pixel id");
    for (size_t i = 0; i < n; ++i) if ((uint64_t)pixel[i] >= n_pixels) return fail(PT_ERR_INVALID_ARGUMENT, "pixel id out of range");
    if (n == 0) return PT_OK;
    HIP_TRY(hipSetDevice(sc->device));
    RenderParams rp;
    memset(&rp, 0, sizeof(rp));
    rp.seed = rdp->seed; rp.width = rdp->width; rp.height = rdp->height;
    rp.wavelength_lo = rdp->wavelength_lo; rp.wavelength_span = rdp->wavelength_hi - rdp->wavelength_lo;
    rp.camera = pth::camera_params(sc->host.cameras[rdp->camera_index], (float)rdp->width / (float)rdp->height);
    rp.chunk_pixels = 1;   // (stage_generate: sample = first_sample + slot / chunk_pixels — the probe hands the sample index in as the slot)
    DevBuf dp, ds, dor, dd, dl;
    HIP_TRY(dp.alloc(4 * n)); HIP_TRY(ds.alloc(4 * n)); HIP_TRY(dor.alloc(12 * n)); HIP_TRY(dd.alloc(12 * n)); HIP_TRY(dl.alloc(4 * n));
    HIP_TRY(hipMemcpy(dp.p, pixel, 4 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ds.p, sample, 4 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe_camera, dim3(256), dim3(kBlock), 0, 0, rp, (uint32_t)n, dp.as<uint32_t>(), ds.as<uint32_t>(), dor.as<float>(), dd.as<float>(), dl.as<float>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(origins, dor.p, 12 * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(directions, dd.p, 12 * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(lambda, dl.p, 4 * n, hipMemcpyDeviceToHost));
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
                    case 7: return sc->host.blob[PT_HDR_CORE_WORDS] * 4;
                    case 8: return (uint32_t)(sc->host.tex.size() > 0xffffffffull ? 0xffffffffull : sc->host.tex.size());   // words of texels + importance-map tables (read through L2, never staged)
                    default: return 0; }
}

}  // extern "C"
