// pt_engine.hip — the HIP engine behind include/pt_api.h (gfx950 / MI355X only).
//
// Wavefront path tracer: per bounce one launch each of extend -> shade -> shadow over segmented SoA queues in HBM
// (pt_stages.h; workgroup b owns segment b of every queue and compacts survivors into its own segment with an LDS prefix
// sum — no global atomics), persistent grids that stage the scene blob (or its core section) into LDS once per workgroup,
// three traversal forms (pt_device.h: BVH walk, leaf sweep, sweep + parked mesh walks), per-slot energy accumulation
// without float atomics, and an accumulate kernel that owns one film pixel per lane so film sums keep the reference's
// order.  No CPU fallback: every entry point fails with PT_ERR_NO_DEVICE when HIP has no device.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pt_api.h"
#include "pt_error.h"
#include "pt_plan.h"
#include "pt_scene_host.h"
#include "pt_stages.h"

using namespace ptd;

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

constexpr int kBlock = 256;
// Register budgets: the number of waves per SIMD the compiler must leave room for (1 = no constraint), per kernel form.
// Measured on MI355X (tools/occupancy_sweep.sh): the traversal kernels are VALU-issue bound and gain from a
// 5th wave; k_shade is a large body (196 VGPRs unconstrained = 2 waves) that gains from a 3rd wave and loses with a 4th.
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 3
#endif
#ifndef PT_SHADE4_WAVES
#define PT_SHADE4_WAVES 2
#endif
#ifndef PT_SWEEP_WAVES
#define PT_SWEEP_WAVES 5
#endif
#ifndef PT_WALK_WAVES
#define PT_WALK_WAVES 1
#endif
#ifndef PT_PARK_WAVES
#define PT_PARK_WAVES 4
#endif
#define PT_PARK_OCC __attribute__((amdgpu_waves_per_eu(PT_PARK_WAVES)))
// (forms of k_shade, see the kernel: FULL = PT_SHADE_WAVES / PT_SHADE4_WAVES above; measured with tools/shade_occupancy.sh.  FULL on C4:
// 11442 us at 3 waves, 14735 at 2, 11869 at 4; NO_ENV on C3: 3564 at 3 or 2, 3902 at 4)
#ifndef PT_SHADE_NO_ENV_WAVES
#define PT_SHADE_NO_ENV_WAVES 3
#endif
#ifndef PT_SHADE4_NO_ENV_WAVES
#define PT_SHADE4_NO_ENV_WAVES 3   // 6082 us at 3 waves, 7029 at 2, 6759 unconstrained (C5 before the lean form existed)
#endif
#ifndef PT_SHADE_LEAN_WAVES
#define PT_SHADE_LEAN_WAVES 3      // 126 VGPRs without a constraint = 4 waves; 5 waves spill (3892 vs 2866 us)
#endif
#ifndef PT_SHADE4_LEAN_WAVES
#define PT_SHADE4_LEAN_WAVES 2     // C5: 3885 us at 2 waves, 4142 at 3-4, 4830 at 5
#endif
#define PT_SHADE_OCC __attribute__((amdgpu_waves_per_eu(NL == 1 ? (FORM == 2 ? PT_SHADE_WAVES : FORM == 1 ? PT_SHADE_NO_ENV_WAVES : PT_SHADE_LEAN_WAVES) \
                                                                : (FORM == 2 ? PT_SHADE4_WAVES : FORM == 1 ? PT_SHADE4_NO_ENV_WAVES : PT_SHADE4_LEAN_WAVES))))
#define PT_TRAV_OCC __attribute__((amdgpu_waves_per_eu(TRAV == PT_TRAV_SWEEP ? PT_SWEEP_WAVES : PT_WALK_WAVES)))
constexpr uint32_t kLdsBlobLimitBytes = 64 * 1024;  // stage the blob in LDS when it fits (keeps >= 2 workgroups per CU)

enum { ST_GENERATE, ST_EXTEND, ST_SHADE, ST_SHADOW, ST_ACCUMULATE, ST_COUNT };

// ------------------------------------------------------------------------------------------------ kernels
// Every kernel is a persistent grid: blocks stage the scene blob into LDS (when USE_LDS), then walk the queue
// with a grid stride.  Queue lengths live in device memory (`counts`), so no host round trip between bounces.
// USE_LDS: 0 = everything is read from HBM/L2; 1 = the whole blob is copied to LDS; 2 = only the core section is (curves,
// materials, instances, top-level BVH, sweep table: the words every lane keeps re-reading), the mesh data stays in HBM/L2
// — scenes whose meshes do not fit the LDS budget but whose core does (C4: 470 KB of monkey, 24 KB of core).
enum { PT_LDS_NONE = 0, PT_LDS_ALL = 1, PT_LDS_CORE = 2 };
template <int USE_LDS>
__device__ __forceinline__ SceneView stage_scene(const uint32_t* __restrict__ blob, uint32_t blob_words, const float* tex, uint32_t* lds) {
    SceneView s;
    s.tex = tex;
    const uint32_t core_words = blob[PT_HDR_CORE_WORDS];
    if (USE_LDS != PT_LDS_NONE) {
        const uint32_t words = USE_LDS == PT_LDS_ALL ? blob_words : core_words;
        const uint4* src = reinterpret_cast<const uint4*>(blob);
        uint4* dst = reinterpret_cast<uint4*>(lds);
        for (uint32_t i = threadIdx.x; i < words / 4; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
        s.w = lds;
        s.m = USE_LDS == PT_LDS_ALL ? lds + core_words : blob + core_words;
    } else {
        s.w = blob;
        s.m = blob + core_words;
    }
    return s;
}

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

// ---- segmented queues -------------------------------------------------------------------------------------------
// Every queue is cut into gridDim.x segments of `seg_cap` items; workgroup b owns segment b in every kernel of a pass:
// it reads items [b*seg_cap, b*seg_cap + count_in[b]) and appends its survivors, compacted, to the same segment of the
// output queue.  Compaction is a wave64 ballot + one LDS atomic per wave (shared_append) — no global atomics (a single hot
// queue head saturates at ~88 returning atomics/us on MI355X, which was the whole cost of the first version of k_shade)
// and no barrier; each wave's writes are one contiguous run per field.  The four waves of a workgroup stride through the
// shared segment, which balances them; survival is statistically uniform over segments, which balances the workgroups.
// Append without a barrier: one LDS atomic per wave claims the wave's run in the workgroup's segment.  The order of the
// waves' runs inside the segment then depends on timing, which no result depends on (every queue item is processed on its
// own; energy and film sums are keyed by slot and pixel).  Measured on k_shade: -6 % against block_append's barrier.
__device__ __forceinline__ uint32_t shared_append(bool flag, uint32_t* lds_head) {
    unsigned long long mask = __ballot(flag);
    uint32_t start = 0;
    if (lane_id() == 0 && mask != 0ull) start = atomicAdd(lds_head, (uint32_t)__popcll(mask));
    start = (uint32_t)__builtin_amdgcn_readfirstlane((int)start);
    return start + (uint32_t)__popcll(mask & ((1ull << lane_id()) - 1ull));
}
__device__ __forceinline__ uint32_t wave_reduce_add(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;  // valid in lane 0
}

// Per-workgroup statistics (Profile counters), owned by the workgroup: plain read-modify-write, summed on the host.
enum { BS_VERTICES, BS_SHADOW_RAYS, BS_ENV_HITS, BS_SEGMENTS, BS_ITEMS, BS_FIELDS };

template <int NL>
__global__ void __launch_bounds__(kBlock) k_generate(RenderParams rp, const uint32_t* __restrict__ pixels, Queue paths, float* __restrict__ energy,
                                                    uint32_t n, uint32_t seg_cap, uint32_t* __restrict__ count_out) {
    uint32_t base = blockIdx.x * seg_cap;
    uint32_t cnt = base < n ? (n - base < seg_cap ? n - base : seg_cap) : 0u;
    for (uint32_t j = threadIdx.x; j < cnt; j += blockDim.x) {
        uint32_t slot = base + j;
        uint32_t pixel = pixels[slot % rp.chunk_pixels];
        PathVertexT<NL> p = stage_generate<NL>(rp, slot, pixel);
        store_path<NL>(paths, slot, p);
        for (int k = 0; k < NL; ++k) energy[(size_t)k * rp.energy_stride + slot] = 0.0f;
    }
    if (threadIdx.x == 0) count_out[blockIdx.x] = cnt;
}

template <int USE_LDS, int TRAV>
__global__ void __launch_bounds__(kBlock) PT_TRAV_OCC k_extend(const uint32_t* __restrict__ blob, uint32_t blob_words, const float* __restrict__ tex,
                                                  Queue paths, Queue hits, uint32_t seg_cap, const uint32_t* __restrict__ count_in) {
    extern __shared__ __align__(16) uint32_t lds[];
    SceneView s = stage_scene<USE_LDS>(blob, blob_words, tex, lds);
    uint32_t base = blockIdx.x * seg_cap, n = count_in[blockIdx.x];
    for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
        uint32_t i = base + j;
        F3 o = f3(qf(paths, PS_OX, i), qf(paths, PS_OY, i), qf(paths, PS_OZ, i));
        F3 d = f3(qf(paths, PS_DX, i), qf(paths, PS_DY, i), qf(paths, PS_DZ, i));
        Hit h;
        world_hit<TRAV>(s, o, d, &h);
        store_hit(hits, i, h);
    }
}

// FORM: what the scene can need at a vertex, so that the rest is compiled out (registers and code size, never results):
// PT_SHADE_LEAN = no light sample picks the environment (env_sampling_probability = 0) and no GGX material (the Cornell box of C2 / C5),
// PT_SHADE_NO_ENV = any material, PT_SHADE_FULL = everything.
enum { PT_SHADE_LEAN = 0, PT_SHADE_NO_ENV = 1, PT_SHADE_FULL = 2 };
template <int USE_LDS, int NL, int FORM>
__global__ void __launch_bounds__(kBlock) PT_SHADE_OCC k_shade(const uint32_t* __restrict__ blob, uint32_t blob_words, const float* __restrict__ tex,
                                                 RenderParams rp, uint32_t bounce, const uint32_t* __restrict__ pixels,
                                                 Queue paths_in, Queue hits, Queue paths_out, Queue shadow, float* __restrict__ energy,
                                                 uint32_t seg_cap, const uint32_t* __restrict__ count_in, uint32_t* __restrict__ count_out,
                                                 uint32_t* __restrict__ shadow_count, unsigned long long* __restrict__ block_stats) {
    extern __shared__ __align__(16) uint32_t lds[];
    __shared__ uint32_t lds_counts[16];  // [0] path queue head, [1] item queue head, [4..6] statistics
    if (threadIdx.x < 16) lds_counts[threadIdx.x] = 0;
    SceneView s = stage_scene<USE_LDS>(blob, blob_words, tex, lds);  // (barrier inside when staging; one below otherwise)
    if (USE_LDS == PT_LDS_NONE) __syncthreads();
    const uint32_t base = blockIdx.x * seg_cap, n = count_in[blockIdx.x];
    uint32_t st_vertices = 0, st_shadow = 0, st_env = 0;
    const uint32_t rounds = (n + blockDim.x - 1) / blockDim.x;
    for (uint32_t r = 0; r < rounds; ++r) {  // whole waves stay in the loop: the appends are ballots
        uint32_t j = r * blockDim.x + threadIdx.x;
        bool active = j < n;
        uint32_t i = base + j;
        PathVertexT<NL> pv; Hit hit; hit.valid = false;
        bool wants_item = false;
        if (active) {
            pv = load_path<NL>(paths_in, i);
            hit = load_hit(hits, i);
            wants_item = shade_wants_item(s, rp, hit);
        }
        // reserve the light-sample item first, so its rays stream straight from registers to the queue
        uint32_t ipos = base + shared_append(wants_item, &lds_counts[1]);
        ShadeOutT<NL> out;
        out.survives = false; out.has_item = false; out.vertex_pushed = false; out.env_hit = false; out.shadow_count = 0; out.add_energy = false; out.env_mask = 0;
        if (active) {
            uint32_t pixel = pixels[pv.slot % rp.chunk_pixels];
            out = stage_shade<NL, FORM == PT_SHADE_FULL, FORM != PT_SHADE_LEAN>(s, rp, bounce, pv, hit, pixel, [&](uint32_t l, const ShadowRayT<NL>& ray) { store_shadow_ray<NL>(shadow, ipos, l, ray); });
            if (wants_item) {
                float lam[NL]; lam[0] = pv.lambda;
                if (NL > 1) hero_lambdas<NL>(rp, pt_draw4(rp.seed, pixel, rp.first_sample + pv.slot / rp.chunk_pixels, PT_DIM_FILM).z, lam);
                qsu(shadow, Layout<NL>::sh_slot, ipos, pv.slot); qsu(shadow, Layout<NL>::sh_flags, ipos, out.env_mask);
                for (int k = 0; k < NL; ++k) qsf(shadow, Layout<NL>::sh_lambda + k, ipos, lam[k]);
                if (!out.has_item) clear_shadow_item<NL>(shadow, ipos, rp.light_samples);  // vertex dropped (NaN pdf, utils.rs:261-263)
            }
            if (out.add_energy) for (int k = 0; k < NL; ++k) energy[(size_t)k * rp.energy_stride + pv.slot] += out.energy_add[k];
        }
        uint32_t pos = base + shared_append(out.survives, &lds_counts[0]);
        if (out.survives) store_path<NL>(paths_out, pos, out.next);
        st_vertices += out.vertex_pushed ? 1u : 0u; st_env += out.env_hit ? 1u : 0u; st_shadow += out.shadow_count;
    }
    // workgroup totals -> this workgroup's statistics record
    st_vertices = wave_reduce_add(st_vertices); st_shadow = wave_reduce_add(st_shadow); st_env = wave_reduce_add(st_env);
    if (lane_id() == 0) { atomicAdd(&lds_counts[4], st_vertices); atomicAdd(&lds_counts[5], st_shadow); atomicAdd(&lds_counts[6], st_env); }
    __syncthreads();
    if (threadIdx.x == 0) {
        count_out[blockIdx.x] = lds_counts[0]; shadow_count[blockIdx.x] = lds_counts[1];
        unsigned long long* bs = block_stats + (size_t)blockIdx.x * BS_FIELDS;
        bs[BS_VERTICES] += lds_counts[4]; bs[BS_SHADOW_RAYS] += lds_counts[5]; bs[BS_ENV_HITS] += lds_counts[6];
        bs[BS_SEGMENTS] += n;
        bs[BS_ITEMS] += lds_counts[1];
    }
}

template <int USE_LDS, int NL, int TRAV, bool ENV = true>
__global__ void __launch_bounds__(kBlock) PT_TRAV_OCC k_shadow(const uint32_t* __restrict__ blob, uint32_t blob_words, const float* __restrict__ tex,
                                                  uint32_t light_samples, Queue shadow, float* __restrict__ energy, uint32_t energy_stride,
                                                  uint32_t seg_cap, const uint32_t* __restrict__ count_in) {
    extern __shared__ __align__(16) uint32_t lds[];
    SceneView s = stage_scene<USE_LDS>(blob, blob_words, tex, lds);
    uint32_t base = blockIdx.x * seg_cap, n = count_in[blockIdx.x];
    for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
        stage_shadow_item<NL, TRAV, ENV>(s, light_samples, shadow, base + j, energy, energy_stride);
    }
}

// ------------------------------------------------------------------------------------------------ parked traversal
// Scenes whose sweep table holds walked meshes (PT_FLAG_SWEEP_WALKS: a few analytic shapes and small meshes around one or
// more big meshes — the gem in the Cornell room, the monkey under the HDRI).  Most rays never enter a big mesh's box, and
// the ones that do are scattered over the waves, so walking the mesh in line would leave a wave waiting for a handful of
// lanes.  Instead a lane that reaches a walked-mesh bit *parks* its sweep state in the workgroup's scratch region, and
// whenever a wave has 64 rays parked (and at the end) it resumes them together: a full wave, every lane in a mesh walk.  Nothing about a ray's own sequence of tests changes (same leaves, same order, same running closest hit).
constexpr uint32_t kParkCap = 512, kParkFields = 16;
enum { PK_ITEM, PK_HIT_LO, PK_HIT_HI, PK_CLOSEST, PK_BEST_INST, PK_BEST_TRIW, PK_T, PK_B0, PK_B1, PK_B2, PK_RAY, PK_BOUND, PK_KIND };
__device__ __forceinline__ void park_store(uint32_t* pk, uint32_t e, uint32_t item, const SweepState& st, uint32_t ray, float bound, uint32_t kind) {
    pk[PK_ITEM * kParkCap + e] = item; pk[PK_HIT_LO * kParkCap + e] = (uint32_t)st.hit; pk[PK_HIT_HI * kParkCap + e] = (uint32_t)(st.hit >> 32);
    pk[PK_CLOSEST * kParkCap + e] = pt_f2u(st.closest); pk[PK_BEST_INST * kParkCap + e] = st.best_inst; pk[PK_BEST_TRIW * kParkCap + e] = st.best_triw;
    pk[PK_T * kParkCap + e] = pt_f2u(st.bh.t); pk[PK_B0 * kParkCap + e] = pt_f2u(st.bh.b0); pk[PK_B1 * kParkCap + e] = pt_f2u(st.bh.b1); pk[PK_B2 * kParkCap + e] = pt_f2u(st.bh.b2);
    pk[PK_RAY * kParkCap + e] = ray; pk[PK_BOUND * kParkCap + e] = pt_f2u(bound); pk[PK_KIND * kParkCap + e] = kind;
}
__device__ __forceinline__ void park_load(const uint32_t* pk, uint32_t e, uint32_t* item, SweepState* st, uint32_t* ray, float* bound, uint32_t* kind) {
    *item = pk[PK_ITEM * kParkCap + e]; st->hit = (uint64_t)pk[PK_HIT_LO * kParkCap + e] | (uint64_t)pk[PK_HIT_HI * kParkCap + e] << 32;
    st->closest = pt_u2f(pk[PK_CLOSEST * kParkCap + e]); st->best_inst = pk[PK_BEST_INST * kParkCap + e]; st->best_triw = pk[PK_BEST_TRIW * kParkCap + e];
    st->bh.t = pt_u2f(pk[PK_T * kParkCap + e]); st->bh.b0 = pt_u2f(pk[PK_B0 * kParkCap + e]); st->bh.b1 = pt_u2f(pk[PK_B1 * kParkCap + e]); st->bh.b2 = pt_u2f(pk[PK_B2 * kParkCap + e]);
    *ray = pk[PK_RAY * kParkCap + e]; *bound = pt_u2f(pk[PK_BOUND * kParkCap + e]); *kind = pk[PK_KIND * kParkCap + e];
}
// The resume loop shared by both kernels, per WAVE: every wave of the workgroup parks into its own quarter of the scratch
// region (128 entries: fewer than 64 left over + at most 64 new per step) and resumes 64 parked rays at a time — full
// waves — with no workgroup barrier anywhere: a wave that is deep in a mesh never holds the other three up.  (The first
// version parked per workgroup with three barriers per drain; rocprofv3 showed the C4 shadow kernel waiting 68 % of its
// wave cycles at 12 % VALU issue.)
constexpr uint32_t kWaveParkCap = kParkCap / (kBlock / 64);
template <typename Resume>
__device__ __forceinline__ void park_drain(uint32_t* pk, uint32_t* park_count, bool last, Resume&& resume) {
    const uint32_t lane = lane_id();
    for (;;) {
        __threadfence_block();             // this wave's parked entries are visible to its other lanes
        const uint32_t cnt = *park_count;  // the same for every lane of the wave
        if (!(cnt >= 64u || (last && cnt > 0u))) break;
        const uint32_t take = cnt < 64u ? cnt : 64u, first = cnt - take;
        const bool mine = lane < take;
        uint32_t item = 0, ray = 0, kind = 0; float bound = PT_INF; SweepState st;
        if (mine) park_load(pk, first + lane, &item, &st, &ray, &bound, &kind);
        __threadfence_block();             // entries are in registers before any lane parks again into these slots
        if (lane == 0) *park_count = first;
        if (mine) resume(item, st, ray, bound, kind);
    }
}

template <int USE_LDS>
__global__ void __launch_bounds__(kBlock) PT_PARK_OCC k_extend_parked(const uint32_t* __restrict__ blob, uint32_t blob_words, const float* __restrict__ tex,
                                                                     Queue paths, Queue hits, uint32_t seg_cap, const uint32_t* __restrict__ count_in,
                                                                     uint32_t* __restrict__ park_all) {
    extern __shared__ __align__(16) uint32_t lds[];
    __shared__ uint32_t park_counts[kBlock / 64];
    SceneView s = stage_scene<USE_LDS>(blob, blob_words, tex, lds);
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t* pk = park_all + (size_t)blockIdx.x * kParkFields * kParkCap + wave * kWaveParkCap;  // field f of entry e at pk[f * kParkCap + e]
    uint32_t* park_count = &park_counts[wave];
    const uint32_t base = blockIdx.x * seg_cap, n = count_in[blockIdx.x];
    const uint32_t rounds = (n + blockDim.x - 1) / blockDim.x;
    if (lane_id() == 0) *park_count = 0;
    auto ray_of = [&](uint32_t i, F3* o, F3* d) {
        *o = f3(qf(paths, PS_OX, i), qf(paths, PS_OY, i), qf(paths, PS_OZ, i));
        *d = f3(qf(paths, PS_DX, i), qf(paths, PS_DY, i), qf(paths, PS_DZ, i));
    };
    auto settle = [&](uint32_t j, F3 o, F3 d, const SweepState& st, bool parked) {
        if (parked) park_store(pk, atomicAdd(park_count, 1u), j, st, 0u, PT_INF, 0u);
        else { Hit h; sweep_finish(s, o, d, st, &h); store_hit(hits, base + j, h); }
    };
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t j = r * blockDim.x + threadIdx.x;
        if (j < n) {
            F3 o, d;
            ray_of(base + j, &o, &d);
            SweepState st;
            sweep_state_init(st, sweep_masks(s, o, d, PT_INF));
            const TriRay wtr = tri_ray_prepare(o, d);
            settle(j, o, d, st, sweep_run(s, o, d, wtr, PT_INF, PT_STOP_NONE, st, true));
        }
        park_drain(pk, park_count, r + 1 == rounds, [&](uint32_t j2, SweepState& st, uint32_t, float, uint32_t) {
            F3 o, d;
            ray_of(base + j2, &o, &d);
            settle(j2, o, d, st, sweep_resume(s, o, d, PT_INF, PT_STOP_NONE, st));
        });
    }
}

template <int USE_LDS, int NL>
__global__ void __launch_bounds__(kBlock) PT_PARK_OCC k_shadow_parked(const uint32_t* __restrict__ blob, uint32_t blob_words, const float* __restrict__ tex,
                                                                     uint32_t light_samples, Queue shadow, float* __restrict__ energy, uint32_t energy_stride,
                                                                     uint32_t seg_cap, const uint32_t* __restrict__ count_in, uint32_t* __restrict__ park_all) {
    extern __shared__ __align__(16) uint32_t lds[];
    __shared__ uint32_t park_counts[kBlock / 64];
    SceneView s = stage_scene<USE_LDS>(blob, blob_words, tex, lds);
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t* pk = park_all + (size_t)blockIdx.x * kParkFields * kParkCap + wave * kWaveParkCap;
    uint32_t* park_count = &park_counts[wave];
    const uint32_t base = blockIdx.x * seg_cap, n = count_in[blockIdx.x];
    const uint32_t rounds = (n + blockDim.x - 1) / blockDim.x;
    if (lane_id() == 0) *park_count = 0;
    // a finished ray leaves its contribution where its factor was; the item's rays are summed in order at the end (an item's
    // rays are parked and resumed by the wave that owns the item, so that sum needs no workgroup barrier either)
    auto settle = [&](uint32_t j, uint32_t l, const ShadowRayT<NL>& ray, bool env, float bound, const SweepState& st, bool parked) {
        if (parked) { park_store(pk, atomicAdd(park_count, 1u), j, st, l, bound, env ? 1u : 0u); return; }
        const uint32_t item = base + j;
        float lambda[NL], c[NL];
        for (int k = 0; k < NL; ++k) lambda[k] = qf(shadow, Layout<NL>::sh_lambda + k, item);
        Hit sh;
        bool hit = sweep_finish(s, ray.o, ray.d, st, &sh);
        shadow_ray_contribution<NL>(s, lambda, ray, env, hit, sh, c);
        for (int k = 0; k < NL; ++k) qsf(shadow, Layout<NL>::sh_head + l * Layout<NL>::sr_fields + SR_FACTOR + k, item, c[k]);
    };
    // one ray of every item per step, so that a step parks at most one ray per lane
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t j = r * blockDim.x + threadIdx.x;
        const uint32_t item = base + j, flags = j < n ? qu(shadow, Layout<NL>::sh_flags, item) : 0u;
        for (uint32_t l = 0; l < light_samples; ++l) {
            ShadowRayT<NL> ray;
            if (j < n && load_shadow_ray<NL>(shadow, item, l, &ray)) {
                const bool env = ((flags >> l) & 1u) != 0;
                float bound = PT_INF; int stop = shadow_env_stop(s);
                if (!env && !shadow_light_bound(s, ray.o, ray.d, &bound, &stop)) {
                    for (int k = 0; k < NL; ++k) qsf(shadow, Layout<NL>::sh_head + l * Layout<NL>::sr_fields + SR_FACTOR + k, item, 0.0f);
                } else {
                    SweepState st;
                    sweep_state_init(st, sweep_masks(s, ray.o, ray.d, bound));
                    const TriRay wtr = tri_ray_prepare(ray.o, ray.d);
                    settle(j, l, ray, env, bound, st, sweep_run(s, ray.o, ray.d, wtr, bound, stop, st, true));
                }
            }
            park_drain(pk, park_count, r + 1 == rounds && l + 1 == light_samples, [&](uint32_t j2, SweepState& st, uint32_t l2, float bound, uint32_t kind) {
                ShadowRayT<NL> pr;
                load_shadow_ray<NL>(shadow, base + j2, l2, &pr);
                const bool env = kind != 0u;
                // a light ray searches with the early stop whenever it has a finite bound (shadow_light_bound)
                const int stop2 = env ? shadow_env_stop(s) : (bound < PT_INF ? PT_STOP_NONLIGHT : PT_STOP_NONE);
                settle(j2, l2, pr, env, bound, st, sweep_resume(s, pr.o, pr.d, bound, stop2, st));
            });
        }
    }
    __threadfence_block();
    for (uint32_t r = 0; r < rounds; ++r) {  // pt.rs:349-392, 596: the item's rays summed in order, divided by L
        const uint32_t j = r * blockDim.x + threadIdx.x;
        if (j >= n) continue;
        const uint32_t item = base + j, slot = qu(shadow, Layout<NL>::sh_slot, item);
        float lc[NL];
        for (int k = 0; k < NL; ++k) lc[k] = 0.0f;
        for (uint32_t l = 0; l < light_samples; ++l)
            for (int k = 0; k < NL; ++k) lc[k] += qf(shadow, Layout<NL>::sh_head + l * Layout<NL>::sr_fields + SR_FACTOR + k, item);
        for (int k = 0; k < NL; ++k) energy[(size_t)k * energy_stride + slot] += lc[k] / (float)light_samples;
    }
}

template <int NL>
__global__ void __launch_bounds__(kBlock) k_accumulate(RenderParams rp, const uint32_t* __restrict__ pixels, const float* __restrict__ energy,
                                                      float* __restrict__ film) {
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < rp.chunk_pixels; p += gridDim.x * blockDim.x) {
        uint32_t pixel = pixels[p];
        float4* px = reinterpret_cast<float4*>(film) + pixel;
        float4 v = *px;
        float f[4] = {v.x, v.y, v.z, v.w};
        stage_accumulate_pixel<NL>(rp, energy, p, pixel, f);
        *px = make_float4(f[0], f[1], f[2], f[3]);
    }
}

// ---- probes (parity tests of single stages)
template <int USE_LDS>
__global__ void __launch_bounds__(kBlock) k_probe_intersect(const uint32_t* __restrict__ blob, uint32_t blob_words, const float* __restrict__ tex,
                                                           uint32_t n, const float* __restrict__ o, const float* __restrict__ d, pt_hit* __restrict__ out) {
    extern __shared__ __align__(16) uint32_t lds[];
    SceneView s = stage_scene<USE_LDS>(blob, blob_words, tex, lds);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        Hit h;
        bool ok = world_hit(s, f3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), f3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), &h);
        pt_hit r;
        memset(&r, 0, sizeof(r));
        if (ok) {
            r.valid = 1; r.t = h.t; r.point[0] = h.p.x; r.point[1] = h.p.y; r.point[2] = h.p.z;
            r.normal[0] = h.n.x; r.normal[1] = h.n.y; r.normal[2] = h.n.z; r.uv[0] = h.u; r.uv[1] = h.v;
            r.material = h.material; r.instance = h.instance;
        }
        out[i] = r;
    }
}
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
struct DeviceBuffers {
    uint32_t capacity = 0, light_samples = 0, nl = 0;
    uint32_t *paths_a = nullptr, *paths_b = nullptr, *hits = nullptr, *shadow = nullptr, *pixels = nullptr, *counts = nullptr, *park = nullptr;
    float* energy = nullptr;
    unsigned long long* block_stats = nullptr;
    size_t pixel_capacity = 0;
    int grid = 0;  // segments per queue == workgroups per launch
    void release() {
        hipFree(paths_a); hipFree(paths_b); hipFree(hits); hipFree(shadow); hipFree(pixels); hipFree(counts); hipFree(energy); hipFree(block_stats); hipFree(park);
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
    int device = 0, num_cus = 0;
    DeviceBuffers buf;
    std::vector<hipEvent_t> events;  // pairs (start, stop), grown on demand
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
    const int grid = sc->num_cus * (int)env_u32("PT_AMD_BLOCKS_PER_CU", 64);  // queue segments = workgroups per launch
    const bool hero = rd.hero_wavelengths == 4;
    if (hero && capacity > (1u << 26)) capacity = 1u << 26;  // 4-wavelength queues are ~1.5x wider: 64 Mi slots ~ 24 GB
    pt_status st = ensure_buffers(sc, capacity, rd.light_samples, pixels.size() ? pixels.size() : 1, grid, hero ? 4u : 1u);
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
    // light samples can pick the environment only if env_sampling_probability > 0: otherwise k_shade is the form without that branch
    float env_prob; std::memcpy(&env_prob, &sc->host.blob[PT_HDR_ENV_PROB], sizeof env_prob);
    bool has_ggx = false;
    for (uint32_t i = 0; i < sc->host.blob[PT_HDR_MATERIAL_COUNT]; ++i) has_ggx = has_ggx || sc->host.blob[sc->host.blob[PT_HDR_MATERIAL_OFF] + i * PT_MAT_WORDS + PT_MAT_KIND] == PT_MATERIAL_GGX;
    const int shade_form = (env_prob != 0.0f || env_u32("PT_AMD_SHADE_FORM", 0) == 2) ? PT_SHADE_FULL : (has_ggx || env_u32("PT_AMD_SHADE_FORM", 0) == 1) ? PT_SHADE_NO_ENV : PT_SHADE_LEAN;
    const uint32_t bounce_limit = rd.only_direct ? 1u : rd.max_bounces;
    const bool timing = env_u32("PT_AMD_STAGE_TIMING", 1) != 0;
    double stage_ms[ST_COUNT] = {0, 0, 0, 0, 0};
    uint64_t stage_launches[ST_COUNT] = {0, 0, 0, 0, 0};
    Queue qa{b.paths_a, b.capacity}, qb{b.paths_b, b.capacity}, qh{b.hits, b.capacity}, qs{b.shadow, b.capacity};
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
        uint32_t seg_cap = segment_capacity(n, grid);
        camera_rays += n;
        const uint32_t* d_px = b.pixels + pass.pixel_begin;
        timed(ST_GENERATE, [&] {
            if (hero) hipLaunchKernelGGL(k_generate<4>, dim3(grid), dim3(kBlock), 0, stream, rp, d_px, qa, b.energy, n, seg_cap, live[0]);
            else hipLaunchKernelGGL(k_generate<1>, dim3(grid), dim3(kBlock), 0, stream, rp, d_px, qa, b.energy, n, seg_cap, live[0]);
        });
        for (uint32_t bounce = 0; bounce < bounce_limit; ++bounce) {
            Queue qin = (bounce & 1) ? qb : qa, qout = (bounce & 1) ? qa : qb;
            uint32_t *cin = live[bounce & 1], *cout = live[(bounce + 1) & 1];
            // kernel variant = staging mode (PT_LDS_*) x traversal form x wavelengths per path
#define PT_ARGS_EXTEND sc->d_blob, sc->blob_words, sc->d_tex, qin, qh, seg_cap, cin
#define PT_ARGS_SHADE sc->d_blob, sc->blob_words, sc->d_tex, rp, bounce, d_px, qin, qh, qout, qs, b.energy, seg_cap, cin, cout, nshadow, b.block_stats
#define PT_ARGS_SHADOW sc->d_blob, sc->blob_words, sc->d_tex, rd.light_samples, qs, b.energy, b.capacity, seg_cap, nshadow
#define PT_BY_MODE(K, ...) do { if (mode == PT_LDS_ALL) launch(K(PT_LDS_ALL), lds_bytes, grid, stream, __VA_ARGS__); \
                                else if (mode == PT_LDS_CORE) launch(K(PT_LDS_CORE), lds_bytes, grid, stream, __VA_ARGS__); \
                                else launch(K(PT_LDS_NONE), lds_bytes, grid, stream, __VA_ARGS__); } while (0)
            timed(ST_EXTEND, [&] {
#define K_EXT_PARKED(M) k_extend_parked<M>
#define K_EXT_ANY(M) k_extend<M, PT_TRAV_ANY>
                if (parked) PT_BY_MODE(K_EXT_PARKED, PT_ARGS_EXTEND, b.park);
                else if (mode != PT_LDS_ALL || (sweep && walks)) PT_BY_MODE(K_EXT_ANY, PT_ARGS_EXTEND);   // (walked meshes in line: PT_AMD_NO_PARK)
                else if (sweep) launch(k_extend<PT_LDS_ALL, PT_TRAV_SWEEP>, lds_bytes, grid, stream, PT_ARGS_EXTEND);
                else launch(k_extend<PT_LDS_ALL, PT_TRAV_WALK>, lds_bytes, grid, stream, PT_ARGS_EXTEND);
            });
            timed(ST_SHADE, [&] {
#define K_SHADE1L(M) k_shade<M, 1, PT_SHADE_LEAN>
#define K_SHADE4L(M) k_shade<M, 4, PT_SHADE_LEAN>
#define K_SHADE1N(M) k_shade<M, 1, PT_SHADE_NO_ENV>
#define K_SHADE4N(M) k_shade<M, 4, PT_SHADE_NO_ENV>
#define K_SHADE1F(M) k_shade<M, 1, PT_SHADE_FULL>
#define K_SHADE4F(M) k_shade<M, 4, PT_SHADE_FULL>
                if (shade_form == PT_SHADE_FULL) { if (hero) PT_BY_MODE(K_SHADE4F, PT_ARGS_SHADE); else PT_BY_MODE(K_SHADE1F, PT_ARGS_SHADE); }
                else if (shade_form == PT_SHADE_NO_ENV) { if (hero) PT_BY_MODE(K_SHADE4N, PT_ARGS_SHADE); else PT_BY_MODE(K_SHADE1N, PT_ARGS_SHADE); }
                else if (hero) PT_BY_MODE(K_SHADE4L, PT_ARGS_SHADE); else PT_BY_MODE(K_SHADE1L, PT_ARGS_SHADE);
            });
            if (rd.light_samples > 0)
                timed(ST_SHADOW, [&] {
#define K_SH_PARKED1(M) k_shadow_parked<M, 1>
#define K_SH_PARKED4(M) k_shadow_parked<M, 4>
#define K_SH_ANY1(M) k_shadow<M, 1, PT_TRAV_ANY>
#define K_SH_ANY4(M) k_shadow<M, 4, PT_TRAV_ANY>
                    if (parked) { if (hero) PT_BY_MODE(K_SH_PARKED4, PT_ARGS_SHADOW, b.park); else PT_BY_MODE(K_SH_PARKED1, PT_ARGS_SHADOW, b.park); }
                    else if (mode != PT_LDS_ALL || (sweep && walks)) { if (hero) PT_BY_MODE(K_SH_ANY4, PT_ARGS_SHADOW); else PT_BY_MODE(K_SH_ANY1, PT_ARGS_SHADOW); }
                    else if (sweep) {   // (shade_form FULL = the scene can produce environment rays)
                        if (shade_form == PT_SHADE_FULL) {
                            if (hero) launch(k_shadow<PT_LDS_ALL, 4, PT_TRAV_SWEEP, true>, lds_bytes, grid, stream, PT_ARGS_SHADOW);
                            else launch(k_shadow<PT_LDS_ALL, 1, PT_TRAV_SWEEP, true>, lds_bytes, grid, stream, PT_ARGS_SHADOW);
                        } else if (hero) launch(k_shadow<PT_LDS_ALL, 4, PT_TRAV_SWEEP, false>, lds_bytes, grid, stream, PT_ARGS_SHADOW);
                        else launch(k_shadow<PT_LDS_ALL, 1, PT_TRAV_SWEEP, false>, lds_bytes, grid, stream, PT_ARGS_SHADOW);
                    } else {
                        if (hero) launch(k_shadow<PT_LDS_ALL, 4, PT_TRAV_WALK>, lds_bytes, grid, stream, PT_ARGS_SHADOW);
                        else launch(k_shadow<PT_LDS_ALL, 1, PT_TRAV_WALK>, lds_bytes, grid, stream, PT_ARGS_SHADOW);
                    }
                });
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
    float *dl = nullptr, *da = nullptr, *db = nullptr, *df = nullptr, *dwo = nullptr, *dp = nullptr;
    size_t m = n ? n : 1;
    HIP_TRY(hipMalloc(&dl, 4 * m)); HIP_TRY(hipMalloc(&da, 4 * m * 3)); HIP_TRY(hipMalloc(&db, 4 * m * 3));
    HIP_TRY(hipMalloc(&df, 4 * m)); HIP_TRY(hipMalloc(&dwo, 4 * m * 3)); HIP_TRY(hipMalloc(&dp, 4 * m));
    HIP_TRY(hipMemcpy(dl, lambda, 4 * n, hipMemcpyHostToDevice));
    if (a) HIP_TRY(hipMemcpy(da, a, 4 * n * a_w, hipMemcpyHostToDevice));
    if (b) HIP_TRY(hipMemcpy(db, b, 4 * n * b_w, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe_material, dim3(256), dim3(kBlock), 0, 0, sc->d_blob, sc->d_tex, mode, record, (uint32_t)n, dl, da, db, df, dwo, dp);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (f) HIP_TRY(hipMemcpy(f, df, 4 * n, hipMemcpyDeviceToHost));
    if (wo) HIP_TRY(hipMemcpy(wo, dwo, 4 * n * 3, hipMemcpyDeviceToHost));
    if (pdf) HIP_TRY(hipMemcpy(pdf, dp, 4 * n, hipMemcpyDeviceToHost));
    hipFree(dl); hipFree(da); hipFree(db); hipFree(df); hipFree(dwo); hipFree(dp);
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

pt_status pt_scene_create(const pt_scene_desc* desc, pt_scene** out) {
    if (!desc || !out) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    pt_status st = ensure_device();
    if (st != PT_OK) return st;
    pt_scene* sc = new pt_scene();
    std::string err;
    if (!pth::build_host_scene(*desc, &sc->host, &err)) { delete sc; return fail(PT_ERR_INVALID_ARGUMENT, err); }
    hipError_t e = hipGetDevice(&sc->device);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, sc->device);
    if (e != hipSuccess) { delete sc; return fail(PT_ERR_NO_DEVICE, hipGetErrorString(e)); }
    sc->num_cus = prop.multiProcessorCount;
    if (env_u32("PT_AMD_EXACT_SLAB", 0)) sc->host.blob[PT_HDR_FLAGS] |= PT_FLAG_EXACT_SLAB;
    if (env_u32("PT_AMD_NO_CULL", 0)) sc->host.blob[PT_HDR_FLAGS] |= PT_FLAG_NO_CULL;
    if (env_u32("PT_AMD_NO_SWEEP", 0)) sc->host.blob[PT_HDR_FLAGS] |= PT_FLAG_NO_SWEEP;
    if (env_u32("PT_AMD_NO_MESH_SWEEP", 0)) sc->host.blob[PT_HDR_FLAGS] |= PT_FLAG_NO_MESH_SWEEP;
    sc->blob_words = (uint32_t)sc->host.blob.size();
    const bool no_lds = env_u32("PT_AMD_NO_LDS", 0) != 0;
    sc->lds_mode = no_lds ? PT_LDS_NONE : (sc->blob_words * 4 <= kLdsBlobLimitBytes ? PT_LDS_ALL
                 : (sc->host.blob[PT_HDR_CORE_WORDS] * 4 <= kLdsBlobLimitBytes && !env_u32("PT_AMD_NO_CORE_LDS", 0) ? PT_LDS_CORE : PT_LDS_NONE));
    e = hipMalloc(&sc->d_blob, sizeof(uint32_t) * sc->host.blob.size());
    if (e == hipSuccess) e = hipMalloc(&sc->d_tex, sizeof(float) * (sc->host.tex.size() + 4));
    if (e == hipSuccess) e = hipMemcpy(sc->d_blob, sc->host.blob.data(), sizeof(uint32_t) * sc->host.blob.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(sc->d_tex, sc->host.tex.data(), sizeof(float) * sc->host.tex.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { pt_scene_destroy(sc); return fail(e == hipErrorOutOfMemory ? PT_ERR_OUT_OF_MEMORY : PT_ERR_DEVICE, hipGetErrorString(e)); }
    if (sc->lds_mode != PT_LDS_NONE) {
        auto allow = [](const void* k) { hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBlobLimitBytes); };
#define PT_ALLOW_MODES(K) allow(reinterpret_cast<const void*>(K(PT_LDS_ALL))); allow(reinterpret_cast<const void*>(K(PT_LDS_CORE)))
#define K1(M) k_extend<M, PT_TRAV_ANY>
#define K2(M) k_shadow<M, 1, PT_TRAV_ANY>
#define K3(M) k_shadow<M, 4, PT_TRAV_ANY>
#define K4(M) k_shade<M, 1, PT_SHADE_LEAN>
#define K5(M) k_shade<M, 4, PT_SHADE_LEAN>
#define K4E(M) k_shade<M, 1, PT_SHADE_FULL>
#define K5E(M) k_shade<M, 4, PT_SHADE_FULL>
#define K4N(M) k_shade<M, 1, PT_SHADE_NO_ENV>
#define K5N(M) k_shade<M, 4, PT_SHADE_NO_ENV>
#define K6(M) k_extend_parked<M>
#define K7(M) k_shadow_parked<M, 1>
#define K8(M) k_shadow_parked<M, 4>
#define K9(M) k_probe_intersect<M>
        PT_ALLOW_MODES(K1); PT_ALLOW_MODES(K2); PT_ALLOW_MODES(K3); PT_ALLOW_MODES(K4); PT_ALLOW_MODES(K5); PT_ALLOW_MODES(K4E); PT_ALLOW_MODES(K5E); PT_ALLOW_MODES(K4N); PT_ALLOW_MODES(K5N); PT_ALLOW_MODES(K6); PT_ALLOW_MODES(K7); PT_ALLOW_MODES(K8); PT_ALLOW_MODES(K9);
#undef K1
#undef K2
#undef K3
#undef K4
#undef K5
#undef K4E
#undef K5E
#undef K4N
#undef K5N
#undef K6
#undef K7
#undef K8
#undef K9
#undef PT_ALLOW_MODES
        allow(reinterpret_cast<const void*>(k_extend<PT_LDS_ALL, PT_TRAV_WALK>)); allow(reinterpret_cast<const void*>(k_extend<PT_LDS_ALL, PT_TRAV_SWEEP>));
        allow(reinterpret_cast<const void*>(k_shadow<PT_LDS_ALL, 1, PT_TRAV_WALK>)); allow(reinterpret_cast<const void*>(k_shadow<PT_LDS_ALL, 1, PT_TRAV_SWEEP>));
        allow(reinterpret_cast<const void*>(k_shadow<PT_LDS_ALL, 4, PT_TRAV_WALK>)); allow(reinterpret_cast<const void*>(k_shadow<PT_LDS_ALL, 4, PT_TRAV_SWEEP>));
        allow(reinterpret_cast<const void*>(k_shadow<PT_LDS_ALL, 1, PT_TRAV_SWEEP, false>)); allow(reinterpret_cast<const void*>(k_shadow<PT_LDS_ALL, 4, PT_TRAV_SWEEP, false>));
    }
    *out = sc;
    return PT_OK;
}

void pt_scene_destroy(pt_scene* sc) {
    if (!sc) return;
    hipSetDevice(sc->device);
    sc->buf.release();
    hipFree(sc->d_blob); hipFree(sc->d_tex);
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
    float* d_film = nullptr;
    size_t bytes = sizeof(float) * 4 * (size_t)rd->width * rd->height;
    HIP_TRY(hipMalloc(&d_film, bytes));
    pt_status st = render_impl(sc, rd, d_film, nullptr, profile);
    if (st == PT_OK) {
        hipError_t e = hipMemcpy(film, d_film, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) st = fail(PT_ERR_DEVICE, hipGetErrorString(e));
    }
    hipFree(d_film);
    return st;
}

pt_status pt_intersect(pt_scene* sc, size_t n, const float* origins, const float* directions, pt_hit* hits) {
    if (!sc || !origins || !directions || !hits) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return PT_OK;
    HIP_TRY(hipSetDevice(sc->device));
    float *dor = nullptr, *dd = nullptr; pt_hit* dh = nullptr;
    HIP_TRY(hipMalloc(&dor, 12 * n)); HIP_TRY(hipMalloc(&dd, 12 * n)); HIP_TRY(hipMalloc(&dh, sizeof(pt_hit) * n));
    HIP_TRY(hipMemcpy(dor, origins, 12 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dd, directions, 12 * n, hipMemcpyHostToDevice));
    int grid = sc->num_cus * 4;
    const uint32_t lds_bytes = sc->lds_mode == PT_LDS_ALL ? sc->blob_words * 4u : (sc->lds_mode == PT_LDS_CORE ? sc->host.blob[PT_HDR_CORE_WORDS] * 4u : 0u);
    if (sc->lds_mode == PT_LDS_ALL) launch(k_probe_intersect<PT_LDS_ALL>, lds_bytes, grid, (hipStream_t)0, sc->d_blob, sc->blob_words, sc->d_tex, (uint32_t)n, dor, dd, dh);
    else if (sc->lds_mode == PT_LDS_CORE) launch(k_probe_intersect<PT_LDS_CORE>, lds_bytes, grid, (hipStream_t)0, sc->d_blob, sc->blob_words, sc->d_tex, (uint32_t)n, dor, dd, dh);
    else launch(k_probe_intersect<PT_LDS_NONE>, lds_bytes, grid, (hipStream_t)0, sc->d_blob, sc->blob_words, sc->d_tex, (uint32_t)n, dor, dd, dh);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(hits, dh, sizeof(pt_hit) * n, hipMemcpyDeviceToHost));
    hipFree(dor); hipFree(dd); hipFree(dh);
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
    float *dx = nullptr, *dy = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(&dx, 4 * n)); HIP_TRY(hipMalloc(&dy, 4 * n)); HIP_TRY(hipMalloc(&dout, 4 * n));
    HIP_TRY(hipMemcpy(dx, x, 4 * n, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(dy, y, 4 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe_numerics, dim3(256), dim3(kBlock), 0, 0, which, (uint32_t)n, dx, dy, dout);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout, 4 * n, hipMemcpyDeviceToHost));
    hipFree(dx); hipFree(dy); hipFree(dout);
    return PT_OK;
}

// Not part of pt_api.h: size of the scene blob and whether kernels read it from LDS (reported by bench.py).
uint32_t pt_debug_scene_info(pt_scene* sc, int what) {
    switch (what) { case 0: return sc->blob_words * 4; case 1: return (uint32_t)sc->lds_mode; case 2: return sc->host.light_count; case 3: return (uint32_t)sc->num_cus;
                    case 4: return sc->host.blob[PT_HDR_SWEEP_OFF] != 0 && !(sc->host.blob[PT_HDR_FLAGS] & PT_FLAG_NO_SWEEP) ? 1u : 0u; default: return 0; }
}

}  // extern "C"
