// pt_kernels.h — the kernels of the wavefront path tracer (gfx950 / MI355X), as templates.
//
// Per bounce one launch each of extend -> shade -> shadow over segmented SoA queues in HBM (pt_stages.h; workgroup b owns segment b
// of every queue and compacts survivors into its own segment with wave ballots + one LDS atomic per wave — no global atomics),
// grids of 64 workgroups per CU that stage the scene blob (or its core section) into LDS once per workgroup, the traversal forms of
// pt_device.h (BVH walk, leaf sweep, sweep + parked mesh walks with static segments or units taken from a counter, and the pooled leaf
// sweep kept as a measured experiment), per-slot energy accumulation without float atomics, and an accumulate kernel that owns one film
// pixel per lane so film sums keep the reference's order.
//
// The kernels are instantiated where they are launched: one translation unit per kernel family (pt_kern_extend / shadow / shade.hip behind
// the launchers of pt_launch.h), so that the build compiles the families side by side and a change to one recompiles one file.
#ifndef PT_KERNELS_H
#define PT_KERNELS_H
#include <hip/hip_runtime.h>
#define PT_WAVE_KERNELS 1   /* pt_device.h: the wave-level device code (needs the HIP runtime header above) */
#include "pt_launch.h"

namespace ptk {
using namespace ptd;

template <typename K, typename... Args>
inline void go(const LaunchCfg& c, K kernel, Args... args) { hipLaunchKernelGGL(kernel, dim3(c.grid), dim3(kBlock), c.lds_bytes, c.stream, args...); }
// (the parked kernels in their bigger workgroups, round 4: `block` threads, `lds_bytes` of dynamic LDS)
template <typename K, typename... Args>
inline void go_block(const LaunchCfg& c, int block, uint32_t lds_bytes, K kernel, Args... args) { hipLaunchKernelGGL(kernel, dim3(c.grid), dim3(block), lds_bytes, c.stream, args...); }

// Register budgets: the number of waves per SIMD the compiler must leave room for (1 = no constraint), per kernel form.
// Measured on MI355X (tools/occupancy_sweep.sh): the traversal kernels are VALU-issue bound and gain from a
// 5th wave; k_shade is a large body (196 VGPRs unconstrained = 2 waves) that gains from a 3rd wave and loses with a 4th.
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 4   // (round 4: the FULL forms at four waves with 26 registers spilled — C4's k_shade 9252 -> 8560 us, C4 +4.8 %, G1 +1 %: the kernel waits for its table fetches; profiles/r4v_full4.txt)
#endif
#ifndef PT_MEDIUM_SPLIT
#define PT_MEDIUM_SPLIT 1   /* k_shade_medium: 1 = its surface vertices in waves of their own (round 5); 2 = and its free flights (measured slower: 4437 -> 5103 us); 0 = every vertex where its item lies */
#endif
#ifndef PT_SHADE_MEDIUM_WAVES
#define PT_SHADE_MEDIUM_WAVES 3
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
// (the closest-hit form gains from a fifth wave, the light-sample form loses: tools/park_occupancy.sh, C3 k_extend_parked 3288 / 2708 / 2797 / 2883 us and
// k_shadow_parked 3745 / 3919 / 3917 / 4142 us at 4 / 5 / 6 / 8 waves; C4 1222 / 1117 / 1156 / 1199 and 6503 / 7018 / 7362 / 7411)
#ifndef PT_PARK_EXTEND_WAVES
#define PT_PARK_EXTEND_WAVES 5
#endif
#define PT_PARK_EXTEND_OCC __attribute__((amdgpu_waves_per_eu(PT_PARK_EXTEND_WAVES)))
// (forms of k_shade, see the kernel: FULL = PT_SHADE_WAVES / PT_SHADE4_WAVES above; measured with tools/shade_occupancy.sh.  FULL on C4:
// 11442 us at 3 waves, 14735 at 2, 11869 at 4; NO_ENV on C3: 3564 at 3 or 2, 3902 at 4)
#ifndef PT_SHADE_NO_ENV_WAVES
#ifndef PT_SHADE_SORT
#define PT_SHADE_SORT 0   /* 1: k_shade's NO_ENV and FULL forms shade their surface vertices sorted by material kind (kSort in the kernel; round 6, EXPERIMENT).  Built, bit-identical
                             (233 GPU tests) and measured on one box, alternating (profiles/r6q_ab_sort.txt): C3 k_shade 2442 / 2469 -> 2305 / 2427 us (C3 +1.5 % / +0.1 %), but G1 2699 ->
                             2811, C4 7312 -> 7553 and G2F 2385 -> 2979 us (-10 %): the classifying turn is one more exposed round trip to the hit queue per 64 vertices and the sorted waves
                             gather their records from several tiles; the vertex kernel is not short of lanes but of memory latency.  Off. */
#endif
#define PT_SHADE_NO_ENV_WAVES 4   // (round 4, built without machine LICM: 141 VGPRs of demand; at four waves C3's k_shade 2650 -> 2505 us, profiles/r4t_noenv4.txt; round 2's 156-register form lost at four)
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
// (the fused form — k_shade that traces its own segment, FUSE_TRAV below — holds the traversal too: its own budget)
#ifndef PT_FUSED_WAVES
#define PT_FUSED_WAVES 4
#endif
#ifndef PT_FUSED4_WAVES
#define PT_FUSED4_WAVES 3
#endif
#define PT_SHADE_OCC __attribute__((amdgpu_waves_per_eu(FUSE_TRAV != PT_NO_FUSE ? (NL == 1 ? PT_FUSED_WAVES : PT_FUSED4_WAVES) : \
                                                        NL == 1 ? (FORM == 2 ? PT_SHADE_WAVES : FORM == 1 ? PT_SHADE_NO_ENV_WAVES : PT_SHADE_LEAN_WAVES) \
                                                                : (FORM == 2 ? PT_SHADE4_WAVES : FORM == 1 ? PT_SHADE4_NO_ENV_WAVES : PT_SHADE4_LEAN_WAVES))))
#define PT_NO_FUSE (-1)
#define PT_TRAV_OCC __attribute__((amdgpu_waves_per_eu(TRAV == PT_TRAV_SWEEP ? PT_SWEEP_WAVES : PT_WALK_WAVES)))
// (k_shadow's sweep form takes a sixth wave: measured on one box after the round-2 changes, tools/occupancy_c2.sh: k_shadow 5203 / 4622 / 4395 /
// 5479 us at 4 / 5 / 6 / 8 waves, k_extend 2610 / 2270 / 2362 / 3989)
#ifndef PT_SHADOW_SWEEP_WAVES
#define PT_SHADOW_SWEEP_WAVES 6
#endif
// (with four wavelengths per path the six-wave form spills 9 registers, the five-wave form none: 91 VGPRs)
#ifndef PT_SHADOW4_SWEEP_WAVES
#define PT_SHADOW4_SWEEP_WAVES 6
#endif
#ifndef PT_SHADOW_LIVE_WAVES
#define PT_SHADOW_LIVE_WAVES 6   // (k_shadow_live: measured below)
#endif
// (round 4, built without machine LICM: the form without transformed instances — C2's — needs 72 VGPRs and takes EIGHT waves with 7 registers spilled, 32 B of
// scratch: 3260 -> 3175 us on one box, twice (profiles/r4j_flags.txt); the general form would spill 35 at eight and stays at six)
#ifndef PT_SHADOW_SWEEP_NOXF_WAVES
#define PT_SHADOW_SWEEP_NOXF_WAVES 8
#endif
// (the hero form likewise: C5 k_shadow 2257 -> 2192 us at eight waves, 32-40 B of scratch; profiles/r4l_hero8.txt)
#ifndef PT_SHADOW4_SWEEP_NOXF_WAVES
#define PT_SHADOW4_SWEEP_NOXF_WAVES 8
#endif
#define PT_SHADOW_OCC __attribute__((amdgpu_waves_per_eu(TRAV == PT_TRAV_SWEEP ? (NL == 1 ? ((LACKS & PT_SCENE_NO_XF) ? PT_SHADOW_SWEEP_NOXF_WAVES : PT_SHADOW_SWEEP_WAVES) \
                                                                                           : ((LACKS & PT_SCENE_NO_XF) ? PT_SHADOW4_SWEEP_NOXF_WAVES : PT_SHADOW4_SWEEP_WAVES)) : PT_WALK_WAVES)))

// The lean form of k_shade (closed scenes: nearly every segment ends on a surface) reads the whole hit record at once instead of waiting for its
// first word to say whether there is a hit (load_hit<EAGER>, pt_stages.h): -1.5 % of the kernel on C2.
#ifndef PT_SHADE_EAGER
#define PT_SHADE_EAGER true
#endif

enum { ST_GENERATE, ST_EXTEND, ST_SHADE, ST_SHADOW, ST_ACCUMULATE, ST_COUNT };
// a wave's list of live light-sample rays (k_shadow_live, k_shadow_parked): fewer than 64 left over + the rays of 64 items — in dynamic LDS behind the
// staged blob, sized by the render's own light_samples (launch_shadow), so that the staged blob keeps its workgroups per CU
__host__ __device__ constexpr uint32_t live_cap(uint32_t light_samples) { return 64u * light_samples + 64u; }

// ------------------------------------------------------------------------------------------------ kernels
// Every kernel is a persistent grid: blocks stage the scene blob into LDS (when USE_LDS), then walk the queue
// with a grid stride.  Queue lengths live in device memory (`counts`), so no host round trip between bounces.
// USE_LDS: 0 = everything is read from HBM/L2; 1 = the whole blob is copied to LDS; 2 = only the core section is (curves,
// materials, instances, top-level BVH, sweep table: the words every lane keeps re-reading), the mesh data stays in HBM/L2
// — scenes whose meshes do not fit the LDS budget but whose core does (C4: 470 KB of monkey, 24 KB of core).
#ifndef PT_EXP_LACKS
#define PT_EXP_LACKS 0u   /* register-pressure experiments in the build container only (tools/isa_pressure.py on one kernel): what that kernel is compiled to
                             assume the scene lacks, whatever its template says.  Never for a library that renders: a scene holding the thing would be wrong. */
#endif
#ifndef PT_EMPTY_SEGMENT_EXIT
#define PT_EMPTY_SEGMENT_EXIT 1   /* a workgroup whose segment is empty returns before it stages the scene (round 5); 0 = stages it, then finds nothing to do */
#endif
template <int USE_LDS, uint32_t LACKS = 0u>
__device__ __forceinline__ SceneView stage_scene(const uint32_t* __restrict__ blob, uint32_t blob_words, const float* tex, uint32_t* lds) {
    SceneView s;
    s.tex = tex;
    s.lacks = LACKS | PT_EXP_LACKS;
    s.certs = blob[PT_HDR_FLAGS] & PT_FLAG_CONVEX;   // (a scalar load of a kernel argument's target: SceneView::certs)
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
// The lane's number computed where it stands (two instructions the compiler may not hoist): for a loop whose only use of threadIdx.x is the
// item index of a round — the input register then dies at once instead of being spilled across the kernel (k_extend_parked: its one spill).
__device__ __forceinline__ uint32_t fresh_lane_id() {
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

#ifdef PT_TIMELINE
// (measurement build, tools/wave_timeline.py: when every wave of a kernel family's main kernel began and ended — 100 MHz clock —, how many items
// its segment held and how much of the family's own unit of work it did; one record per launch, workgroup and wave.  Every family's
// translation unit has its own copy and its own accessor, PT_TL_ACCESSOR.)
constexpr uint32_t kTlLaunches = 32, kTlBlocks = 16384;
static __device__ unsigned long long g_tl[kTlLaunches][kTlBlocks][4][4];
static __device__ uint32_t g_tl_launch;
static __global__ void k_tl_bump() { g_tl_launch = g_tl_launch + 1u; }
#define PT_TL_BEGIN() const unsigned long long tl0 = wall_clock64(); uint32_t tl_work = 0
#define PT_TL_WORK() (++tl_work)
#define PT_TL_END(n) do { if (g_tl_launch < kTlLaunches && blockIdx.x < kTlBlocks) { unsigned long long* rec = g_tl[g_tl_launch][blockIdx.x][threadIdx.x >> 6]; \
    if (lane_id() == 0) { rec[0] = tl0; rec[1] = wall_clock64(); rec[2] = (n); } atomicAdd(&rec[3], (unsigned long long)tl_work); } } while (0)
#define PT_TL_BUMP(stream) hipLaunchKernelGGL(k_tl_bump, dim3(1), dim3(1), 0, stream)
#define PT_TL_ACCESSOR(fn) extern "C" int fn(unsigned long long* out, size_t bytes, uint32_t* launches) { \
    if (hipDeviceSynchronize() != hipSuccess) return 1; \
    if (hipMemcpyFromSymbol(launches, HIP_SYMBOL(g_tl_launch), sizeof(uint32_t)) != hipSuccess) return 2; \
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl), bytes < sizeof(g_tl) ? bytes : sizeof(g_tl)) != hipSuccess) return 3; \
    const uint32_t zero = 0; void* p = nullptr; \
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_tl_launch), &zero, sizeof(zero)) != hipSuccess) return 4; \
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_tl)) != hipSuccess || hipMemset(p, 0, sizeof(g_tl)) != hipSuccess) return 5; \
    return 0; }
#else
#define PT_TL_BEGIN()
#define PT_TL_WORK()
#define PT_TL_END(n)
#define PT_TL_BUMP(stream)
#endif

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
// The importance map's marginal tables (rows x (cmf, pdf) pairs + the guide of rows + 3 words: 12 KB for 1024 rows) into LDS behind the staged blob.  Only the
// interleaved layout (PT_HDR_IMAP_STRIDE = 2: one contiguous block) is staged; marginal_lds_bytes (pt_launch.h) is the host's side of this layout.
template <int USE_LDS>
__device__ __forceinline__ void stage_marginal(SceneView& s, const uint32_t* __restrict__ blob, uint32_t blob_words, uint32_t* lds) {
    const uint32_t rows = blob[PT_HDR_IMAP_ROWS], stride = blob[PT_HDR_IMAP_STRIDE];
    if (blob[PT_HDR_ENV_KIND] != PT_ENV_HDR || rows == 0u || rows > PT_MARG_LDS_MAX_ROWS || stride != 2u) return;
    const uint32_t used = USE_LDS == PT_LDS_ALL ? blob_words : (USE_LDS == PT_LDS_CORE ? blob[PT_HDR_CORE_WORDS] : 0u);
    if (((used + 3u) & ~3u) * 4u + PT_MARG_LDS_BYTES(rows, blob[PT_HDR_IMAP_MARG_GUIDE] != 0u) > PT_SHADE_LDS_BUDGET) return;   // (marginal_lds_bytes gave 0: nothing was reserved)
    float* dst = reinterpret_cast<float*>(lds + ((used + 3u) & ~3u));
    const uint32_t a = blob[PT_HDR_IMAP_MARG_PDF], b = blob[PT_HDR_IMAP_MARG_CMF], base = a < b ? a : b, pairs = 2u * rows, mg = blob[PT_HDR_IMAP_MARG_GUIDE];
    for (uint32_t i = threadIdx.x; i < pairs; i += blockDim.x) dst[i] = s.tex[base + i];
    if (mg != 0u) for (uint32_t i = threadIdx.x; i < rows + 3u; i += blockDim.x) dst[pairs + i] = s.tex[mg + i];
    __syncthreads();
    s.marg = dst; s.marg_words = pairs; s.marg_base = base; s.marg_guide = mg != 0u ? pairs : 0u;
}
__device__ __forceinline__ uint32_t shared_append(bool flag, uint32_t* lds_head) {
    unsigned long long mask = __ballot(flag);
    uint32_t start = 0;
    if (lane_id() == 0 && mask != 0ull) start = atomicAdd(lds_head, (uint32_t)__popcll(mask));
    start = (uint32_t)__builtin_amdgcn_readfirstlane((int)start);
    return start + (uint32_t)__popcll(mask & ((1ull << lane_id()) - 1ull));
}
// Two appends with one wait: both atomics are issued before either result is read.
__device__ __forceinline__ void shared_append2(bool flag_a, uint32_t* head_a, bool flag_b, uint32_t* head_b, uint32_t* pos_a, uint32_t* pos_b) {
    const unsigned long long ma = __ballot(flag_a), mb = __ballot(flag_b);
    uint32_t sa = 0, sb = 0;
    if (lane_id() == 0) {
        if (ma != 0ull) sa = atomicAdd(head_a, (uint32_t)__popcll(ma));
        if (mb != 0ull) sb = atomicAdd(head_b, (uint32_t)__popcll(mb));
    }
    sa = (uint32_t)__builtin_amdgcn_readfirstlane((int)sa); sb = (uint32_t)__builtin_amdgcn_readfirstlane((int)sb);
    const unsigned long long below = (1ull << lane_id()) - 1ull;
    *pos_a = sa + (uint32_t)__popcll(ma & below); *pos_b = sb + (uint32_t)__popcll(mb & below);
}
__device__ __forceinline__ uint32_t wave_reduce_add(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;  // valid in lane 0
}

// Per-workgroup statistics (Profile counters), owned by the workgroup: plain read-modify-write, summed on the host.

template <int NL>
__global__ void __launch_bounds__(kBlock) k_generate(RenderParams rp, const uint32_t* __restrict__ pixels, Queue paths, float* __restrict__ energy,
                                                    uint32_t n, uint32_t seg_cap, uint32_t* __restrict__ count_out) {
    uint32_t base = blockIdx.x * seg_cap;
    uint32_t cnt = base < n ? (n - base < seg_cap ? n - base : seg_cap) : 0u;
    for (uint32_t j = threadIdx.x; j < cnt; j += blockDim.x) {
        uint32_t slot = base + j;
        uint32_t pixel = pixels[slot % rp.chunk_pixels];
        float u = 0.0f;
        PathVertexT<NL> p = stage_generate<NL>(rp, slot, pixel, &u);
        if (PT_CAMERA_RECORD && rp.camera_record) store_path_camera<NL>(paths, slot, p); else store_path<NL>(paths, slot, p);
        for (int k = 0; k < NL; ++k) energy[(size_t)k * rp.energy_stride + slot] = 0.0f;
        if (PT_STORED_WAVELENGTH) energy[(size_t)NL * rp.energy_stride + slot] = u;   // (the plane behind the energies: k_accumulate's wavelength sample)
    }
    if (threadIdx.x == 0) count_out[blockIdx.x] = cnt;
}

// LACKS (here and in k_shade / k_shadow): what the scene is known not to hold (PT_SCENE_*, pt_device.h) — compiled out of the form.
template <int USE_LDS, int TRAV, uint32_t LACKS = 0u>
__global__ void __launch_bounds__(kBlock) PT_TRAV_OCC k_extend(const uint32_t* __restrict__ blob, uint32_t blob_words, const float* __restrict__ tex,
                                                  Queue paths, Queue hits, uint32_t seg_cap, const uint32_t* __restrict__ count_in) {
    extern __shared__ __align__(16) uint32_t lds[];
    if (PT_EMPTY_SEGMENT_EXIT && count_in[blockIdx.x] == 0u) return;   // (an empty segment: nothing to stage the scene for — the deep bounces of an open scene are mostly such launches)
    SceneView s = stage_scene<USE_LDS, LACKS>(blob, blob_words, tex, lds);
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
// FUSE_TRAV: PT_NO_FUSE = the closest hits come from k_extend through the hit queue; a traversal form (PT_TRAV_*) = this kernel traces its
// own segments first (World::hit, then the vertex, as random_walk's loop body does: utils.rs:171-221) — the 44-byte hit record never
// reaches HBM, the ray is read once, and one launch per bounce goes away.  Not for the FULL form (it sorts its items by the hit queue).
template <int USE_LDS, int NL, int FORM, uint32_t LACKS = 0u, int FUSE_TRAV = PT_NO_FUSE>
__global__ void __launch_bounds__(kBlock) PT_SHADE_OCC k_shade(const uint32_t* __restrict__ blob, uint32_t blob_words, const float* __restrict__ tex,
                                                 RenderParams rp, uint32_t bounce, const uint32_t* __restrict__ pixels,
                                                 Queue paths_in, Queue hits, Queue paths_out, Queue shadow, float* __restrict__ energy,
                                                 uint32_t seg_cap, const uint32_t* __restrict__ count_in, uint32_t* __restrict__ count_out,
                                                 uint32_t* __restrict__ shadow_count, unsigned long long* __restrict__ block_stats) {
    extern __shared__ __align__(16) uint32_t lds[];
    __shared__ uint32_t lds_counts[16];  // [0] path queue head, [1] item queue head, [4..6] statistics
    if (PT_EMPTY_SEGMENT_EXIT && count_in[blockIdx.x] == 0u) {   // (an empty segment leaves empty segments: no scene staged, no barrier met)
        if (threadIdx.x == 0) { count_out[blockIdx.x] = 0u; shadow_count[blockIdx.x] = 0u; shadow_count[gridDim.x + blockIdx.x] = 0u; }
        return;
    }
    if (threadIdx.x < 16) lds_counts[threadIdx.x] = 0;
    SceneView s = stage_scene<USE_LDS, LACKS>(blob, blob_words, tex, lds);  // (barrier inside when staging; one below otherwise)
    if (FORM == PT_SHADE_FULL) stage_marginal<USE_LDS>(s, blob, blob_words, lds);   // (the importance map's marginal tables behind the blob: launch_shade sized the LDS for them)
    if (USE_LDS == PT_LDS_NONE) __syncthreads();
    const uint32_t base = blockIdx.x * seg_cap, n = count_in[blockIdx.x];
    uint32_t st_vertices = 0, st_shadow = 0, st_env = 0;
    const uint32_t rounds = (n + blockDim.x - 1) / blockDim.x;
    PT_TL_BEGIN();
    // FULL form (scenes with an environment that light samples can pick): a path that left the scene is a cheap vertex (one MIS-weighted
    // emission), a surface vertex an expensive one (C4: six light samples through the importance map), and a wave that holds both runs the
    // expensive part with the lanes of the cheap ones idle (C4: 38 % of the vertices, lane utilisation 0.60).  So a wave shades the
    // environment vertices of its 64 items at once and keeps the surface vertices' indices in a list of its own, shading them 64 at a time
    // whenever the list holds that many (and what is left at the end): full waves in the expensive part.  No vertex' result depends on
    // when it is shaded (queue appends are order-free, energy is keyed by slot).  One call site of the vertex body for both.
    constexpr bool kSplit = FORM == PT_SHADE_FULL;
    // SORTED forms (round 6; the forms that hold the microfacet code, unfused): the surface vertices go to TWO lists per wave, by the kind of their material — the microfacet
    // vertices (GGX, PassthroughFilter) to the second — and each list is shaded 64 at a time: a wave of C3's gem scene that held glass and wall vertices ran the two BSDFs one
    // after the other, each with the other's lanes idle (lane utilisation 0.47).  Same order-free argument as above; one more read of the hit's material word per vertex.
    constexpr bool kSort = PT_SHADE_SORT && FUSE_TRAV == PT_NO_FUSE && (FORM == PT_SHADE_FULL || FORM == PT_SHADE_NO_ENV);
    constexpr uint32_t kLists = kSort ? 2u : 1u;
    __shared__ uint32_t later[(kSplit || kSort) ? kBlock * 2 * kLists : 1];   // per wave and list: 128 item indices (fewer than 64 left over + at most 64 new)
    uint32_t* my_later = later + ((kSplit || kSort) ? (threadIdx.x >> 6) * 128u * kLists : 0u);
    uint32_t* my_later2 = my_later + (kSort ? 128u : 0u);
    uint32_t later_count = 0, later2_count = 0;   // (wave-uniform)
    for (uint32_t r = 0;;) {  // whole waves stay in the loop: the appends are ballots
        uint32_t i = 0;
        bool active = false;
        if ((kSplit || kSort) && (later_count >= 64u || (r == rounds && later_count > 0u))) {
            const uint32_t take = later_count < 64u ? later_count : 64u;
            later_count -= take;
            active = lane_id() < take;
            if (active) i = my_later[later_count + lane_id()];
        } else if (kSort && (later2_count >= 64u || (r == rounds && later2_count > 0u))) {
            const uint32_t take = later2_count < 64u ? later2_count : 64u;
            later2_count -= take;
            active = lane_id() < take;
            if (active) i = my_later2[later2_count + lane_id()];
        } else if (r < rounds) {
            const uint32_t j = r * blockDim.x + threadIdx.x;
            ++r;
            active = j < n; i = base + j;
            if (kSplit || kSort) {
                // (FULL: a path that left the scene is shaded now; NO_ENV: every vertex goes through a list)
                const bool surface = active && (kSplit ? qf(hits, HS_T, i) >= 0.0f : true);
                bool second = false;
                if (kSort && surface && (kSplit || qf(hits, HS_T, i) >= 0.0f)) {
                    const uint32_t mat = qu(hits, HS_MAT, i);
                    if (PT_MATERIAL_TAG(mat) != PT_TAG_LIGHT) { const uint32_t kind = bu(s, material_record(s, mat) + PT_MAT_KIND); second = kind == PT_MATERIAL_GGX || kind == PT_MATERIAL_PASSTHROUGH; }
                }
                const unsigned long long m = __ballot(surface && !second);
                if (surface && !second) my_later[later_count + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull))] = i;
                later_count += (uint32_t)__popcll(m);
                if (kSort) {
                    const unsigned long long m2 = __ballot(second);
                    if (second) my_later2[later2_count + (uint32_t)__popcll(m2 & ((1ull << lane_id()) - 1ull))] = i;
                    later2_count += (uint32_t)__popcll(m2);
                }
                active = active && !surface;
                __builtin_amdgcn_wave_barrier();   // (the lists are the wave's own: its writes are in LDS before any of its lanes reads them)
                if (!kSplit) continue;   // (nothing to shade in this turn)
            }
        } else break;
        PathVertexT<NL> pv; Hit hit; hit.valid = false;
        bool wants_item = false;
        static_assert(FUSE_TRAV == PT_NO_FUSE || FORM != PT_SHADE_FULL, "the fused form has no hit queue to sort by");
        if (active) {
            if (FUSE_TRAV != PT_NO_FUSE) {   // the ray alone during the search; the rest of the path record after it
                const F3 o = f3(qf(paths_in, PS_OX, i), qf(paths_in, PS_OY, i), qf(paths_in, PS_OZ, i));
                const F3 d = f3(qf(paths_in, PS_DX, i), qf(paths_in, PS_DY, i), qf(paths_in, PS_DZ, i));
                world_hit<FUSE_TRAV == PT_NO_FUSE ? PT_TRAV_ANY : FUSE_TRAV>(s, o, d, &hit);
                pv = load_path<NL>(paths_in, i, bounce == 0u);
            } else {
                // (the path record first: its sixteen loads are in flight while the hit record's first word — which decides whether the rest
                // is read at all — comes back; the other order leaves that latency exposed: k_shade 2506 -> 2893 us on C2)
                // (the camera vertex' lean record, pt_stages.h — not in the NO_ENV form: the branch costs that form, at its register budget, 7 %: C3 k_shade 2377 -> 2540 us,
                // so a render that takes it has k_generate write the full record, RenderParams::camera_record)
                pv = load_path<NL>(paths_in, i, FORM != PT_SHADE_NO_ENV && bounce == 0u);
                hit = load_hit<PT_SHADE_EAGER && FORM == PT_SHADE_LEAN>(hits, i);
            }
            if (!(LACKS & PT_SCENE_NO_CERTS)) pv.slot &= ~PT_PATH_INSIDE_MARK;   // (the forms of scenes with certificates: the previous vertex' mark for the closest-hit kernel)
            wants_item = shade_wants_item(s, rp, hit);
        }
        // reserve the light-sample item first, so its rays stream straight from registers to the queue
        uint32_t ipos = base + shared_append(wants_item, &lds_counts[1]);
        // (the list of live items: built by the lean form — the one the BASELINE Cornell configurations take — when the render's light-sample kernel walks it;
        // in the other forms the three registers it holds across the vertex code cost more scratch than the list saves: C3 k_shade 2347 -> 2455 us, measured)
        constexpr bool kLiveList = FORM == PT_SHADE_LEAN;
        bool item_lives = false;   // some ray of the item has a non-zero factor
        ShadeOutT<NL> out;
        out.survives = false; out.has_item = false; out.vertex_pushed = false; out.env_hit = false; out.shadow_count = 0; out.add_energy = false; out.env_mask = 0;
        if (active) {
            uint32_t pixel = pixels[pv.slot % rp.chunk_pixels];
            out = stage_shade<NL, FORM == PT_SHADE_FULL, FORM != PT_SHADE_LEAN>(s, rp, bounce, pv, hit, pixel, [&](uint32_t l, const ShadowRayT<NL>& ray) { store_shadow_ray<NL>(shadow, ipos, l, ray); if (kLiveList) item_lives = item_lives || ray_is_live<NL>(ray); });
            if (wants_item) {
                float lam[NL]; lam[0] = pv.lambda;
                if (NL > 1) hero_lambdas<NL>(rp, pv.lambda, lam);
                qsu(shadow, Layout<NL>::sh_slot, ipos, pv.slot); qsu(shadow, Layout<NL>::sh_flags, ipos, out.env_mask);
                for (int k = 0; k < NL; ++k) qsf(shadow, Layout<NL>::sh_lambda + k, ipos, lam[k]);
                if (!out.has_item) clear_shadow_item<NL>(shadow, ipos, rp.light_samples);  // vertex dropped (NaN pdf, utils.rs:261-263)
            }
            if (out.add_energy) for (int k = 0; k < NL; ++k) energy[(size_t)k * rp.energy_stride + pv.slot] += out.energy_add[k];
        }
        uint32_t pos;
        if (kLiveList && rp.live_list) {   // the list of the segment's live items (Layout::shadow_live_field), appended with the surviving paths
            uint32_t lpos;
            shared_append2(out.survives, &lds_counts[0], item_lives, &lds_counts[2], &pos, &lpos);
            pos += base;
            if (item_lives) qsu(shadow, Layout<NL>::shadow_live_field(rp.light_samples), base + lpos, ipos);
        } else pos = base + shared_append(out.survives, &lds_counts[0]);
        if (out.survives) store_path<NL>(paths_out, pos, out.next);
        st_vertices += out.vertex_pushed ? 1u : 0u; st_env += out.env_hit ? 1u : 0u; st_shadow += out.shadow_count;
#ifdef PT_TIMELINE
        tl_work += out.vertex_pushed ? 1u : 0u;   // (timeline: surface vertices shaded)
#endif
    }
    PT_TL_END(n);
    // workgroup totals -> this workgroup's statistics record
    st_vertices = wave_reduce_add(st_vertices); st_shadow = wave_reduce_add(st_shadow); st_env = wave_reduce_add(st_env);
    if (lane_id() == 0) { atomicAdd(&lds_counts[4], st_vertices); atomicAdd(&lds_counts[5], st_shadow); atomicAdd(&lds_counts[6], st_env); }
    __syncthreads();
    if (threadIdx.x == 0) {
        count_out[blockIdx.x] = lds_counts[0]; shadow_count[blockIdx.x] = lds_counts[1]; shadow_count[gridDim.x + blockIdx.x] = lds_counts[2];
        unsigned long long* bs = block_stats + (size_t)blockIdx.x * BS_FIELDS;
        bs[BS_VERTICES] += lds_counts[4]; bs[BS_SHADOW_RAYS] += lds_counts[5]; bs[BS_ENV_HITS] += lds_counts[6];
        bs[BS_SEGMENTS] += n;
        bs[BS_ITEMS] += lds_counts[1];
    }
}

// The medium-aware walk's vertex kernel (stage_shade_medium; pt_render_desc::medium_aware): k_shade with the tracked mediums and the
// "previous vertex was a medium vertex" flag carried in two more fields of the path record.
template <int USE_LDS>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PT_SHADE_MEDIUM_WAVES)))
k_shade_medium(const uint32_t* __restrict__ blob, uint32_t blob_words, const float* __restrict__ tex, RenderParams rp, uint32_t bounce, const uint32_t* __restrict__ pixels,
               Queue paths_in, Queue hits, Queue paths_out, Queue shadow, float* __restrict__ energy, uint32_t seg_cap, const uint32_t* __restrict__ count_in,
               uint32_t* __restrict__ count_out, uint32_t* __restrict__ shadow_count, unsigned long long* __restrict__ block_stats) {
    extern __shared__ __align__(16) uint32_t lds[];
    __shared__ uint32_t lds_counts[16];
    if (PT_EMPTY_SEGMENT_EXIT && count_in[blockIdx.x] == 0u) {
        if (threadIdx.x == 0) { count_out[blockIdx.x] = 0u; shadow_count[blockIdx.x] = 0u; shadow_count[gridDim.x + blockIdx.x] = 0u; }
        return;
    }
    if (threadIdx.x < 16) lds_counts[threadIdx.x] = 0;
    SceneView s = stage_scene<USE_LDS>(blob, blob_words, tex, lds);
    if (USE_LDS == PT_LDS_NONE) __syncthreads();
    const uint32_t base = blockIdx.x * seg_cap, n = count_in[blockIdx.x];
    uint32_t st_vertices = 0, st_shadow = 0, st_env = 0, st_drops = 0;
    const uint32_t rounds = (n + blockDim.x - 1) / blockDim.x;
    // (round 5) The split by vertex kind that k_shade's FULL form has, behind the free-flight step: a wave first takes its 64 items through stage_medium_flight —
    // environment vertices and medium vertices (a phase-function sample) are finished there — and keeps the SURFACE vertices (material, light samples: the expensive
    // kind) with the throughput their segment left in a list of its own, shading them 64 at a time whenever the list holds that many, and what is left at the end.
    // The kinds shared waves at a lane utilisation of 0.36 (profiles/r5z_F4_summary.json).  No vertex' result depends on when it is shaded; a light-sample item is
    // reserved when its surface vertex is shaded (a vertex that scattered in a medium first used to reserve one and clear it).
    // (PT_MEDIUM_SPLIT 2 — built, bit-identical, slower, not the default: a second list in front of it.  Whether a path tracks a medium at all is in its record: a fresh round only SORTS its 64 items —
    // an environment vertex is finished there, a hit without a tracked medium goes to the surface list with its throughput as it is (the attenuation of no medium is
    // the factor 1.0 twice: the same bits), a hit behind tracked mediums to the flight list — and the free flights, too, run for whole waves: stage_medium_flight for 64
    // of those, its medium vertices finished, its surface vertices passed on.  A wave drains the surface list first, then the flight list, then takes fresh items: either
    // list holds fewer than 64 entries when something is added to it, and a step adds at most 64.)
    constexpr bool kSplit = PT_MEDIUM_SPLIT != 0, kSort = PT_MEDIUM_SPLIT == 2;
    __shared__ uint32_t later[kSplit ? kBlock * 2 : 1];
    __shared__ float later_beta[kSplit ? kBlock * 2 : 1];
    __shared__ uint32_t flights[kSort ? kBlock * 2 : 1];
    uint32_t* my_later = later + (kSplit ? (threadIdx.x >> 6) * 128u : 0u);
    float* my_beta = later_beta + (kSplit ? (threadIdx.x >> 6) * 128u : 0u);
    uint32_t* my_flights = flights + (kSort ? (threadIdx.x >> 6) * 128u : 0u);
    uint32_t later_count = 0, flight_count = 0;   // (wave-uniform)
    enum { FRESH, FLIGHT, SURFACE };
    for (uint32_t r = 0;;) {
        uint32_t i = 0;
        bool active = false;
        int mode = FRESH;   // (wave-uniform)
        float beta_in = 0.0f;
        if (kSplit && (later_count >= 64u || (r == rounds && flight_count == 0u && later_count > 0u))) {
            const uint32_t take = later_count < 64u ? later_count : 64u;
            later_count -= take;
            active = lane_id() < take; mode = SURFACE;
            if (active) { i = my_later[later_count + lane_id()]; beta_in = my_beta[later_count + lane_id()]; }
        } else if (kSort && (flight_count >= 64u || (r == rounds && flight_count > 0u))) {
            const uint32_t take = flight_count < 64u ? flight_count : 64u;
            flight_count -= take;
            active = lane_id() < take; mode = FLIGHT;
            if (active) i = my_flights[flight_count + lane_id()];
        } else if (r < rounds) {
            const uint32_t j = r * blockDim.x + threadIdx.x;
            ++r;
            active = j < n; i = base + j;
        } else break;
        PathVertexT<1> pv; Hit hit; hit.valid = false;
        MediumState ms{0u, 0u}, ms_next{0u, 0u};
        ShadeOutT<1> out;
        shade_out_clear(&out);
        auto append = [&](bool flag, uint32_t* list, uint32_t* count, float* betas, float beta) {   // the wave's own list: one ballot
            const unsigned long long m = __ballot(flag);
            if (flag) { const uint32_t e = *count + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull)); list[e] = i; if (betas != nullptr) betas[e] = beta; }
            *count += (uint32_t)__popcll(m);
        };
        auto load_item = [&] {
            pv = load_path<1>(paths_in, i, rp.camera_record != 0u && bounce == 0u);   // (the camera vertex' lean record, pt_stages.h)
            hit = load_hit(hits, i);
            if (bounce != 0) { ms.mediums = qu(paths_in, PS_MEDIUMS, i); ms.prev_medium = qu(paths_in, PS_PREV_MEDIUM, i); }   // (the camera starts in vacuum)
        };
        if (kSort && mode == FRESH) {
            // sort: the first word of the hit record and the tracked mediums say which kind the item is
            const bool hits_something = active && qf(hits, HS_T, i) >= 0.0f;
            const bool tracks = hits_something && bounce != 0 && qu(paths_in, PS_MEDIUMS, i) != 0u;
            const bool leaves = active && !hits_something;
            if (leaves) {   // the environment vertex
                load_item();
                float beta;
                stage_medium_flight(s, rp, bounce, pv, hit, pixels[pv.slot % rp.chunk_pixels], ms, &ms_next, &out, &beta);
                if (out.add_energy) energy[pv.slot] += out.energy_add[0];
            }
            append(tracks, my_flights, &flight_count, nullptr, 0.0f);
            append(hits_something && !tracks, my_later, &later_count, my_beta, hits_something && !tracks ? qf(paths_in, PS_BETA, i) : 0.0f);
            __builtin_amdgcn_wave_barrier();   // (the lists are the wave's own: their writes are in LDS before any of its lanes reads them)
        } else if (kSplit && mode != SURFACE) {
            // the free flights (PT_MEDIUM_SPLIT 1: of a fresh round's items; 2: of 64 items that track a medium)
            if (active) load_item();
            bool surface = false; float beta = 0.0f;
            if (active) surface = stage_medium_flight(s, rp, bounce, pv, hit, pixels[pv.slot % rp.chunk_pixels], ms, &ms_next, &out, &beta);
            append(surface, my_later, &later_count, my_beta, beta);
            __builtin_amdgcn_wave_barrier();
            if (active && out.add_energy) energy[pv.slot] += out.energy_add[0];
        } else {
            // the surface vertices of the list (or, without the split, every vertex in one step)
            if (active) load_item();
            const bool wants_item = active && shade_medium_wants_item(s, rp, hit, ms);
            const uint32_t ipos = base + shared_append(wants_item, &lds_counts[1]);
            if (active) {
                const uint32_t pixel = pixels[pv.slot % rp.chunk_pixels];
                auto sink = [&](uint32_t l, const ShadowRayT<1>& ray) { store_shadow_ray<1>(shadow, ipos, l, ray); };
                if (kSplit) out = stage_medium_surface(s, rp, bounce, pv, hit, pixel, ms, beta_in, &ms_next, sink);
                else out = stage_shade_medium(s, rp, bounce, pv, hit, pixel, ms, &ms_next, sink);
                if (wants_item) {
                    qsu(shadow, Layout<1>::sh_slot, ipos, pv.slot); qsu(shadow, Layout<1>::sh_flags, ipos, out.env_mask);
                    qsf(shadow, Layout<1>::sh_lambda, ipos, pv.lambda);
                    if (!out.has_item) clear_shadow_item<1>(shadow, ipos, rp.light_samples);   // a surface vertex that was never pushed (or, without the split, a medium vertex)
                }
                if (out.add_energy) energy[pv.slot] += out.energy_add[0];
            }
        }
        const uint32_t pos = base + shared_append(out.survives, &lds_counts[0]);
        if (out.survives) { store_path<1>(paths_out, pos, out.next); qsu(paths_out, PS_MEDIUMS, pos, ms_next.mediums); qsu(paths_out, PS_PREV_MEDIUM, pos, ms_next.prev_medium); }
        st_vertices += out.vertex_pushed ? 1u : 0u; st_env += out.env_hit ? 1u : 0u; st_shadow += out.shadow_count; st_drops += ms_next.dropped;
    }
    st_vertices = wave_reduce_add(st_vertices); st_shadow = wave_reduce_add(st_shadow); st_env = wave_reduce_add(st_env); st_drops = wave_reduce_add(st_drops);
    if (lane_id() == 0) { atomicAdd(&lds_counts[4], st_vertices); atomicAdd(&lds_counts[5], st_shadow); atomicAdd(&lds_counts[6], st_env); atomicAdd(&lds_counts[7], st_drops); }
    __syncthreads();
    if (threadIdx.x == 0) {
        count_out[blockIdx.x] = lds_counts[0]; shadow_count[blockIdx.x] = lds_counts[1];
        unsigned long long* bs = block_stats + (size_t)blockIdx.x * BS_FIELDS;
        bs[BS_VERTICES] += lds_counts[4]; bs[BS_SHADOW_RAYS] += lds_counts[5]; bs[BS_ENV_HITS] += lds_counts[6];
        bs[BS_SEGMENTS] += n;
        bs[BS_ITEMS] += lds_counts[1];
        bs[BS_MEDIUM_DROPS] += lds_counts[7];
    }
}

constexpr uint32_t kShadowListed = 0x80000000u;   // (a flag in k_shadow's seg_cap argument: segments are far smaller)
template <int USE_LDS, int NL, int TRAV, bool ENV = true, uint32_t LACKS = 0u>
__global__ void __launch_bounds__(kBlock) PT_SHADOW_OCC k_shadow(const uint32_t* __restrict__ blob, uint32_t blob_words, const float* __restrict__ tex,
                                                  uint32_t light_samples, Queue shadow, float* __restrict__ energy, uint32_t energy_stride,
                                                  uint32_t seg_cap, const uint32_t* __restrict__ count_in) {
    extern __shared__ __align__(16) uint32_t lds[];
    if (PT_EMPTY_SEGMENT_EXIT && count_in[((seg_cap & kShadowListed) != 0u ? gridDim.x : 0u) + blockIdx.x] == 0u) return;
    SceneView s = stage_scene<USE_LDS, LACKS>(blob, blob_words, tex, lds);
    // `seg_cap` with its top bit set (kShadowListed; the engine sets it when the vertex kernel built the list): the segment's LIVE items, through their list
    // (Layout::shadow_live_field; the count behind the items' counts) — an item without a live ray is not read.  Otherwise every item of the segment.
    const bool listed = (seg_cap & kShadowListed) != 0u;
    seg_cap &= ~kShadowListed;
    uint32_t base = blockIdx.x * seg_cap, n = count_in[(listed ? gridDim.x : 0u) + blockIdx.x];
    for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
        const uint32_t item = listed ? qu(shadow, Layout<NL>::shadow_live_field(light_samples), base + j) : base + j;
        stage_shadow_item<NL, TRAV, ENV>(s, light_samples, shadow, item, energy, energy_stride);
    }
}

#ifdef PT_EXPERIMENTS
// MEASURED, NOT FASTER (profiles/r4_experiments.md: C2 k_shadow 3279 us, this kernel 3299 at six waves and 3501 at five; C5 2278 / 2418) — a measurement build's kernel.
// k_shadow with the rays that are traced at all compacted per wave (round 4).  Of C2's light-sample rays 12 % are dead (factor 0: the sample lies below
// its surface's horizon) and 19 % of the rest meet no light (a ceiling vertex' ray, offset off the ceiling, starts below the lamp it aims at): in k_shadow,
// which takes "ray l of every item" per step, their lanes sit idle through the whole search — box tests at a lane utilisation of 0.65 where k_extend's run at
// 0.93 (tools/phase_util.sh).  Here a wave first LISTS the rays that search: per sample number, the lanes read their item's ray, run the light pre-pass
// (shadow_light_bound: the nearest light hit bounds the search) and append (item, l, stop, kind | bound | bounding light) to the wave's list in LDS with one
// ballot; whenever the list holds 64 rays — and at the end — the wave traces 64 of them, every lane busy, and leaves each ray's contribution where its factor
// was; the item's rays are summed in order at the end as before (pt.rs:349-392).  k_shadow_parked's listing, one step further (it lists before the light
// pre-pass) and without the parking.  A ray's own search is untouched: same bound, same stop rule, same known light.
template <int USE_LDS, int NL, int TRAV, bool ENV = true, uint32_t LACKS = 0u>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PT_SHADOW_LIVE_WAVES))) k_shadow_live(const uint32_t* __restrict__ blob, uint32_t blob_words, const float* __restrict__ tex,
                                                                      uint32_t light_samples, Queue shadow, float* __restrict__ energy, uint32_t energy_stride,
                                                                      uint32_t seg_cap, const uint32_t* __restrict__ count_in, uint32_t live_off) {
    extern __shared__ __align__(16) uint32_t lds[];
    SceneView s = stage_scene<USE_LDS, LACKS>(blob, blob_words, tex, lds);
    const uint32_t cap = live_cap(light_samples);   // entries per wave; field f of entry e at live[f * cap + e]
    const uint32_t wave = PT_UNIFORM(threadIdx.x >> 6);   // (a scalar: with fresh_lane_id below, threadIdx.x need not stay in a register across the rounds)
    uint32_t* live = lds + live_off + wave * cap * 3u;
    const uint32_t base = blockIdx.x * seg_cap, n = PT_UNIFORM(count_in[blockIdx.x]);   // (kBlock, not blockDim.x, below: a division by a run-time value runs on the VALU and takes the loop's scalars with it)
    const uint32_t rounds = (n + (uint32_t)kBlock - 1) / (uint32_t)kBlock;
    uint32_t live_count = 0;   // (wave-uniform)
    for (uint32_t r = 0;;) {
        const bool flush = r == rounds && live_count > 0u;
        if (!(live_count >= 64u || flush)) {
            if (r == rounds) break;
            const uint32_t j = r * (uint32_t)kBlock + (wave << 6 | fresh_lane_id()), item = base + j;
            ++r;
            const uint32_t flags = (ENV && j < n) ? qu(shadow, Layout<NL>::sh_flags, item) : 0u;
            for (uint32_t l = 0; l < light_samples; ++l) {
                bool searches = false;
                float bound = PT_INF; int stop = PT_STOP_NONE;
                uint32_t light = 0xffffffffu;
                const bool env = ENV && ((flags >> l) & 1u) != 0u;
                ShadowRayT<NL> ray;
                if (j < n && load_shadow_ray<NL>(shadow, item, l, &ray)) {
                    if (env) { stop = shadow_env_stop(s); searches = true; }
                    else if (shadow_light_bound(s, ray.o, ray.d, &bound, &stop, &light)) searches = true;
                    else for (int k = 0; k < NL; ++k) qsf(shadow, Layout<NL>::sh_head + l * Layout<NL>::sr_fields + SR_FACTOR + k, item, 0.0f);   // (no light on its line: it adds nothing)
                }
                const unsigned long long m = __ballot(searches);
                if (searches) {
                    const uint32_t e = live_count + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull));
                    live[e] = j | l << 24 | (env ? 1u << 31 : 0u);
                    live[cap + e] = pt_f2u(bound); live[2u * cap + e] = light;
                }
                live_count += (uint32_t)__popcll(m);
            }
            __builtin_amdgcn_wave_barrier();   // (the list is the wave's own: its writes are in LDS before any of its lanes reads them)
            if (r < rounds || live_count >= 64u) continue;
        }
        const uint32_t take = live_count < 64u ? live_count : 64u;
        live_count -= take;
        if (lane_id() < take) {
            const uint32_t e = live_count + lane_id(), w = live[e], j = w & 0xffffffu, l = (w >> 24) & 7u, item = base + j;
            const float bound = pt_u2f(live[cap + e]);
            const uint32_t light = live[2u * cap + e];
            const bool env = ENV && (w >> 31) != 0u;
            // one address per ray — the first field of ray l of the item — held across the search; every field after it is an immediate offset (the sample number
            // differs from lane to lane here: addresses formed field by field from (item, l) are six 64-bit values the compiler keeps alive through phase 3)
            uint32_t* const rb = shadow.base + qindex(shadow, Layout<NL>::sh_head + l * Layout<NL>::sr_fields, item);
            auto rf = [&](uint32_t field) { return pt_u2f(rb[(size_t)field << 6]); };
            const F3 o = f3(rf(SR_OX), rf(SR_OY), rf(SR_OZ));
            const F3 d = f3(rf(SR_DX), rf(SR_DY), rf(SR_DZ));
            // (one wavelength: the factor and the wavelength come with the ray — one exposed load latency per step instead of a second one behind the search)
            const float factor0 = NL == 1 ? rf(SR_FACTOR) : 0.0f, lambda0 = NL == 1 ? qf(shadow, Layout<NL>::sh_lambda, item) : 0.0f;
            Hit sh;
            // (the stop rule is the scene's, not the ray's — what shadow_light_bound / shadow_env_stop said when the ray was listed: wave-uniform for the search)
            const int stop = env ? shadow_env_stop(s) : shadow_light_stop(s);
            const bool hit = world_hit<TRAV, true>(s, o, d, &sh, bound, stop, light, bound);
            float c[NL];
            shadow_ray_contribution<NL>(s, [&](int k) { return NL == 1 ? lambda0 : qf(shadow, Layout<NL>::sh_lambda + (uint32_t)k, item); },
                                        [&](int k) { return NL == 1 ? factor0 : rf(SR_FACTOR + (uint32_t)k); }, d, env, hit, sh, c);
            for (int k = 0; k < NL; ++k) rb[(size_t)(SR_FACTOR + k) << 6] = pt_f2u(c[k]);
        }
        __builtin_amdgcn_wave_barrier();
    }
    __threadfence_block();   // (an item's rays were traced by the wave that owns the item: its sums below read what its own lanes wrote)
    for (uint32_t r = 0; r < rounds; ++r) {  // pt.rs:349-392, 596: the item's rays summed in order, divided by L
        const uint32_t j = r * (uint32_t)kBlock + (wave << 6 | fresh_lane_id());
        if (j >= n) continue;
        const uint32_t item = base + j, slot = qu(shadow, Layout<NL>::sh_slot, item);
        float lc[NL];
        for (int k = 0; k < NL; ++k) lc[k] = 0.0f;
        for (uint32_t l = 0; l < light_samples; ++l)
            for (int k = 0; k < NL; ++k) lc[k] += qf(shadow, Layout<NL>::sh_head + l * Layout<NL>::sr_fields + SR_FACTOR + k, item);
        for (int k = 0; k < NL; ++k) energy[(size_t)k * energy_stride + slot] += lc[k] / (float)light_samples;
    }
}

#endif   // PT_EXPERIMENTS (k_shadow_live)

// ------------------------------------------------------------------------------------------------ pooled traversal
// Pure sweep scenes (every leaf in the table: the Cornell box of C2 / C5).  Phases 1 and 2 per lane as in k_extend / k_shadow; phase
// 3 for the wave's 64 rays together (sweep_run_pooled, pt_device.h): the candidate triangles of all of them are pooled in LDS and
// tested 64 at a time, then every lane replays the acceptance of its own candidates in leaf order.  Bit-identical results; the
// per-lane loop it replaces ran at ~0.4 lane utilisation (6.5 iterations per wave for 2.4 candidates per ray).
#ifndef PT_POOL_WAVES
#define PT_POOL_WAVES 5
#endif
#define PT_POOL_OCC __attribute__((amdgpu_waves_per_eu(PT_POOL_WAVES)))
constexpr uint32_t kPoolBytes = (kBlock / 64) * PT_POOL_WORDS * 4u;

template <int USE_LDS>
__global__ void __launch_bounds__(kBlock) PT_POOL_OCC k_extend_pooled(const uint32_t* __restrict__ blob, uint32_t blob_words, const float* __restrict__ tex,
                                                                     Queue paths, Queue hits, uint32_t seg_cap, const uint32_t* __restrict__ count_in) {
    extern __shared__ __align__(16) uint32_t lds[];
    __shared__ __align__(16) uint32_t pool_all[(kBlock / 64) * PT_POOL_WORDS];
    SceneView s = stage_scene<USE_LDS>(blob, blob_words, tex, lds);
    uint32_t* ws = pool_all + (threadIdx.x >> 6) * PT_POOL_WORDS;
    const uint32_t base = blockIdx.x * seg_cap, n = count_in[blockIdx.x];
    const uint32_t rounds = (n + blockDim.x - 1) / blockDim.x;
    for (uint32_t r = 0; r < rounds; ++r) {  // whole waves stay in the loop
        const uint32_t j = r * blockDim.x + threadIdx.x, i = base + j;
        const bool active = j < n;
        F3 o = f3(0, 0, 0), d = f3(0, 0, 1);
        SweepState st;
        sweep_state_init(st, 0);
        if (active) {
            o = f3(qf(paths, PS_OX, i), qf(paths, PS_OY, i), qf(paths, PS_OZ, i));
            d = f3(qf(paths, PS_DX, i), qf(paths, PS_DY, i), qf(paths, PS_DZ, i));
            st.hit = sweep_masks(s, o, d, PT_INF);
        }
        const TriRay wtr = tri_ray_prepare(o, d);
        sweep_run_pooled(s, ws, o, d, wtr, st);
        if (active) { Hit h; sweep_finish(s, o, d, st, &h); store_hit(hits, i, h); }
    }
}

template <int USE_LDS, int NL, bool ENV>
__global__ void __launch_bounds__(kBlock) PT_POOL_OCC k_shadow_pooled(const uint32_t* __restrict__ blob, uint32_t blob_words, const float* __restrict__ tex,
                                                                     uint32_t light_samples, Queue shadow, float* __restrict__ energy, uint32_t energy_stride,
                                                                     uint32_t seg_cap, const uint32_t* __restrict__ count_in) {
    extern __shared__ __align__(16) uint32_t lds[];
    __shared__ __align__(16) uint32_t pool_all[(kBlock / 64) * PT_POOL_WORDS];
    SceneView s = stage_scene<USE_LDS>(blob, blob_words, tex, lds);
    uint32_t* ws = pool_all + (threadIdx.x >> 6) * PT_POOL_WORDS;
    const uint32_t base = blockIdx.x * seg_cap, n = count_in[blockIdx.x];
    const uint32_t rounds = (n + blockDim.x - 1) / blockDim.x;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t j = r * blockDim.x + threadIdx.x, item = base + j;
        const bool active = j < n;
        const uint32_t flags = (ENV && active) ? qu(shadow, Layout<NL>::sh_flags, item) : 0u;
        float lambda[NL], lc[NL];
        for (int k = 0; k < NL; ++k) { lambda[k] = active ? qf(shadow, Layout<NL>::sh_lambda + k, item) : 0.0f; lc[k] = 0.0f; }
        for (uint32_t l = 0; l < light_samples; ++l) {   // stage_shadow_item, the rays of the wave's 64 items side by side
            ShadowRayT<NL> ray; ray.o = f3(0, 0, 0); ray.d = f3(0, 0, 1);
            bool live = active && load_shadow_ray<NL>(shadow, item, l, &ray);
            const bool env = ENV && ((flags >> l) & 1u) != 0u;
            SweepState st;
            sweep_state_init(st, 0);
            if (live) {
                float bound = PT_INF; int stop = PT_STOP_NONE;
                if (!env) live = shadow_light_bound(s, ray.o, ray.d, &bound, &stop);  // no light on the ray: nothing to trace
                if (live) st.hit = sweep_masks(s, ray.o, ray.d, bound);
            }
            const TriRay wtr = tri_ray_prepare(ray.o, ray.d);
            sweep_run_pooled(s, ws, ray.o, ray.d, wtr, st);
            if (live) {
                float c[NL];
                Hit sh; sh.valid = false;
                const bool hit = st.best_inst != 0xffffffffu;
                // only a light (an analytic shape on this path) or nothing at all contributes: the record of an occluder is never read
                const bool wanted = hit && !env && sweep_best_is_light(s, st);
                if (wanted) sweep_finish(s, ray.o, ray.d, st, &sh);
                if (hit && !wanted) { sh.valid = true; sh.material = PT_MATERIAL_ID(PT_TAG_MATERIAL, 0); }
                shadow_ray_contribution<NL>(s, [&](int k) { return pl_get<NL>(lambda, k); }, ray, env, hit, sh, c);
                for (int k = 0; k < NL; ++k) lc[k] += c[k];
            }
        }
        if (active) {
            const uint32_t slot = qu(shadow, Layout<NL>::sh_slot, item);
            for (int k = 0; k < NL; ++k) energy[(size_t)k * energy_stride + slot] += lc[k] / (float)light_samples;
        }
    }
}

#ifdef PT_EXPERIMENTS
// Measurement only (make EXTRA=-DPT_EXPERIMENTS, PT_AMD_EXP=<bits>; tools/phase_costs.sh): k_extend with parts left out, launched in
// front of the real kernel on the same input (its output is overwritten), so that the stage time grows by the cost of what is left in.
// bits: 1 pooled form, 2 no phase 3, 4 no hit record, 8 / 0x10 / 0x20 / 0x40 pooled: no triangle chunks / replay / winner / candidate lists
template <int EXP>
__global__ void __launch_bounds__(kBlock) PT_POOL_OCC k_extend_exp(const uint32_t* __restrict__ blob, uint32_t blob_words, const float* __restrict__ tex,
                                                                  Queue paths, Queue hits, uint32_t seg_cap, const uint32_t* __restrict__ count_in) {
    extern __shared__ __align__(16) uint32_t lds[];
    __shared__ __align__(16) uint32_t pool_all[(EXP & 1) ? (kBlock / 64) * PT_POOL_WORDS : 4];
    SceneView s = stage_scene<PT_LDS_ALL>(blob, blob_words, tex, lds);
    uint32_t* ws = pool_all + ((EXP & 1) ? (threadIdx.x >> 6) * PT_POOL_WORDS : 0);
    const uint32_t base = blockIdx.x * seg_cap, n = count_in[blockIdx.x];
    const uint32_t rounds = (n + blockDim.x - 1) / blockDim.x;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t j = r * blockDim.x + threadIdx.x, i = base + j;
        const bool active = j < n;
        F3 o = f3(0, 0, 0), d = f3(0, 0, 1);
        SweepState st;
        sweep_state_init(st, 0);
        if (active) {
            o = f3(qf(paths, PS_OX, i), qf(paths, PS_OY, i), qf(paths, PS_OZ, i));
            d = f3(qf(paths, PS_DX, i), qf(paths, PS_DY, i), qf(paths, PS_DZ, i));
            st.hit = sweep_masks(s, o, d, PT_INF);
        }
        if (EXP & 2) { if (active) { qsu(hits, HS_T, i, (uint32_t)st.hit); qsu(hits, HS_PX, i, (uint32_t)(st.hit >> 32)); } continue; }
        const TriRay wtr = tri_ray_prepare(o, d);
        if (EXP & 1) sweep_run_pooled<EXP>(s, ws, o, d, wtr, st);
        else if (active) sweep_run<false>(s, o, d, wtr, PT_INF, PT_STOP_NONE, st, false);
        if (active) {
            if (EXP & 4) { qsf(hits, HS_T, i, st.closest); qsu(hits, HS_PX, i, st.best_inst); }
            else { Hit h; sweep_finish(s, o, d, st, &h); store_hit(hits, i, h); }
        }
    }
}
// k_shadow (lane form, no environment rays) with parts left out; its results go to the hit queue, which is idle at that time.
// bits: 2 stop after the masks, 8 stop after the light bound, 4 no hit record / contribution
template <int EXP>
__global__ void __launch_bounds__(kBlock) PT_POOL_OCC k_shadow_exp(const uint32_t* __restrict__ blob, uint32_t blob_words, const float* __restrict__ tex,
                                                                  uint32_t light_samples, Queue shadow, Queue sink, uint32_t seg_cap, const uint32_t* __restrict__ count_in) {
    extern __shared__ __align__(16) uint32_t lds[];
    SceneView s = stage_scene<PT_LDS_ALL>(blob, blob_words, tex, lds);
    uint32_t base = blockIdx.x * seg_cap, n = count_in[blockIdx.x];
    for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
        const uint32_t item = base + j;
        float lambda[1] = {qf(shadow, Layout<1>::sh_lambda, item)}, lc = 0.0f;
        for (uint32_t l = 0; l < light_samples; ++l) {
            ShadowRayT<1> ray;
            if (!load_shadow_ray<1>(shadow, item, l, &ray)) continue;
            float bound = PT_INF; int stop = PT_STOP_NONE;
            uint32_t light = 0xffffffffu;
            if (!shadow_light_bound(s, ray.o, ray.d, &bound, &stop, &light)) continue;
            if (EXP & 8) { lc += bound; continue; }
            SweepState st;
            sweep_state_init(st, sweep_masks(s, ray.o, ray.d, bound));
            if (EXP & 2) { lc += (float)(uint32_t)st.hit + (float)(uint32_t)(st.hit >> 32); continue; }
            const TriRay wtr = tri_ray_prepare(ray.o, ray.d);
            sweep_run<false>(s, ray.o, ray.d, wtr, bound, stop, st, false, light, bound);
            if (EXP & 4) { lc += st.closest + (float)st.best_inst; continue; }
            Hit sh; float c[1];
            bool hit = st.best_inst != 0xffffffffu;
            if (hit && !sweep_best_is_light(s, st)) { sh.valid = true; sh.material = PT_MATERIAL_ID(PT_TAG_MATERIAL, 0); }
            else hit = sweep_finish(s, ray.o, ray.d, st, &sh);
            shadow_ray_contribution<1>(s, [&](int) { return lambda[0]; }, ray, false, hit, sh, c);
            lc += c[0];
        }
        qsf(sink, HS_T, item, lc);
    }
}
#endif

// ------------------------------------------------------------------------------------------------ parked traversal
// Scenes whose sweep table holds walked meshes (PT_FLAG_SWEEP_WALKS: a few analytic shapes and small meshes around one or
// more big meshes — the gem in the Cornell room, the monkey under the HDRI).  Most rays never enter a big mesh's box, and
// the ones that do are scattered over the waves, so walking the mesh in line would leave a wave waiting for a handful of
// lanes.  Instead a lane that reaches a walked-mesh bit *parks* its sweep state in the workgroup's scratch region, and
// whenever a wave has 64 rays parked (and at the end) it resumes them together: a full wave, every lane in a mesh walk.  Nothing about a ray's own sequence of tests changes (same leaves, same order, same running closest hit).
#ifndef PT_CONVEX_SKIP
#define PT_CONVEX_SKIP 1   /* the parked kernels honour the mark "cannot hit its instance again" (round 6); 0 = ignore it: the mesh is walked and nothing found */
#endif
enum { PK_ITEM, PK_HIT_LO, PK_HIT_HI, PK_CLOSEST, PK_BEST_INST, PK_BEST_TRIW, PK_T, PK_B0, PK_B1, PK_B2, PK_RAY, PK_BOUND, PK_KIND, PK_CURSOR };
static_assert(PK_CURSOR < kParkFields, "a parked entry's fields");
// REC (round 5: the round-4 verdict's "compacted park record", built instead of priced): which fields an entry carries.  PK_FULL = all fourteen; PK_SEGMENT = a path segment's
// (k_extend_parked: no ray number, bound or kind — constants there): eleven; PK_LIGHT_RAY = a light-sample ray's whose closest hit's barycentrics nobody will read
// (only a light's record is ever built, shadow_ray_contribution, and a light is an analytic shape unless a mesh instance is overridden with one —
// `with_bh`, wave-uniform, says whether this scene has such a mesh): ten.
enum { PK_FULL = 0, PK_SEGMENT = 1, PK_LIGHT_RAY = 2 };
#ifndef PT_PARK_SEGMENT_REC
#define PT_PARK_SEGMENT_REC PK_FULL   /* k_extend_parked's record: PK_SEGMENT measured +0.7 % on C3 (3210 -> 3234 us: the kernel allocates worse), equal on C4 and G1 — not taken */
#endif
#ifndef PT_PARK_COMPACT
#define PT_PARK_COMPACT 1   /* 0: every kernel stores and loads every field (the record of rounds 2-4) */
#endif
template <uint32_t kParkCap = ptk::kParkCap, int REC = PK_FULL>   // (kParkCap: the stride between the fields of a workgroup's entries: 128 per wave)
__device__ __forceinline__ void park_store(uint32_t* pk, uint32_t e, uint32_t item, const SweepState& st, uint32_t ray, float bound, uint32_t kind, uint32_t cursor, bool with_bh = true) {
    pk[PK_ITEM * kParkCap + e] = item; pk[PK_HIT_LO * kParkCap + e] = (uint32_t)st.hit; pk[PK_HIT_HI * kParkCap + e] = (uint32_t)(st.hit >> 32);
    pk[PK_CLOSEST * kParkCap + e] = pt_f2u(st.closest); pk[PK_BEST_INST * kParkCap + e] = st.best_inst; pk[PK_BEST_TRIW * kParkCap + e] = st.best_triw;
    if (!PT_PARK_COMPACT || REC != PK_LIGHT_RAY || with_bh) {
        pk[PK_T * kParkCap + e] = pt_f2u(st.bh.t); pk[PK_B0 * kParkCap + e] = pt_f2u(st.bh.b0); pk[PK_B1 * kParkCap + e] = pt_f2u(st.bh.b1); pk[PK_B2 * kParkCap + e] = pt_f2u(st.bh.b2);
    }
    if (!PT_PARK_COMPACT || REC != PK_SEGMENT) { pk[PK_RAY * kParkCap + e] = ray; pk[PK_BOUND * kParkCap + e] = pt_f2u(bound); pk[PK_KIND * kParkCap + e] = kind; }
    pk[PK_CURSOR * kParkCap + e] = cursor;
}
template <uint32_t kParkCap = ptk::kParkCap, int REC = PK_FULL>
__device__ __forceinline__ void park_load(const uint32_t* pk, uint32_t e, uint32_t* item, SweepState* st, uint32_t* ray, float* bound, uint32_t* kind, uint32_t* cursor, bool with_bh = true) {
    *item = pk[PK_ITEM * kParkCap + e]; st->hit = (uint64_t)pk[PK_HIT_LO * kParkCap + e] | (uint64_t)pk[PK_HIT_HI * kParkCap + e] << 32;
    st->closest = pt_u2f(pk[PK_CLOSEST * kParkCap + e]); st->best_inst = pk[PK_BEST_INST * kParkCap + e]; st->best_triw = pk[PK_BEST_TRIW * kParkCap + e];
    st->bh.t = 0.0f; st->bh.b0 = st->bh.b1 = st->bh.b2 = 0.0f;
    if (!PT_PARK_COMPACT || REC != PK_LIGHT_RAY || with_bh) {
        st->bh.t = pt_u2f(pk[PK_T * kParkCap + e]); st->bh.b0 = pt_u2f(pk[PK_B0 * kParkCap + e]); st->bh.b1 = pt_u2f(pk[PK_B1 * kParkCap + e]); st->bh.b2 = pt_u2f(pk[PK_B2 * kParkCap + e]);
    }
    *ray = 0u; *bound = PT_INF; *kind = 0u;
    if (!PT_PARK_COMPACT || REC != PK_SEGMENT) { *ray = pk[PK_RAY * kParkCap + e]; *bound = pt_u2f(pk[PK_BOUND * kParkCap + e]); *kind = pk[PK_KIND * kParkCap + e]; }
    *cursor = pk[PK_CURSOR * kParkCap + e];
}
// The resume loop shared by both kernels, per WAVE: every wave of the workgroup parks into its own quarter of the scratch
// region (128 entries: fewer than 64 left over + at most 64 new per step) and resumes 64 parked rays at a time — full
// waves — with no workgroup barrier anywhere: a wave that is deep in a mesh never holds the other three up.  (The first
// version parked per workgroup with three barriers per drain; rocprofv3 showed the C4 shadow kernel waiting 68 % of its
// wave cycles at 12 % VALU issue.)
// (mesh_walk's speculation — a lane that holds a leaf searches on for its next one while others still search for their first — per parked kernel family)
#ifndef PT_WALK_SPEC_SHADOW
#define PT_WALK_SPEC_SHADOW false
#endif
#ifndef PT_WALK_SPEC_EXTEND
#define PT_WALK_SPEC_EXTEND false
#endif
#ifndef PT_PARKED_NT_RAY
#define PT_PARKED_NT_RAY false
#endif
#ifndef PT_PARKED_EAGER
#define PT_PARKED_EAGER true
#endif
#ifndef PT_PARKED_EXP
#define PT_PARKED_EXP 0   // measurement variants of k_shadow_parked (tools/phase_costs_parked.sh); 0 = the product
#endif
constexpr uint32_t kWaveParkCap = kParkCap / (kBlock / 64);
static_assert(kWaveParkCap == 128, "a wave's park list: fewer than 64 entries left over + at most 64 new ones per step");

// `walk_policy` (mesh_walk's: pt_tuning::walk_evict_below | walk_search_below << 8): a resumed wave's walks are left by its last lanes once
// fewer than walk_evict_below are still walking — they park again with their cursor and go on in a later drain.  Not in the very last drain
// of a wave, which has nobody left to wait for.
template <bool ALL_LANES, uint32_t PCAP = ptk::kParkCap, int REC = PK_FULL, typename Resume>
__device__ __forceinline__ void park_drain(uint32_t* pk, uint32_t* park_count, bool last, uint32_t walk_policy, Resume&& resume, bool with_bh = true) {
    const uint32_t lane = lane_id();
    for (;;) {
        __threadfence_block();             // this wave's parked entries are visible to its other lanes
        const uint32_t cnt = *park_count;  // the same for every lane of the wave
        if (!(cnt >= 64u || (last && cnt > 0u))) break;
        const uint32_t take = cnt < 64u ? cnt : 64u, first = cnt - take;
        const bool mine = lane < take;
        uint32_t item = 0, ray = 0, kind = 0, cursor = 0; float bound = PT_INF; SweepState st;
        if (ALL_LANES) sweep_state_init(st, 0);
        if (mine) park_load<PCAP, REC>(pk, first + lane, &item, &st, &ray, &bound, &kind, &cursor, with_bh);
        __threadfence_block();             // entries are in registers before any lane parks again into these slots
        if (lane == 0) *park_count = first;
        // (ALL_LANES, the forms that scan axis rays: a lane without an entry helps with the scans of the others' — mesh_walk's `alive`)
        if (ALL_LANES || mine) resume(item, st, ray, bound, kind, cursor, last && first == 0u ? walk_policy & ~0xfffe00ffu : walk_policy, mine);   // (the very last drain: neither the mesh walks, nor the grouped sweep's group loops, nor the top-level walks are left early)
    }
}

// TOP = 1: no sweep table — the top-level tree is walked per lane and a lane parks at every mesh instance it reaches (top_walk_run / top_walk_resume,
// pt_device.h): the parked kernels' mesh walks for scenes of more than 64 instances.
// BLK (round 4): threads per workgroup.  A blob too big to stage whole at 256 threads — the gem scene's 65 KB leaves two workgroups per CU, two waves
// per SIMD — can be staged whole by workgroups of 512: two of them are sixteen waves per CU, four per SIMD, and the mesh is read by ds_read.  Waves never
// meet after the staging barrier, so nothing else changes: a wave's park list, its rounds over the segment, its drains.
template <int USE_LDS, int TOP = 0, int BLK = kBlock>
__global__ void __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu(BLK == kBlock ? PT_PARK_EXTEND_WAVES : PT_PARK_WAVES)))
k_extend_parked(const uint32_t* __restrict__ blob, uint32_t blob_words, const float* __restrict__ tex,
                Queue paths, Queue hits, uint32_t seg_cap, const uint32_t* __restrict__ count_in,
                uint32_t* __restrict__ park_all, uint32_t walk_policy, uint32_t path_marks) {
    // `path_marks`: 1 + the instance a MARKED segment cannot hit (PT_HDR_CONVEX_INST; the mark is the sign of the record's previous-pdf word, written by the vertex kernel —
    // so never at bounce 0, whose records k_generate wrote: the engine passes 0 there, and for a scene without such an instance)
    extern __shared__ __align__(16) uint32_t lds[];
    __shared__ uint32_t park_counts[BLK / 64];
    constexpr uint32_t kParkCap = kWaveParkCap * (BLK / 64);   // (shadows ptk::kParkCap: this workgroup's entries per field)
    if (PT_EMPTY_SEGMENT_EXIT && count_in[blockIdx.x] == 0u) return;
    SceneView s = stage_scene<USE_LDS>(blob, blob_words, tex, lds);
    const uint32_t wave = PT_UNIFORM(threadIdx.x >> 6);   // (a scalar: with fresh_lane_id below, threadIdx.x need not stay in a register across the rounds)
    uint32_t* pk = park_all + (size_t)blockIdx.x * kParkFields * kParkCap + wave * kWaveParkCap;  // field f of entry e at pk[f * kParkCap + e]
    uint32_t* park_count = &park_counts[wave];
    const uint32_t base = blockIdx.x * seg_cap, n = count_in[blockIdx.x];
    const uint32_t rounds = (n + blockDim.x - 1) / blockDim.x;
    if (lane_id() == 0) *park_count = 0;
    auto ray_of = [&](uint32_t i, F3* o, F3* d) {
        *o = f3(qf(paths, PS_OX, i), qf(paths, PS_OY, i), qf(paths, PS_OZ, i));
        *d = f3(qf(paths, PS_DX, i), qf(paths, PS_DY, i), qf(paths, PS_DZ, i));
    };
    // (`aux`: the groups a ray evicted from the grouped mesh sweep has still to do, mesh_walk — in the two words of an entry that a path segment has to spare: ray number and kind)
    auto settle = [&](uint32_t j, F3 o, F3 d, const SweepState& st, bool parked, uint32_t cursor, uint64_t aux = 0ull) {
        if (parked) park_store<kParkCap, PT_PARK_SEGMENT_REC>(pk, atomicAdd(park_count, 1u), j, st, (uint32_t)aux, PT_INF, (uint32_t)(aux >> 32), cursor);
        else { Hit h; sweep_finish(s, o, d, st, &h); store_hit(hits, base + j, h); }
    };
    PT_TL_BEGIN();
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t j = r * blockDim.x + (wave << 6 | fresh_lane_id());
        if (j < n) {
            F3 o, d;
            ray_of(base + j, &o, &d);
            SweepState st;
            bool parks, evicted = false;
            if (TOP) { top_walk_init(st); parks = top_walk_run(s, o, d, PT_INF, PT_STOP_NONE, st, true, (walk_policy >> 24) & 0xffu, &evicted); }
            else {
                sweep_state_init(st, sweep_masks(s, o, d, PT_INF));
                if (PT_CONVEX_SKIP && path_marks != 0u && qf(paths, PS_PREV_PDF, base + j) < 0.0f) st.hit &= ~sweep_instance_mask(s, path_marks - 1u);
                const TriRay wtr = tri_ray_prepare(o, d);
                parks = sweep_run(s, o, d, wtr, PT_INF, PT_STOP_NONE, st, true);   // (parks at the walked mesh: `inside` is looked up again when the ray is resumed)
            }
            settle(j, o, d, st, parks, evicted ? PT_TOP_EVICTED : 0u);
        }
        park_drain<false, kParkCap, PT_PARK_SEGMENT_REC>(pk, park_count, r + 1 == rounds, walk_policy & ~PT_WALK_SCAN_AXIS, [&](uint32_t j2, SweepState& st, uint32_t aux_lo, float, uint32_t aux_hi, uint32_t cursor, uint32_t policy, bool mine) {
            if (mine) PT_TL_WORK();   // (timeline: parked rays resumed)
            F3 o = f3(0.0f, 0.0f, 0.0f), d = f3(0.0f, 0.0f, 0.0f);
            if (mine) ray_of(base + j2, &o, &d);
            uint64_t aux = (uint64_t)aux_lo | (uint64_t)aux_hi << 32;
            // (a segment marked "starts inside the scene's one certified body", PT_PATH_INSIDE_MARK: mesh_walk may end that body's sweep at the first triangle accepted well inside itself)
            const uint32_t inside_inst = (PT_CONVEX_SKIP && path_marks != 0u && mine && (qu(paths, PS_SLOT, base + j2) & PT_PATH_INSIDE_MARK)) ? path_marks - 1u : 0xffffffffu;
            const bool again = TOP ? top_walk_resume<PT_WALK_SPEC_EXTEND>(s, o, d, PT_INF, PT_STOP_NONE, st, &cursor, policy, mine)
                                   : sweep_resume<PT_WALK_SPEC_EXTEND>(s, o, d, PT_INF, PT_STOP_NONE, st, 0xffffffffu, 0.0f, &cursor, policy, mine, &aux, inside_inst);
            if (mine) settle(j2, o, d, st, again, cursor, aux);
        });
    }
    PT_TL_END(n);
}

// (LACKS = PT_SCENE_NO_LIGHTS: the light list is empty — hdri_test — so every light-sample ray is an environment ray: the light pre-pass, the
// light's record and its emission are compiled out)
// Rays parallel to an axis of their mesh are scanned by the whole wave (mesh_scan, pt_device.h): the environment sample at the pole of the importance map is such a
// ray, 1.5 in 10 000 of C4's light samples, each 8000 box tests long when walked.  (The closest-hit kernel, k_extend_parked, walks them: there an axis-parallel
// direction is a coincidence of the scene's set-up, and the scan's registers cost that 96-VGPR kernel 6 spilled.  Here the form with the scan happens to
// allocate better than the one without: C3's kernel 4755 -> 4658 us, so every form carries it.)
// `live_off`: where in the dynamic LDS, in words, the waves' lists of live rays begin (behind the staged blob; launch_shadow).  BLK: see k_extend_parked.
template <int USE_LDS, int NL, uint32_t LACKS = 0u, int TOP = 0, int BLK = kBlock>
__global__ void __launch_bounds__(BLK) PT_PARK_OCC k_shadow_parked(const uint32_t* __restrict__ blob, uint32_t blob_words, const float* __restrict__ tex,
                                                                     uint32_t light_samples, Queue shadow, float* __restrict__ energy, uint32_t energy_stride,
                                                                     uint32_t seg_cap, const uint32_t* __restrict__ count_in, uint32_t* __restrict__ park_all, uint32_t walk_policy,
                                                                     uint32_t live_off) {
    extern __shared__ __align__(16) uint32_t lds[];
    __shared__ uint32_t park_counts[BLK / 64];
    constexpr uint32_t kParkCap = kWaveParkCap * (BLK / 64);   // (shadows ptk::kParkCap: this workgroup's entries per field)
    if (PT_EMPTY_SEGMENT_EXIT && count_in[blockIdx.x] == 0u) return;
    SceneView s = stage_scene<USE_LDS, LACKS>(blob, blob_words, tex, lds);
    constexpr bool kOnlyEnv = (LACKS & PT_SCENE_NO_LIGHTS) != 0u;
    // (a parked light-sample ray carries its closest hit's barycentrics only where a triangle can be a light: a mesh instance overridden with a light material)
    const bool with_bh = (PT_UNIFORM(bu(s, PT_HDR_FLAGS)) & PT_FLAG_NO_SHADOW_BOUND) != 0u;
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t* pk = park_all + (size_t)blockIdx.x * kParkFields * kParkCap + wave * kWaveParkCap;
    uint32_t* park_count = &park_counts[wave];
    const uint32_t base = blockIdx.x * seg_cap, n = count_in[blockIdx.x];
    const uint32_t rounds = (n + blockDim.x - 1) / blockDim.x;
    if (lane_id() == 0) *park_count = 0;
    // a finished ray leaves its contribution where its factor was; the item's rays are summed in order at the end (an item's
    // rays are parked and resumed by the wave that owns the item, so that sum needs no workgroup barrier either)
    // (`lambda0`, one wavelength per path: the item's wavelength, read along with the ray — PT_PARKED_EAGER — instead of behind the search)
    auto settle = [&](uint32_t j, uint32_t l, const ShadowRayT<NL>& ray, bool env, float bound, const SweepState& st, bool parked, uint32_t light, uint32_t cursor, float lambda0) {
        if (parked) { park_store<kParkCap, PK_LIGHT_RAY>(pk, atomicAdd(park_count, 1u), j, st, l, bound, (env ? 1u : 0u) | (light + 1u) << 1, cursor, with_bh); return; }
        const uint32_t item = base + j;
        float lambda[NL], c[NL];
        for (int k = 0; k < NL; ++k) lambda[k] = (NL == 1 && PT_PARKED_EAGER) ? lambda0 : qf(shadow, Layout<NL>::sh_lambda + k, item);
        Hit sh; sh.valid = false;
        const bool hit = st.best_inst != 0xffffffffu;
        // only a light or nothing at all contributes: the record of an occluder is never read (shadow_ray_contribution)
        const bool wanted = hit && !env && sweep_best_is_light(s, st);
        if (wanted) sweep_finish(s, ray.o, ray.d, st, &sh);
        if (hit && !wanted) { sh.valid = true; sh.material = PT_MATERIAL_ID(PT_TAG_MATERIAL, 0); }
        shadow_ray_contribution<NL>(s, [&](int k) { return pl_get<NL>(lambda, k); }, ray, env, hit, sh, c);
        for (int k = 0; k < NL; ++k) qsf(shadow, Layout<NL>::sh_head + l * Layout<NL>::sr_fields + SR_FACTOR + k, item, c[k]);
    };
    // (the parked `kind` word: bit 0 = an environment sample, the rest = 1 + the light whose hit bounds the search, sweep_run's known_inst)
    PT_TL_BEGIN();
    auto resume_parked = [&](uint32_t j2, SweepState& st, uint32_t l2, float bound, uint32_t kind, uint32_t cursor, uint32_t policy, bool mine) {
        if (mine) PT_TL_WORK();   // (timeline: parked rays resumed)
        ShadowRayT<NL> pr;
        pr.o = f3(0.0f, 0.0f, 0.0f); pr.d = f3(0.0f, 0.0f, 0.0f);
        if (mine) load_shadow_ray<NL, PT_PARKED_EAGER, PT_PARKED_NT_RAY>(shadow, base + j2, l2, &pr);
        const float lam0 = (NL == 1 && PT_PARKED_EAGER && mine) ? qf(shadow, Layout<NL>::sh_lambda, base + j2) : 0.0f;
        const bool env = kOnlyEnv || (kind & 1u) != 0u;
        // a light ray searches with the early stop whenever it has a finite bound (shadow_light_bound)
        const int stop2 = env ? shadow_env_stop(s) : (bound < PT_INF ? PT_STOP_NONLIGHT : PT_STOP_NONE);
        bool again;
        if (PT_PARKED_EXP & 8) {   // (measurement: parked rays are stored, loaded and carried on behind their mesh — everything but the mesh walk itself)
            again = false;
            if (mine) { st.hit &= st.hit - 1; if (st.hit != 0) { const TriRay wtr = tri_ray_prepare(pr.o, pr.d); again = sweep_run<true>(s, pr.o, pr.d, wtr, bound, stop2, st, true, (kind >> 1) - 1u, bound); } }
        } else
        again = TOP ? top_walk_resume<PT_WALK_SPEC_SHADOW>(s, pr.o, pr.d, bound, stop2, st, &cursor, policy, mine)
                    : sweep_resume<PT_WALK_SPEC_SHADOW>(s, pr.o, pr.d, bound, stop2, st, (kind >> 1) - 1u, bound, &cursor, policy, mine);
        if (mine) settle(j2, l2, pr, env, bound, st, again, (kind >> 1) - 1u, cursor, lam0);
    };
    // The rays that are traced at all — a light sample below the horizon of its surface, or with a zero factor, is not: four in five of
    // C4's — are listed per wave (item and sample number) and traced 64 at a time, so that a step of the wave is 64 live rays and parks at
    // most one ray per lane.  (One ray of every item per step left the lanes of the dead ones idle: lane utilisation 0.21 on C4.)
    uint32_t* live = lds + live_off + wave * live_cap(light_samples);
    uint32_t live_count = 0;   // (wave-uniform)
    for (uint32_t r = 0;;) {
        const bool flush = r == rounds && live_count > 0u;
        if (!(live_count >= 64u || flush)) {
            if (r == rounds) break;
            // list the live rays of the wave's next 64 items: per sample number one ballot
            const uint32_t j = r * blockDim.x + threadIdx.x;
            ++r;
            for (uint32_t l = 0; l < light_samples; ++l) {
                bool lives = false;
                if (j < n) for (int k = 0; k < NL; ++k) lives = lives || qf(shadow, Layout<NL>::sh_head + l * Layout<NL>::sr_fields + SR_FACTOR + k, base + j) != 0.0f;
                const unsigned long long m = __ballot(lives);
                if (lives) live[live_count + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull))] = j << 3 | l;
                live_count += (uint32_t)__popcll(m);
            }
            __builtin_amdgcn_wave_barrier();
            if (r < rounds || live_count >= 64u) continue;   // (after the last round: what is left is traced below, then the last parked rays are resumed)
        }
        const uint32_t take = live_count < 64u ? live_count : 64u;
        live_count -= take;
        if (PT_PARKED_EXP & 1) continue;   // (measurement: the listing of the live rays alone)
        if (lane_id() < take) {
            const uint32_t e = live[live_count + lane_id()], j = e >> 3, l = e & 7u, item = base + j;
            ShadowRayT<NL> ray;
            // (a listed ray is live: its origin and direction are read along with its factors, one round trip to memory instead of two in a row — these
            // kernels run three waves per SIMD, which hide little: round 4, PT_PARKED_EAGER)
            load_shadow_ray<NL, PT_PARKED_EAGER, PT_PARKED_NT_RAY>(shadow, item, l, &ray);
            const float lam0 = (NL == 1 && PT_PARKED_EAGER) ? qf(shadow, Layout<NL>::sh_lambda, item) : 0.0f;
            // (the item's flag word: bit l = an environment sample; bit 8 + l = the ray left a certified convex body outward and cannot hit that instance, word >> 16, again:
            // pt_blob.h PT_INST_CONVEX_OUT — its bits leave the ray's leaf mask, so the ray neither parks at that mesh nor walks it)
            const uint32_t iflags = kOnlyEnv ? 0u : qu(shadow, Layout<NL>::sh_flags, item);
            const bool env = kOnlyEnv || ((iflags >> l) & 1u) != 0;
            float bound = PT_INF; int stop = shadow_env_stop(s);
            uint32_t light = 0xffffffffu;
            if (!env && !shadow_light_bound(s, ray.o, ray.d, &bound, &stop, &light)) {
                for (int k = 0; k < NL; ++k) qsf(shadow, Layout<NL>::sh_head + l * Layout<NL>::sr_fields + SR_FACTOR + k, item, 0.0f);
            } else {
                SweepState st;
                bool parks = false, evicted = false;
                if (TOP) { top_walk_init(st); parks = top_walk_run(s, ray.o, ray.d, bound, stop, st, true, (walk_policy >> 24) & 0xffu, &evicted); }
                else {
                    sweep_state_init(st, sweep_masks(s, ray.o, ray.d, bound));
                    if (PT_CONVEX_SKIP && ((iflags >> (8u + l)) & 1u) != 0u) st.hit &= ~sweep_instance_mask(s, iflags >> 16);
                    if (!(PT_PARKED_EXP & 2)) {
                        const TriRay wtr = tri_ray_prepare(ray.o, ray.d);
                        parks = sweep_run(s, ray.o, ray.d, wtr, bound, stop, st, true, light, bound);
                    }
                }
                // (measurement variants — no lane may leave the wave's step early, the barrier and the drain below are the whole wave's: 2 = up to the masks, 4 = parked rays dropped)
                if (PT_PARKED_EXP & 2) qsf(shadow, Layout<NL>::sh_head + l * Layout<NL>::sr_fields + SR_FACTOR, item, (float)(uint32_t)st.hit);
                else if ((PT_PARKED_EXP & 4) && parks) qsf(shadow, Layout<NL>::sh_head + l * Layout<NL>::sr_fields + SR_FACTOR, item, 0.0f);
                else settle(j, l, ray, env, bound, st, parks, light, evicted ? PT_TOP_EVICTED : 0u, lam0);
            }
        }
        __builtin_amdgcn_wave_barrier();
        park_drain<true, kParkCap, PK_LIGHT_RAY>(pk, park_count, r == rounds && live_count == 0u, walk_policy, resume_parked, with_bh);
    }
    PT_TL_END(n);
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

// ---- the parked kernels with dynamic work distribution ------------------------------------------------------------------------------
// The static form above gives every workgroup one queue segment; its park lists start empty and end with a partly filled drain, and
// at deep bounces a segment holds too few rays that enter the big mesh to ever fill a wave.  Here a few persistent workgroups per CU
// take UNITS of work — kUnitItems consecutive items of a segment — from one counter, wave by wave: a wave's park list lives across
// all the units it takes, so drains stay full until the very end, and the hardware no longer has to balance segments of different cost
// (the static variant with several segments per workgroup gained 7 % on C3 and lost 24 % on C4 for that reason).  Waves are fully
// independent: no workgroup barrier after the blob is staged.
constexpr uint32_t kUnitItems = 1024;
__device__ __forceinline__ bool next_unit(uint32_t* __restrict__ counter, uint32_t total_units, uint32_t units_per_seg, uint32_t seg_cap,
                                          const uint32_t* __restrict__ count_in, uint32_t* first, uint32_t* cnt) {
    uint32_t u = 0;
    if (lane_id() == 0) u = atomicAdd(counter, 1u);
    u = (uint32_t)__builtin_amdgcn_readfirstlane((int)u);
    *first = 0; *cnt = 0;
    if (u >= total_units) return false;
    const uint32_t seg = u / units_per_seg, c = u - seg * units_per_seg, n = count_in[seg];
    *first = seg * seg_cap + c * kUnitItems;
    *cnt = n > c * kUnitItems ? (n - c * kUnitItems < kUnitItems ? n - c * kUnitItems : kUnitItems) : 0u;
    return true;
}

template <int USE_LDS>
__global__ void __launch_bounds__(kBlock) PT_PARK_EXTEND_OCC k_extend_parked_dyn(const uint32_t* __restrict__ blob, uint32_t blob_words, const float* __restrict__ tex,
                                                                         Queue paths, Queue hits, uint32_t seg_cap, const uint32_t* __restrict__ count_in,
                                                                         uint32_t* __restrict__ park_all, uint32_t n_segments, uint32_t* __restrict__ unit_counter, uint32_t walk_policy) {
    extern __shared__ __align__(16) uint32_t lds[];
    __shared__ uint32_t park_counts[kBlock / 64];
    SceneView s = stage_scene<USE_LDS>(blob, blob_words, tex, lds);
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    uint32_t* pk = park_all + (size_t)blockIdx.x * kParkFields * kParkCap + wave * kWaveParkCap;
    uint32_t* park_count = &park_counts[wave];
    if (lane == 0) *park_count = 0;
    const uint32_t units_per_seg = (seg_cap + kUnitItems - 1) / kUnitItems, total_units = n_segments * units_per_seg;
    auto ray_of = [&](uint32_t i, F3* o, F3* d) {
        *o = f3(qf(paths, PS_OX, i), qf(paths, PS_OY, i), qf(paths, PS_OZ, i));
        *d = f3(qf(paths, PS_DX, i), qf(paths, PS_DY, i), qf(paths, PS_DZ, i));
    };
    auto settle = [&](uint32_t i, F3 o, F3 d, const SweepState& st, bool parked, uint32_t cursor) {   // `i`: the item's index in the queue
        if (parked) park_store(pk, atomicAdd(park_count, 1u), i, st, 0u, PT_INF, 0u, cursor);
        else { Hit h; sweep_finish(s, o, d, st, &h); store_hit(hits, i, h); }
    };
    auto resume = [&](uint32_t i2, SweepState& st, uint32_t, float, uint32_t, uint32_t cursor, uint32_t policy, bool mine) {
        F3 o = f3(0.0f, 0.0f, 0.0f), d = f3(0.0f, 0.0f, 0.0f);
        if (mine) ray_of(i2, &o, &d);
        const bool again = sweep_resume(s, o, d, PT_INF, PT_STOP_NONE, st, 0xffffffffu, 0.0f, &cursor, policy, mine);
        if (mine) settle(i2, o, d, st, again, cursor);
    };
    // one loop, one drain site (the resume code is the bulk of the kernel: two inlined copies cost 25 KB of instruction cache): each turn
    // takes the next 64 items of the current unit — or a new unit, or nothing when the counter has run out — and drains; the last turn
    // drains the partial wave that is left
    uint32_t first = 0, cnt = 0, off = 0;
    bool more = true;
    for (;;) {
        if (off >= cnt && more) { more = next_unit(unit_counter, total_units, units_per_seg, seg_cap, count_in, &first, &cnt); off = 0; }
        if (off + lane < cnt) {
            const uint32_t i = first + off + lane;
            F3 o, d;
            ray_of(i, &o, &d);
            SweepState st;
            sweep_state_init(st, sweep_masks(s, o, d, PT_INF));
            const TriRay wtr = tri_ray_prepare(o, d);
            settle(i, o, d, st, sweep_run(s, o, d, wtr, PT_INF, PT_STOP_NONE, st, true), 0u);
        }
        off += 64u;
        const bool last = !more && off >= cnt;
        park_drain<false>(pk, park_count, last, walk_policy & ~PT_WALK_SCAN_AXIS, resume);
        if (last) break;
    }
}

// The rays only: every finished ray leaves its contribution where its factor was; k_shadow_sum adds an item's rays up afterwards.
template <int USE_LDS, int NL>
__global__ void __launch_bounds__(kBlock) PT_PARK_OCC k_shadow_parked_dyn(const uint32_t* __restrict__ blob, uint32_t blob_words, const float* __restrict__ tex,
                                                                         uint32_t light_samples, Queue shadow, uint32_t seg_cap, const uint32_t* __restrict__ count_in,
                                                                         uint32_t* __restrict__ park_all, uint32_t n_segments, uint32_t* __restrict__ unit_counter, uint32_t walk_policy) {
    extern __shared__ __align__(16) uint32_t lds[];
    __shared__ uint32_t park_counts[kBlock / 64];
    SceneView s = stage_scene<USE_LDS>(blob, blob_words, tex, lds);
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    uint32_t* pk = park_all + (size_t)blockIdx.x * kParkFields * kParkCap + wave * kWaveParkCap;
    uint32_t* park_count = &park_counts[wave];
    if (lane == 0) *park_count = 0;
    const uint32_t units_per_seg = (seg_cap + kUnitItems - 1) / kUnitItems, total_units = n_segments * units_per_seg;
    auto settle = [&](uint32_t item, uint32_t l, const ShadowRayT<NL>& ray, bool env, float bound, const SweepState& st, bool parked, uint32_t light, uint32_t cursor) {
        if (parked) { park_store(pk, atomicAdd(park_count, 1u), item, st, l, bound, (env ? 1u : 0u) | (light + 1u) << 1, cursor); return; }
        float lambda[NL], c[NL];
        for (int k = 0; k < NL; ++k) lambda[k] = qf(shadow, Layout<NL>::sh_lambda + k, item);
        Hit sh; sh.valid = false;
        const bool hit = st.best_inst != 0xffffffffu;
        const bool wanted = hit && !env && sweep_best_is_light(s, st);   // (the record of an occluder is never read)
        if (wanted) sweep_finish(s, ray.o, ray.d, st, &sh);
        if (hit && !wanted) { sh.valid = true; sh.material = PT_MATERIAL_ID(PT_TAG_MATERIAL, 0); }
        shadow_ray_contribution<NL>(s, [&](int k) { return pl_get<NL>(lambda, k); }, ray, env, hit, sh, c);
        for (int k = 0; k < NL; ++k) qsf(shadow, Layout<NL>::sh_head + l * Layout<NL>::sr_fields + SR_FACTOR + k, item, c[k]);
    };
    auto resume = [&](uint32_t item2, SweepState& st, uint32_t l2, float bound, uint32_t kind, uint32_t cursor, uint32_t policy, bool mine) {   // (`kind`: see k_shadow_parked)
        ShadowRayT<NL> pr;
        pr.o = f3(0.0f, 0.0f, 0.0f); pr.d = f3(0.0f, 0.0f, 0.0f);
        if (mine) load_shadow_ray<NL>(shadow, item2, l2, &pr);
        const bool env = (kind & 1u) != 0u;
        const int stop2 = env ? shadow_env_stop(s) : (bound < PT_INF ? PT_STOP_NONLIGHT : PT_STOP_NONE);
        const bool again = sweep_resume(s, pr.o, pr.d, bound, stop2, st, (kind >> 1) - 1u, bound, &cursor, policy, mine);
        if (mine) settle(item2, l2, pr, env, bound, st, again, (kind >> 1) - 1u, cursor);
    };
    uint32_t first = 0, cnt = 0, off = 0;   // (one loop, one drain site: see k_extend_parked_dyn)
    bool more = true;
    for (;;) {
        if (off >= cnt && more) { more = next_unit(unit_counter, total_units, units_per_seg, seg_cap, count_in, &first, &cnt); off = 0; }
        const bool active = off + lane < cnt;
        const uint32_t item = first + off + lane, flags = active ? qu(shadow, Layout<NL>::sh_flags, item) : 0u;
        off += 64u;
        const bool last_turn = !more && off >= cnt;
        for (uint32_t l = 0; l < light_samples; ++l) {   // one ray of every item per step, so that a step parks at most one ray per lane
            ShadowRayT<NL> ray;
            if (active && load_shadow_ray<NL>(shadow, item, l, &ray)) {
                const bool env = ((flags >> l) & 1u) != 0;
                float bound = PT_INF; int stop = shadow_env_stop(s);
                uint32_t light = 0xffffffffu;
                if (!env && !shadow_light_bound(s, ray.o, ray.d, &bound, &stop, &light)) {
                    for (int k = 0; k < NL; ++k) qsf(shadow, Layout<NL>::sh_head + l * Layout<NL>::sr_fields + SR_FACTOR + k, item, 0.0f);
                } else {
                    SweepState st;
                    sweep_state_init(st, sweep_masks(s, ray.o, ray.d, bound));
                    if (PT_CONVEX_SKIP && ((flags >> (8u + l)) & 1u) != 0u) st.hit &= ~sweep_instance_mask(s, flags >> 16);   // (see k_shadow_parked)
                    const TriRay wtr = tri_ray_prepare(ray.o, ray.d);
                    settle(item, l, ray, env, bound, st, sweep_run(s, ray.o, ray.d, wtr, bound, stop, st, true, light, bound), light, 0u);
                }
            }
            park_drain<false>(pk, park_count, last_turn && l + 1 == light_samples, walk_policy & ~PT_WALK_SCAN_AXIS, resume);
        }
        if (last_turn) break;
    }
}
// pt.rs:349-392, 596: an item's rays summed in order, divided by L, added to its slot (workgroup b owns segment b, as everywhere)
template <int NL>
__global__ void __launch_bounds__(kBlock) k_shadow_sum(uint32_t light_samples, Queue shadow, float* __restrict__ energy, uint32_t energy_stride,
                                                      uint32_t seg_cap, const uint32_t* __restrict__ count_in) {
    const uint32_t base = blockIdx.x * seg_cap, n = count_in[blockIdx.x];
    for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
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
            r.material = h.material; r.instance = hit_instance_index(s, h.instance);
        }
        out[i] = r;
    }
}
}  // namespace ptk
#endif
