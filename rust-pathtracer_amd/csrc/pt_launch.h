// pt_launch.h — host-side launchers of the kernel families.  Each family's kernels are instantiated in its own translation unit
// (pt_kern_extend.hip, pt_kern_shade.hip, pt_kern_shadow.hip: the build compiles them in parallel); the engine picks the variant —
// staging mode (PT_LDS_*) x traversal form x wavelengths per path x what the scene can need — and calls these.
#ifndef PT_LAUNCH_H
#define PT_LAUNCH_H
#include <hip/hip_runtime.h>
#include "pt_stages.h"

namespace ptk {

enum { PT_LDS_NONE = 0, PT_LDS_ALL = 1, PT_LDS_CORE = 2 };                 // what stage_scene copies into LDS
enum { PT_SHADE_LEAN = 0, PT_SHADE_NO_ENV = 1, PT_SHADE_FULL = 2, PT_SHADE_MEDIUM = 3 };   // k_shade forms (MEDIUM: k_shade_medium, one wavelength)
enum { PT_FORM_ANY = 0, PT_FORM_WALK = 1, PT_FORM_SWEEP = 2, PT_FORM_POOLED = 3, PT_FORM_PARKED = 4, PT_FORM_PARKED_WALK = 5 };  // traversal kernels (5: the parked kernels over the top-level tree instead of a sweep table)
constexpr int kBlock = 256;
constexpr uint32_t kLdsBlobLimitBytes = 64 * 1024;  // the most LDS a staged blob (or its core section) may take
// The whole blob is staged only while it leaves the CU its occupancy: six workgroups of 24 KB fit the 160 KB.  Measured (tools/lds_mode.sh):
// the 65 KB blob of C3 staged whole leaves two workgroups per CU, 615 Msamples/s; its 6.7 KB core alone, the mesh read through L1/L2,
// 752.  C2's 14 KB blob staged whole: 1519; its 7 KB core alone: 1089 (the sweep reads the triangles of the two boxes for every ray).
constexpr uint32_t kLdsAllLimitBytes = PT_BLOB_LDS_ALL_BYTES;
// The parked kernels in workgroups of 512 / 1024 threads (pt_tuning::park_block) stage the whole blob whatever the other kernels do: two workgroups of
// 512 per CU are four waves per SIMD, and 2 x (72 KB + the waves' lists of live rays) fit the CU's 160 KB.  The gem scene of C3: 66 256 B.
constexpr uint32_t kParkBlobLimitBytes = 72 * 1024;
constexpr uint32_t kParkCap = 512, kParkFields = 16;
constexpr uint32_t kLightPrepassMax = 16;   // pt_tuning::light_prepass_max's default: the most lights whose boxes a light-sample ray tests one by one for its bound
constexpr uint32_t kGroupEvictBelow = 32;   // pt_tuning::group_evict_below's default (1 = never)
constexpr uint32_t kTopEvictBelow = 48;   // pt_tuning::top_evict_below's default (1 = never); the check runs behind a leaf test of the while-while walk: G2F k_shadow_parked 2545 (never) / 2314 (16) / 2235 (32) / 2126 (40) / 2108 us (48)
constexpr uint32_t kWalkEvictBelow = 32, kWalkSearchBelow = 16;   // pt_tuning::walk_evict_below's and walk_search_below's defaults
enum { BS_VERTICES, BS_SHADOW_RAYS, BS_ENV_HITS, BS_SEGMENTS, BS_ITEMS, BS_MEDIUM_DROPS, BS_FIELDS };  // per-workgroup statistics (Profile counters)
#ifdef PT_EXPERIMENTS
uint32_t pool_lds_bytes();  // static LDS of the pooled traversal kernels, on top of the staged blob
#endif

struct LaunchCfg { int grid; uint32_t lds_bytes; hipStream_t stream; int lds_mode;
                   int dyn_grid = 0; uint32_t* unit_counter = nullptr;   // parked kernels with dynamic units: persistent workgroups and this launch's counter (zeroed)
                   uint32_t lacks = 0;    // PT_SCENE_* bits of what the scene does not hold: the pure sweep forms and the lean k_shade have forms without it
                   uint32_t walk_policy = 0;   // parked kernels: mesh_walk's policy word (pt_tuning::walk_evict_below | walk_search_below << 8)
                   int park_block = 0, park_block_extend = 0;   // parked kernels (light-sample / closest-hit), static form, one wavelength: 512 or 1024 = workgroups of that many threads that stage the WHOLE blob
                   uint32_t park_blob_bytes = 0;   //   (this many bytes of LDS) whatever lds_mode says for the other kernels; 0 = workgroups of kBlock, lds_mode
                   bool live_lists = false;    // measurement builds (-DPT_EXPERIMENTS, PT_AMD_LIVE_LISTS=1): k_shadow's sweep forms list the rays that search per wave (k_shadow_live)
                   bool certs = false;         // the scene holds a convex-body certificate (blob PT_FLAG_CONVEX): the vertex forms that carry the certificate code
                   uint32_t path_marks = 0;    // k_extend_parked: 1 + the instance a marked path segment cannot hit (blob PT_HDR_CONVEX_INST), 0 at bounce 0 and for scenes without one
                   bool fuse = false; };  // k_shade traces its own segments (pure sweep scenes, lean form, no transforms): no k_extend launch, no hit queue
struct SceneArgs { const uint32_t* blob; uint32_t blob_words; const float* tex; uint32_t marg_bytes = 0; /* LDS the FULL vertex form takes behind the blob (marginal_lds_bytes) */ };
// The importance map's marginal tables as k_shade's FULL form stages them (pt_kernels.h stage_marginal): 2 x rows floats + the guide's rows + 3 words, only for the interleaved layout.
// `staged_bytes`: what the kernel stages of the blob (all of it, its core, nothing).  0 = the tables stay in L2 (no importance map, too many rows, or no room: PT_SHADE_LDS_BUDGET).
inline uint32_t marginal_lds_bytes(const uint32_t* host_blob, uint32_t staged_bytes) {
    const uint32_t rows = host_blob[PT_HDR_IMAP_ROWS];
    if (host_blob[PT_HDR_ENV_KIND] != PT_ENV_HDR || rows == 0u || rows > PT_MARG_LDS_MAX_ROWS || host_blob[PT_HDR_IMAP_STRIDE] != 2u) return 0u;
    const uint32_t bytes = PT_MARG_LDS_BYTES(rows, host_blob[PT_HDR_IMAP_MARG_GUIDE] != 0u);
    return ((staged_bytes + 15u) & ~15u) + bytes <= PT_SHADE_LDS_BUDGET ? bytes : 0u;
}

void launch_extend(const LaunchCfg& c, int form, const SceneArgs& sc, ptd::Queue paths, ptd::Queue hits, uint32_t seg_cap, const uint32_t* count_in, uint32_t* park);
void launch_shade(const LaunchCfg& c, int nl, int form, const SceneArgs& sc, const ptd::RenderParams& rp, uint32_t bounce, const uint32_t* pixels, ptd::Queue paths_in,
                  ptd::Queue hits, ptd::Queue paths_out, ptd::Queue shadow, float* energy, uint32_t seg_cap, const uint32_t* count_in, uint32_t* count_out,
                  uint32_t* shadow_count, unsigned long long* block_stats);
void launch_shadow(const LaunchCfg& c, int form, int nl, bool env, const SceneArgs& sc, uint32_t light_samples, ptd::Queue shadow, float* energy, uint32_t energy_stride,
                   uint32_t seg_cap, const uint32_t* count_in, uint32_t* park, ptd::Queue idle_hits);
void launch_probe_intersect(const LaunchCfg& c, const SceneArgs& sc, uint32_t n, const float* o, const float* d, pt_hit* out);
// hipFuncAttributeMaxDynamicSharedMemorySize for every kernel of the family that stages the blob
hipError_t allow_lds_extend(uint32_t bytes);
hipError_t allow_lds_shade(uint32_t bytes);
hipError_t allow_lds_shadow(uint32_t bytes);

}  // namespace ptk
#endif
