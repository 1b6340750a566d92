// pt_kern_extend.hip — the closest-hit kernels (k_extend in its traversal forms, k_probe_intersect) and their launcher.
#include <cstdlib>
#include "pt_kernels.h"

namespace ptk {

#ifdef PT_EXPERIMENTS
uint32_t pool_lds_bytes() { return (kBlock / 64) * PT_POOL_WORDS * 4u; }
#endif

#define PT_GO(K, ...) go(c, K, __VA_ARGS__)
#define PT_BY_MODE(K, ...) do { if (c.lds_mode == PT_LDS_ALL) PT_GO(K(PT_LDS_ALL), __VA_ARGS__); else if (c.lds_mode == PT_LDS_CORE) PT_GO(K(PT_LDS_CORE), __VA_ARGS__); \
                                else PT_GO(K(PT_LDS_NONE), __VA_ARGS__); } while (0)
#define K_EXT_PARKED(M) k_extend_parked<M>
#define K_EXT_PARKED_W(M) k_extend_parked<M, 1>
#define K_EXT_ANY(M) k_extend<M, PT_TRAV_ANY>
#define K_PROBE(M) k_probe_intersect<M>

// (PT_FORM_POOLED and the k_*_exp measurement variants: builds with EXTRA=-DPT_EXPERIMENTS only — measured slower, profiles/r2_experiments.md)
// PT_FORM_WALK / SWEEP / POOLED exist for the fully staged blob only (the engine asks for PT_FORM_ANY otherwise)
void launch_extend(const LaunchCfg& c, int form, const SceneArgs& sc, Queue paths, Queue hits, uint32_t seg_cap, const uint32_t* count_in, uint32_t* park) {
#ifdef PT_EXPERIMENTS
    if (const char* v = getenv("PT_AMD_EXP")) {
        const int e = atoi(v);
#define PT_EXP_CASE(E) if (e == E) { static bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(k_extend_exp<E>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536), true); (void)once; \
                                     PT_GO(k_extend_exp<E>, sc.blob, sc.blob_words, sc.tex, paths, hits, seg_cap, count_in); }
        PT_EXP_CASE(0x6) PT_EXP_CASE(0x4) PT_EXP_CASE(0x0) PT_EXP_CASE(0x7) PT_EXP_CASE(0x7d) PT_EXP_CASE(0x3d) PT_EXP_CASE(0x35) PT_EXP_CASE(0x25) PT_EXP_CASE(0x5) PT_EXP_CASE(0x1)
    }
#endif
    if (form == PT_FORM_PARKED && c.unit_counter) {
        LaunchCfg d = c; d.grid = c.dyn_grid;
#define K_EXT_PARKED_DYN(M) k_extend_parked_dyn<M>
        if (c.lds_mode == PT_LDS_ALL) go(d, K_EXT_PARKED_DYN(PT_LDS_ALL), sc.blob, sc.blob_words, sc.tex, paths, hits, seg_cap, count_in, park, (uint32_t)c.grid, c.unit_counter, c.walk_policy);
        else if (c.lds_mode == PT_LDS_CORE) go(d, K_EXT_PARKED_DYN(PT_LDS_CORE), sc.blob, sc.blob_words, sc.tex, paths, hits, seg_cap, count_in, park, (uint32_t)c.grid, c.unit_counter, c.walk_policy);
        else go(d, K_EXT_PARKED_DYN(PT_LDS_NONE), sc.blob, sc.blob_words, sc.tex, paths, hits, seg_cap, count_in, park, (uint32_t)c.grid, c.unit_counter, c.walk_policy);
    } else if (form == PT_FORM_PARKED && c.park_block_extend == 512) { go_block(c, 512, c.park_blob_bytes, k_extend_parked<PT_LDS_ALL, 0, 512>, sc.blob, sc.blob_words, sc.tex, paths, hits, seg_cap, count_in, park, c.walk_policy, c.path_marks); PT_TL_BUMP(c.stream); }
    else if (form == PT_FORM_PARKED && c.park_block_extend == 1024) { go_block(c, 1024, c.park_blob_bytes, k_extend_parked<PT_LDS_ALL, 0, 1024>, sc.blob, sc.blob_words, sc.tex, paths, hits, seg_cap, count_in, park, c.walk_policy, c.path_marks); PT_TL_BUMP(c.stream); }
    else if (form == PT_FORM_PARKED_WALK) { PT_BY_MODE(K_EXT_PARKED_W, sc.blob, sc.blob_words, sc.tex, paths, hits, seg_cap, count_in, park, c.walk_policy, 0u); PT_TL_BUMP(c.stream); }
    else if (form == PT_FORM_PARKED) { PT_BY_MODE(K_EXT_PARKED, sc.blob, sc.blob_words, sc.tex, paths, hits, seg_cap, count_in, park, c.walk_policy, c.path_marks); PT_TL_BUMP(c.stream); }
#ifdef PT_EXPERIMENTS
    else if (form == PT_FORM_POOLED) PT_GO(k_extend_pooled<PT_LDS_ALL>, sc.blob, sc.blob_words, sc.tex, paths, hits, seg_cap, count_in);
#endif
    else if (form == PT_FORM_SWEEP && (c.lacks & PT_SCENE_NO_XF)) PT_GO((k_extend<PT_LDS_ALL, PT_TRAV_SWEEP, PT_SCENE_NO_XF>), sc.blob, sc.blob_words, sc.tex, paths, hits, seg_cap, count_in);
    else if (form == PT_FORM_SWEEP) PT_GO((k_extend<PT_LDS_ALL, PT_TRAV_SWEEP>), sc.blob, sc.blob_words, sc.tex, paths, hits, seg_cap, count_in);
    else if (form == PT_FORM_WALK) PT_GO((k_extend<PT_LDS_ALL, PT_TRAV_WALK>), sc.blob, sc.blob_words, sc.tex, paths, hits, seg_cap, count_in);
    else PT_BY_MODE(K_EXT_ANY, sc.blob, sc.blob_words, sc.tex, paths, hits, seg_cap, count_in);
}

void launch_probe_intersect(const LaunchCfg& c, const SceneArgs& sc, uint32_t n, const float* o, const float* d, pt_hit* out) {
    PT_BY_MODE(K_PROBE, sc.blob, sc.blob_words, sc.tex, n, o, d, out);
}

hipError_t allow_lds_extend(uint32_t bytes) {
    hipError_t worst = hipSuccess;
    auto allow = [&](const void* k) { hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); if (e != hipSuccess) worst = e; };
#define PT_ALLOW_MODES(K) allow(reinterpret_cast<const void*>(K(PT_LDS_ALL))); allow(reinterpret_cast<const void*>(K(PT_LDS_CORE)))
    PT_ALLOW_MODES(K_EXT_ANY); PT_ALLOW_MODES(K_EXT_PARKED); PT_ALLOW_MODES(K_EXT_PARKED_W); PT_ALLOW_MODES(K_PROBE);
#define K_EXT_PARKED_DYN2(M) k_extend_parked_dyn<M>
    PT_ALLOW_MODES(K_EXT_PARKED_DYN2);
    allow(reinterpret_cast<const void*>(k_extend_parked<PT_LDS_ALL, 0, 512>)); allow(reinterpret_cast<const void*>(k_extend_parked<PT_LDS_ALL, 0, 1024>));
    allow(reinterpret_cast<const void*>(k_extend<PT_LDS_ALL, PT_TRAV_WALK>)); allow(reinterpret_cast<const void*>(k_extend<PT_LDS_ALL, PT_TRAV_SWEEP>));
    allow(reinterpret_cast<const void*>(k_extend<PT_LDS_ALL, PT_TRAV_SWEEP, PT_SCENE_NO_XF>));
#ifdef PT_EXPERIMENTS
    allow(reinterpret_cast<const void*>(k_extend_pooled<PT_LDS_ALL>));
#endif
    return worst;
}

#ifdef PT_TIMELINE
PT_TL_ACCESSOR(pt_debug_timeline_extend)
#endif

}  // namespace ptk
