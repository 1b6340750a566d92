// pt_kern_shadow.hip — the light-sample kernels (k_shadow in its traversal forms) and their launcher.
#define PT_MESH_DOP_ONLY_ANY 1   /* (pt_device.h mesh_surely_missed: in this family only the environment rays try a mesh's extra slabs) */
#include <cstdlib>
#ifdef PT_TIMELINE
#define PT_TIMELINE_RAYS   // (measurement build: pt_device.h records the long walks of this family's kernels)
#endif
#include "pt_kernels.h"

namespace ptk {

#define PT_GO(K, ...) go(c, K, __VA_ARGS__)
#define PT_BY_MODE(K, ...) do { if (c.lds_mode == PT_LDS_ALL) PT_GO(K(PT_LDS_ALL), __VA_ARGS__); else if (c.lds_mode == PT_LDS_CORE) PT_GO(K(PT_LDS_CORE), __VA_ARGS__); \
                                else PT_GO(K(PT_LDS_NONE), __VA_ARGS__); } while (0)
#define K_SH_PARKED1(M) k_shadow_parked<M, 1>
#define K_SH_PARKED1E(M) k_shadow_parked<M, 1, PT_SCENE_NO_LIGHTS>
#define K_SH_PARKED4(M) k_shadow_parked<M, 4>
#define K_SH_PARKED1W(M) k_shadow_parked<M, 1, 0u, 1>
#define K_SH_PARKED4W(M) k_shadow_parked<M, 4, 0u, 1>
#define K_SH_ANY1(M) k_shadow<M, 1, PT_TRAV_ANY>
#define K_SH_ANY4(M) k_shadow<M, 4, PT_TRAV_ANY>
#define PT_ARGS sc.blob, sc.blob_words, sc.tex, light_samples, shadow, energy, energy_stride, seg_cap, count_in

// `env`: the scene can produce environment rays (env_sampling_probability > 0); the sweep forms without them are leaner
void launch_shadow(const LaunchCfg& c, int form, int nl, bool env, const SceneArgs& sc, uint32_t light_samples, Queue shadow, float* energy, uint32_t energy_stride,
                   uint32_t seg_cap, const uint32_t* count_in, uint32_t* park, Queue idle_hits) {
    const bool hero = nl == 4;
    // the static parked forms keep their waves' lists of live rays in dynamic LDS behind the staged blob: block threads, blob bytes -> total bytes, list offset in words
    auto parked_lds = [&](int block, uint32_t blob_bytes, uint32_t* live_off) {
        const uint32_t at = (blob_bytes + 15u) & ~15u;
        *live_off = at / 4u;
        return at + (uint32_t)(block / 64) * live_cap(light_samples) * 4u;
    };
#define PT_PARKED_BY_MODE(K) do { uint32_t live_off; const uint32_t bytes = parked_lds(kBlock, c.lds_bytes, &live_off); \
        if (c.lds_mode == PT_LDS_ALL) go_block(c, kBlock, bytes, K(PT_LDS_ALL), PT_ARGS, park, c.walk_policy, live_off); \
        else if (c.lds_mode == PT_LDS_CORE) go_block(c, kBlock, bytes, K(PT_LDS_CORE), PT_ARGS, park, c.walk_policy, live_off); \
        else go_block(c, kBlock, bytes, K(PT_LDS_NONE), PT_ARGS, park, c.walk_policy, live_off); } while (0)
#ifdef PT_EXPERIMENTS
    if (const char* v = getenv("PT_AMD_EXP_SHADOW")) {
        const int e = atoi(v);
#define PT_EXP_CASE(E) if (e == E) { static bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(k_shadow_exp<E>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536), true); (void)once; \
                                     PT_GO(k_shadow_exp<E>, sc.blob, sc.blob_words, sc.tex, light_samples, shadow, idle_hits, seg_cap, count_in); }
        PT_EXP_CASE(8) PT_EXP_CASE(2) PT_EXP_CASE(4) PT_EXP_CASE(0)
    }
#endif
    if (form == PT_FORM_PARKED && c.unit_counter) {
        LaunchCfg d = c; d.grid = c.dyn_grid;
#define PT_DYN_ARGS sc.blob, sc.blob_words, sc.tex, light_samples, shadow, seg_cap, count_in, park, (uint32_t)c.grid, c.unit_counter, c.walk_policy
#define PT_DYN_BY_MODE(NL) do { if (c.lds_mode == PT_LDS_ALL) go(d, k_shadow_parked_dyn<PT_LDS_ALL, NL>, PT_DYN_ARGS); else if (c.lds_mode == PT_LDS_CORE) go(d, k_shadow_parked_dyn<PT_LDS_CORE, NL>, PT_DYN_ARGS); \
                                else go(d, k_shadow_parked_dyn<PT_LDS_NONE, NL>, PT_DYN_ARGS); } while (0)
        LaunchCfg plain = c; plain.lds_bytes = 0;
        if (hero) { PT_DYN_BY_MODE(4); go(plain, k_shadow_sum<4>, light_samples, shadow, energy, energy_stride, seg_cap, count_in); }
        else { PT_DYN_BY_MODE(1); go(plain, k_shadow_sum<1>, light_samples, shadow, energy, energy_stride, seg_cap, count_in); }
    } else if (form == PT_FORM_PARKED_WALK) {   // (one general form per wavelength count: with the scan, with the light list)
        if (hero) PT_PARKED_BY_MODE(K_SH_PARKED4W); else PT_PARKED_BY_MODE(K_SH_PARKED1W);
        PT_TL_BUMP(c.stream);
    } else if (form == PT_FORM_PARKED && !hero && (c.park_block == 512 || c.park_block == 1024)) {   // (the whole blob staged by bigger workgroups)
        uint32_t live_off;
        const uint32_t bytes = parked_lds(c.park_block, c.park_blob_bytes, &live_off);
        if (c.park_block == 512) go_block(c, 512, bytes, k_shadow_parked<PT_LDS_ALL, 1, 0u, 0, 512>, PT_ARGS, park, c.walk_policy, live_off);
        else go_block(c, 1024, bytes, k_shadow_parked<PT_LDS_ALL, 1, 0u, 0, 1024>, PT_ARGS, park, c.walk_policy, live_off);
        PT_TL_BUMP(c.stream);
    } else if (form == PT_FORM_PARKED) {
        if (hero) PT_PARKED_BY_MODE(K_SH_PARKED4);
        else if (c.lacks & PT_SCENE_NO_LIGHTS) PT_PARKED_BY_MODE(K_SH_PARKED1E);   // (an environment is the scene's only emitter)
        else PT_PARKED_BY_MODE(K_SH_PARKED1);
        PT_TL_BUMP(c.stream);
    }
#ifdef PT_EXPERIMENTS
    else if (form == PT_FORM_POOLED) {
        if (env) { if (hero) PT_GO((k_shadow_pooled<PT_LDS_ALL, 4, true>), PT_ARGS); else PT_GO((k_shadow_pooled<PT_LDS_ALL, 1, true>), PT_ARGS); }
        else if (hero) PT_GO((k_shadow_pooled<PT_LDS_ALL, 4, false>), PT_ARGS); else PT_GO((k_shadow_pooled<PT_LDS_ALL, 1, false>), PT_ARGS);
    }
#endif
#ifdef PT_EXPERIMENTS
    else if (form == PT_FORM_SWEEP && c.live_lists) {   // (the wave's list: three words per entry behind the blob)
        const uint32_t at = (c.lds_bytes + 15u) & ~15u, live_off = at / 4u, bytes = at + (uint32_t)(kBlock / 64) * live_cap(light_samples) * 12u;
#define PT_LIVE(NLv, ENVv, LACKSv) go_block(c, kBlock, bytes, k_shadow_live<PT_LDS_ALL, NLv, PT_TRAV_SWEEP, ENVv, LACKSv>, PT_ARGS, live_off)
        if (c.lacks & PT_SCENE_NO_XF) {
            if (env) { if (hero) PT_LIVE(4, true, PT_SCENE_NO_XF); else PT_LIVE(1, true, PT_SCENE_NO_XF); }
            else if (hero) PT_LIVE(4, false, PT_SCENE_NO_XF); else PT_LIVE(1, false, PT_SCENE_NO_XF);
        } else {
            if (env) { if (hero) PT_LIVE(4, true, 0u); else PT_LIVE(1, true, 0u); }
            else if (hero) PT_LIVE(4, false, 0u); else PT_LIVE(1, false, 0u);
        }
    }
#endif
    else if (form == PT_FORM_SWEEP && (c.lacks & PT_SCENE_NO_XF)) {
        if (env) { if (hero) PT_GO((k_shadow<PT_LDS_ALL, 4, PT_TRAV_SWEEP, true, PT_SCENE_NO_XF>), PT_ARGS); else PT_GO((k_shadow<PT_LDS_ALL, 1, PT_TRAV_SWEEP, true, PT_SCENE_NO_XF>), PT_ARGS); }
        else if (hero) PT_GO((k_shadow<PT_LDS_ALL, 4, PT_TRAV_SWEEP, false, PT_SCENE_NO_XF>), PT_ARGS); else PT_GO((k_shadow<PT_LDS_ALL, 1, PT_TRAV_SWEEP, false, PT_SCENE_NO_XF>), PT_ARGS);
    } else if (form == PT_FORM_SWEEP) {
        if (env) { if (hero) PT_GO((k_shadow<PT_LDS_ALL, 4, PT_TRAV_SWEEP, true>), PT_ARGS); else PT_GO((k_shadow<PT_LDS_ALL, 1, PT_TRAV_SWEEP, true>), PT_ARGS); }
        else if (hero) PT_GO((k_shadow<PT_LDS_ALL, 4, PT_TRAV_SWEEP, false>), PT_ARGS); else PT_GO((k_shadow<PT_LDS_ALL, 1, PT_TRAV_SWEEP, false>), PT_ARGS);
    } else if (form == PT_FORM_WALK) { if (hero) PT_GO((k_shadow<PT_LDS_ALL, 4, PT_TRAV_WALK>), PT_ARGS); else PT_GO((k_shadow<PT_LDS_ALL, 1, PT_TRAV_WALK>), PT_ARGS); }
    else if (hero) PT_BY_MODE(K_SH_ANY4, PT_ARGS); else PT_BY_MODE(K_SH_ANY1, PT_ARGS);
}

hipError_t allow_lds_shadow(uint32_t bytes) {
    hipError_t worst = hipSuccess;
    auto allow = [&](const void* k) { hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); if (e != hipSuccess) worst = e; };
#define PT_ALLOW_MODES(K) allow(reinterpret_cast<const void*>(K(PT_LDS_ALL))); allow(reinterpret_cast<const void*>(K(PT_LDS_CORE)))
    PT_ALLOW_MODES(K_SH_ANY1); PT_ALLOW_MODES(K_SH_ANY4); PT_ALLOW_MODES(K_SH_PARKED1); PT_ALLOW_MODES(K_SH_PARKED4); PT_ALLOW_MODES(K_SH_PARKED1E); PT_ALLOW_MODES(K_SH_PARKED1W); PT_ALLOW_MODES(K_SH_PARKED4W);
#define K_SH_DYN1(M) k_shadow_parked_dyn<M, 1>
#define K_SH_DYN4(M) k_shadow_parked_dyn<M, 4>
    PT_ALLOW_MODES(K_SH_DYN1); PT_ALLOW_MODES(K_SH_DYN4);
#ifdef PT_EXPERIMENTS
#define PT_ALLOW_LIVE(NLv, ENVv) allow(reinterpret_cast<const void*>(k_shadow_live<PT_LDS_ALL, NLv, PT_TRAV_SWEEP, ENVv, 0u>)); allow(reinterpret_cast<const void*>(k_shadow_live<PT_LDS_ALL, NLv, PT_TRAV_SWEEP, ENVv, PT_SCENE_NO_XF>))
    PT_ALLOW_LIVE(1, true); PT_ALLOW_LIVE(1, false); PT_ALLOW_LIVE(4, true); PT_ALLOW_LIVE(4, false);
#endif
    allow(reinterpret_cast<const void*>(k_shadow_parked<PT_LDS_ALL, 1, 0u, 0, 512>)); allow(reinterpret_cast<const void*>(k_shadow_parked<PT_LDS_ALL, 1, 0u, 0, 1024>));
    { // (the parked forms without a staged blob still keep their live lists in dynamic LDS)
        allow(reinterpret_cast<const void*>(K_SH_PARKED1(PT_LDS_NONE))); allow(reinterpret_cast<const void*>(K_SH_PARKED4(PT_LDS_NONE))); allow(reinterpret_cast<const void*>(K_SH_PARKED1E(PT_LDS_NONE)));
        allow(reinterpret_cast<const void*>(K_SH_PARKED1W(PT_LDS_NONE))); allow(reinterpret_cast<const void*>(K_SH_PARKED4W(PT_LDS_NONE))); }
    allow(reinterpret_cast<const void*>(k_shadow<PT_LDS_ALL, 1, PT_TRAV_WALK>)); allow(reinterpret_cast<const void*>(k_shadow<PT_LDS_ALL, 4, PT_TRAV_WALK>));
    allow(reinterpret_cast<const void*>(k_shadow<PT_LDS_ALL, 1, PT_TRAV_SWEEP, true>)); allow(reinterpret_cast<const void*>(k_shadow<PT_LDS_ALL, 4, PT_TRAV_SWEEP, true>));
    allow(reinterpret_cast<const void*>(k_shadow<PT_LDS_ALL, 1, PT_TRAV_SWEEP, false>)); allow(reinterpret_cast<const void*>(k_shadow<PT_LDS_ALL, 4, PT_TRAV_SWEEP, false>));
    allow(reinterpret_cast<const void*>(k_shadow<PT_LDS_ALL, 1, PT_TRAV_SWEEP, true, PT_SCENE_NO_XF>)); allow(reinterpret_cast<const void*>(k_shadow<PT_LDS_ALL, 4, PT_TRAV_SWEEP, true, PT_SCENE_NO_XF>));
    allow(reinterpret_cast<const void*>(k_shadow<PT_LDS_ALL, 1, PT_TRAV_SWEEP, false, PT_SCENE_NO_XF>)); allow(reinterpret_cast<const void*>(k_shadow<PT_LDS_ALL, 4, PT_TRAV_SWEEP, false, PT_SCENE_NO_XF>));
#ifdef PT_EXPERIMENTS
    allow(reinterpret_cast<const void*>(k_shadow_pooled<PT_LDS_ALL, 1, false>)); allow(reinterpret_cast<const void*>(k_shadow_pooled<PT_LDS_ALL, 4, false>));
    allow(reinterpret_cast<const void*>(k_shadow_pooled<PT_LDS_ALL, 1, true>)); allow(reinterpret_cast<const void*>(k_shadow_pooled<PT_LDS_ALL, 4, true>));
#endif
    return worst;
}

#ifdef PT_TIMELINE
PT_TL_ACCESSOR(pt_debug_timeline_shadow)
// (the walks of more than 1500 box tests that pt_device.h recorded in this family's kernels: count, then 12 floats per ray)
extern "C" int pt_debug_long_walks(float* out, size_t bytes) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ptd::g_tl_rays), bytes < sizeof(ptd::g_tl_rays) ? bytes : sizeof(ptd::g_tl_rays)) != hipSuccess) return 2;
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(ptd::g_tl_rays)) != hipSuccess || hipMemset(p, 0, sizeof(ptd::g_tl_rays)) != hipSuccess) return 3;
    return 0;
}
#endif

}  // namespace ptk
