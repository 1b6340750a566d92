// pt_kern_shade.hip — the vertex kernels (k_shade: 3 staging modes x 1 or 4 wavelengths x 3 forms) and their launcher.
// PT_SHADE_NL (1 | 4 wavelengths per path) and PT_SHADE_PART (0 = the lean forms, 1 = NO_ENV, FULL, medium) select the quarter of the family this translation unit
// holds: the build compiles the four side by side, and gives part 1 — the forms of scenes with meshes in L1/L2 and importance-map tables — PT_QUEUE_NT (queue words
// non-temporal, so that the caches keep what is read again; it does not pay for the lean forms, whose scenes live in LDS).
#ifndef PT_SHADE_PART
#error "PT_SHADE_PART must be 0 (lean forms) or 1 (NO_ENV, FULL, medium)"
#endif
#include <cstdio>
#include <cstdlib>
#include "pt_kernels.h"

namespace ptk {

#define PT_GO(K, ...) go(c, K, __VA_ARGS__)
#define PT_BY_MODE(K, ...) do { if (c.lds_mode == PT_LDS_ALL) PT_GO(K(PT_LDS_ALL), __VA_ARGS__); else if (c.lds_mode == PT_LDS_CORE) PT_GO(K(PT_LDS_CORE), __VA_ARGS__); \
                                else PT_GO(K(PT_LDS_NONE), __VA_ARGS__); } while (0)
// (PT_SCENE_NO_CERTS: the certificate code compiled out — every form but the two a scene WITH a convex-body certificate takes, K_SHADE_NC / K_SHADE_FC: the engine gives such a
// scene at least the NO_ENV form, never the lean, fused or light-free ones)
#define K_SHADE_L(M) k_shade<M, PT_SHADE_NL, PT_SHADE_LEAN, PT_SCENE_NO_CERTS>
#define K_SHADE_LX(M) k_shade<M, PT_SHADE_NL, PT_SHADE_LEAN, PT_SCENE_NO_XF | PT_SCENE_NO_CERTS>
#define K_SHADE_FUSED k_shade<PT_LDS_ALL, PT_SHADE_NL, PT_SHADE_LEAN, PT_SCENE_NO_XF | PT_SCENE_NO_CERTS, PT_TRAV_SWEEP>
#define K_SHADE_N(M) k_shade<M, PT_SHADE_NL, PT_SHADE_NO_ENV, PT_SCENE_NO_CERTS>
#define K_SHADE_F(M) k_shade<M, PT_SHADE_NL, PT_SHADE_FULL, PT_SCENE_NO_CERTS>
#define K_SHADE_FE(M) k_shade<M, PT_SHADE_NL, PT_SHADE_FULL, PT_SCENE_NO_LIGHTS | PT_SCENE_NO_CERTS>
#define K_SHADE_NC(M) k_shade<M, PT_SHADE_NL, PT_SHADE_NO_ENV>
#define K_SHADE_FC(M) k_shade<M, PT_SHADE_NL, PT_SHADE_FULL>
#define PT_ARGS sc.blob, sc.blob_words, sc.tex, rp, bounce, pixels, paths_in, hits, paths_out, shadow, energy, seg_cap, count_in, count_out, shadow_count, block_stats
#define PT_ARGS_FWD sc, rp, bounce, pixels, paths_in, hits, paths_out, shadow, energy, seg_cap, count_in, count_out, shadow_count, block_stats
#define PT_CAT2(a, b) a##b
#define PT_CAT(a, b) PT_CAT2(a, b)

#define PT_PARTNAME(base) PT_CAT(PT_CAT(PT_CAT(base, PT_SHADE_NL), _p), PT_SHADE_PART)
void PT_PARTNAME(launch_shade_nl)(const LaunchCfg& c, int form, const SceneArgs& sc, const RenderParams& rp, uint32_t bounce, const uint32_t* pixels, Queue paths_in,
                                          Queue hits, Queue paths_out, Queue shadow, float* energy, uint32_t seg_cap, const uint32_t* count_in, uint32_t* count_out,
                                          uint32_t* shadow_count, unsigned long long* block_stats) {
#if PT_SHADE_PART == 1
#if PT_SHADE_NL == 1
#define K_SHADE_M(M) k_shade_medium<M>
    if (form == PT_SHADE_MEDIUM) { PT_BY_MODE(K_SHADE_M, PT_ARGS); return; }
#endif
    if (form == PT_SHADE_FULL) {   // (the marginal tables of the importance map behind the blob: stage_marginal)
        LaunchCfg d = c; d.lds_bytes = ((c.lds_bytes + 15u) & ~15u) + sc.marg_bytes;
#define PT_GO_D(K, ...) go(d, K, __VA_ARGS__)
#define PT_BY_MODE_D(K, ...) do { if (c.lds_mode == PT_LDS_ALL) PT_GO_D(K(PT_LDS_ALL), __VA_ARGS__); else if (c.lds_mode == PT_LDS_CORE) PT_GO_D(K(PT_LDS_CORE), __VA_ARGS__); \
                                  else PT_GO_D(K(PT_LDS_NONE), __VA_ARGS__); } while (0)
        if (c.certs) PT_BY_MODE_D(K_SHADE_FC, PT_ARGS);
        else if ((c.lacks & PT_SCENE_NO_LIGHTS) && PT_SHADE_NL == 1) PT_BY_MODE_D(K_SHADE_FE, PT_ARGS);   // (an environment is the scene's only emitter)
        else PT_BY_MODE_D(K_SHADE_F, PT_ARGS);
    }
    else if (c.certs) PT_BY_MODE(K_SHADE_NC, PT_ARGS);
    else PT_BY_MODE(K_SHADE_N, PT_ARGS);
#else
    (void)form;
    if (c.fuse) PT_GO(K_SHADE_FUSED, PT_ARGS);   // (the engine asks for it only where this form exists: PT_LDS_ALL, lean, no transforms, pure sweep)
    else if (c.lacks & PT_SCENE_NO_XF) PT_BY_MODE(K_SHADE_LX, PT_ARGS);
    else PT_BY_MODE(K_SHADE_L, PT_ARGS);
#endif
    PT_TL_BUMP(c.stream);
}
hipError_t PT_PARTNAME(allow_lds_shade_nl)(uint32_t bytes) {
    hipError_t worst = hipSuccess;
    auto allow = [&](const void* k) { hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); if (e != hipSuccess) worst = e; };
#define PT_ALLOW_MODES(K) allow(reinterpret_cast<const void*>(K(PT_LDS_ALL))); allow(reinterpret_cast<const void*>(K(PT_LDS_CORE))); allow(reinterpret_cast<const void*>(K(PT_LDS_NONE)))
#if PT_SHADE_PART == 0
    allow(reinterpret_cast<const void*>(K_SHADE_FUSED));
    PT_ALLOW_MODES(K_SHADE_L); PT_ALLOW_MODES(K_SHADE_LX);
#else
    PT_ALLOW_MODES(K_SHADE_N); PT_ALLOW_MODES(K_SHADE_F); PT_ALLOW_MODES(K_SHADE_NC); PT_ALLOW_MODES(K_SHADE_FC);
#if PT_SHADE_NL == 1
    PT_ALLOW_MODES(K_SHADE_FE);
    PT_ALLOW_MODES(K_SHADE_M);
#endif
#endif
    return worst;
}

#if PT_SHADE_NL == 1 && PT_SHADE_PART == 0
#define PT_SHADE_SIG const LaunchCfg&, int, const SceneArgs&, const RenderParams&, uint32_t, const uint32_t*, Queue, Queue, Queue, Queue, float*, uint32_t, const uint32_t*, uint32_t*, uint32_t*, unsigned long long*
void launch_shade_nl1_p1(PT_SHADE_SIG); void launch_shade_nl4_p0(PT_SHADE_SIG); void launch_shade_nl4_p1(PT_SHADE_SIG);
hipError_t allow_lds_shade_nl1_p1(uint32_t); hipError_t allow_lds_shade_nl4_p0(uint32_t); hipError_t allow_lds_shade_nl4_p1(uint32_t);
void launch_shade(const LaunchCfg& c, int nl, int form, const SceneArgs& sc, const RenderParams& rp, uint32_t bounce, const uint32_t* pixels, Queue paths_in,
                  Queue hits, Queue paths_out, Queue shadow, float* energy, uint32_t seg_cap, const uint32_t* count_in, uint32_t* count_out,
                  uint32_t* shadow_count, unsigned long long* block_stats) {
    const bool lean = form == PT_SHADE_LEAN;
    // (round-5 advisor) k_generate writes the camera vertex' lean record when rp.camera_record is set; k_shade's forms read it at bounce 0 by their FORM alone (the uniform branch
    // on rp.camera_record costs the NO_ENV form 7 %, pt_kernels.h).  The two sides must agree: a FORM / camera_record pairing that does not would read nine stale words per path.
    if (c.certs && form == PT_SHADE_LEAN) { fprintf(stderr, "launch_shade: a scene with convex-body certificates takes the NO_ENV or FULL form\n"); abort(); }
    if ((rp.camera_record != 0u) != (form != PT_SHADE_NO_ENV)) { fprintf(stderr, "launch_shade: form %d launched with camera_record = %u\n", form, rp.camera_record); abort(); }
    if (nl == 4) { if (lean) launch_shade_nl4_p0(c, form, PT_ARGS_FWD); else launch_shade_nl4_p1(c, form, PT_ARGS_FWD); }
    else if (lean) launch_shade_nl1_p0(c, form, PT_ARGS_FWD); else launch_shade_nl1_p1(c, form, PT_ARGS_FWD);
}
hipError_t allow_lds_shade(uint32_t bytes) {
    hipError_t r = hipSuccess;
    for (hipError_t e : {allow_lds_shade_nl1_p0(bytes), allow_lds_shade_nl1_p1(bytes), allow_lds_shade_nl4_p0(bytes), allow_lds_shade_nl4_p1(bytes)}) if (e != hipSuccess) r = e;
    return r;
}
#endif
#if defined(PT_TIMELINE) && PT_SHADE_NL == 1 && PT_SHADE_PART == 1
PT_TL_ACCESSOR(pt_debug_timeline_shade)   // (the one-wavelength general forms' records: the scenes the timeline was built for)
#endif

}  // namespace ptk
