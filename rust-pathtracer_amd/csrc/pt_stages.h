// pt_stages.h — what one lane does in each stage of the wavefront path tracer.
//
// The reference runs one recursive-free but strictly sequential loop per camera sample
// (PathTracingIntegrator::color, src/integrator/pt.rs:397-615, calling random_walk,
// src/integrator/utils.rs:152-376).  Here the same work is cut into stages that run as separate kernels
// over dense SoA queues in HBM:
//
//   generate  : jitter, wavelength, thin-lens camera ray                       (tiled.rs:366-380, pt.rs:404-417)
//   extend    : closest hit of the path segment                                (utils.rs:171 -> World::hit)
//   shade     : everything the walk and the second pass do at one vertex       (utils.rs:172-329, pt.rs:481-604)
//               emission + MIS for light / environment vertices, BSDF sample, roulette, next ray,
//               and the set-up of the light samples (light pick, Hittable::sample, bsdf, weight)
//   shadow    : closest hit of the light-sample rays + emission at the hit     (pt.rs:171-217, 252-330)
//   accumulate: energy -> CIE XYZ, summed into the film in the reference's order (pt.rs:614, tiled.rs:366-398)
//
// NEE is done at the bounce that creates the vertex instead of in a second pass over a vertex list: every
// light sample only needs that vertex's hit, wi, material and the throughput *before* the bounce
// (utils.rs:177-193), and random numbers are addressed by (bounce, light sample), so nothing changes.
#ifndef PT_STAGES_H
#define PT_STAGES_H

#include "pt_device.h"

namespace ptd {

// ---- SoA queues.  Field k of item i lives at base[k * capacity + i]: every field is a dense, coalesced stream.
// NL = wavelengths carried per path: 1 (the reference's live integrator) or 4 (hero-wavelength variant, BASELINE C5:
// lambda_0 drives every decision, three passengers carry their own throughput; definition in oracle/ptref.cpp).
enum { PS_OX, PS_OY, PS_OZ, PS_DX, PS_DY, PS_DZ, PS_BETA, PS_LAMBDA, PS_SLOT, PS_PREV_PDF, PS_PNX, PS_PNY, PS_PNZ, PS_PPX, PS_PPY, PS_PPZ, PS_FIELDS };
enum { HS_T, HS_PX, HS_PY, HS_PZ, HS_NX, HS_NY, HS_NZ, HS_U, HS_V, HS_MAT, HS_INST, HS_FIELDS };
enum { SR_OX, SR_OY, SR_OZ, SR_DX, SR_DY, SR_DZ, SR_FACTOR };
#define PT_MAX_LIGHT_SAMPLES 8
template <int NL> struct Layout {
    static constexpr uint32_t path_fields = PS_FIELDS + (NL - 1);  // passenger throughputs at PS_FIELDS + k - 1
    // light-sample item: slot, lambda[NL], flags (bit l: sub-ray l is an environment sample), then per light sample 6 + NL floats
    static constexpr uint32_t sh_slot = 0, sh_lambda = 1, sh_flags = 1 + NL, sh_head = 2 + NL, sr_fields = 6 + NL;
    static constexpr uint32_t shadow_fields(uint32_t light_samples) { return sh_head + light_samples * sr_fields; }
    // One more field behind the rays: the LIST of the segment's live items — entry p of the segment holds the position of its p-th item that has a ray with a
    // non-zero factor.  An item whose rays are all dead (a vertex next to the lamp: every sampled direction below its horizon — 10 % of C2's items) adds 0 to
    // its slot whether or not it is read: the light-sample kernel walks the list, every lane on a live item, and never touches the others.
    static constexpr uint32_t shadow_live_field(uint32_t light_samples) { return shadow_fields(light_samples); }
    static constexpr uint32_t shadow_queue_fields(uint32_t light_samples) { return shadow_fields(light_samples) + 1u; }
};

struct Queue { uint32_t* base; uint32_t capacity; uint32_t fields; };
#ifdef PT_QUEUE_SOA
PT_HD size_t qindex(const Queue& q, uint32_t field, uint32_t i) { return (size_t)field * q.capacity + i; }
PT_HD size_t qstride(const Queue& q) { return q.capacity; }   // words from one field of an item to the next
#else
// Tiles of 64 items: field k of item i at base[((i / 64) * fields + k) * 64 + i % 64].  A wave still reads or writes 256 contiguous bytes per
// field, and all fields of an item lie within fields * 256 bytes of one address: one 64-bit address per item and queue, the field as the
// instruction's immediate offset (the field-major layout kept a 64-bit address or an add per field alive: 13 VGPRs in k_shadow).
// (tile < 2^24 since capacity <= 2^30, fields < 2^24: masked operands let the compiler take the full-rate 24-bit multiply)
PT_HD uint32_t qtile(uint32_t tile, uint32_t fields) { return (tile & 0xffffffu) * (fields & 0xffffffu); }
PT_HD size_t qindex(const Queue& q, uint32_t field, uint32_t i) { return ((size_t)qtile(i >> 6, q.fields) << 6) + (i & 63u) + ((size_t)field << 6); }
PT_HD size_t qstride(const Queue&) { return 64; }   // words from one field of an item to the next
#endif
// Queue words are read once by one lane and written once by one lane: streamed.  PT_QUEUE_NT marks their loads and stores non-temporal (the `nt` cache policy: first to be
// evicted), so that the caches keep what IS read again — mesh nodes and triangles, importance-map rows, texels (measured: DESIGN.md section 6).
#if defined(PT_QUEUE_NT) && PT_QUEUE_NT && defined(__HIP_DEVICE_COMPILE__)
#define PT_QLOAD(p) __builtin_nontemporal_load(p)
#define PT_QSTORE(v, p) __builtin_nontemporal_store((v), (p))
#else
#define PT_QLOAD(p) (*(p))
#define PT_QSTORE(v, p) (*(p) = (v))
#endif
// (a load that is non-temporal whatever PT_QUEUE_NT says: a light-sample ray's origin and direction in the parked kernels, read once — PT_PARKED_NT_RAY)
#if defined(__HIP_DEVICE_COMPILE__)
#define PT_QLOAD_NT(p) __builtin_nontemporal_load(p)
#else
#define PT_QLOAD_NT(p) (*(p))
#endif
PT_HD float qf(const Queue& q, uint32_t field, uint32_t i) { return pt_u2f(PT_QLOAD(&q.base[qindex(q, field, i)])); }
PT_HD uint32_t qu(const Queue& q, uint32_t field, uint32_t i) { return PT_QLOAD(&q.base[qindex(q, field, i)]); }
PT_HD void qsf(const Queue& q, uint32_t field, uint32_t i, float v) { PT_QSTORE(pt_f2u(v), &q.base[qindex(q, field, i)]); }
PT_HD void qsu(const Queue& q, uint32_t field, uint32_t i, uint32_t v) { PT_QSTORE(v, &q.base[qindex(q, field, i)]); }

struct RenderParams {
    uint64_t seed;
    uint32_t width, height;
    uint32_t min_bounces, max_bounces, light_samples, only_direct;
    float wavelength_lo, wavelength_span;
    uint32_t chunk_pixels;      // pixels in this pass (slot = s_local * chunk_pixels + p)
    uint32_t first_sample;      // absolute sample index of s_local = 0
    uint32_t pass_samples;      // samples per pixel in this pass
    uint32_t spp;               // RenderSettings::min_samples
    uint32_t range_end;         // first_sample + sample_count of the whole call
    uint32_t normalize;         // divide by spp when the last phase of a whole render is flushed
    uint32_t energy_stride;     // energy of wavelength k of slot i at energy[k * energy_stride + i]
    uint32_t phase;             // samples per partial sum (10: tiled.rs:347-361; spp: naive.rs:82-103)
    uint32_t camera_record;     // k_generate writes the camera vertex' lean record (store_path_camera): the render's vertex kernel rebuilds the rest at bounce 0
    uint32_t live_list;         // the light-sample kernel of this render walks the list of live items (Layout::shadow_live_field): the vertex kernel builds it
    CameraParams camera;
};

// lambda_k = lo + frac(u + k/4) * span; lambda_0 is the single-wavelength render's wavelength (pt.rs:406)
template <int NL>
PT_HD void hero_lambdas(const RenderParams& rp, float u, float* lambda) {
    lambda[0] = rp.wavelength_lo + u * rp.wavelength_span;
    for (int k = 1; k < NL; ++k) {
        float x = u + (float)k * 0.25f;
        x = x - pt_floor(x);
        lambda[k] = rp.wavelength_lo + x * rp.wavelength_span;
    }
}

template <int NL>
struct PathVertexT {  // the register-resident state of one path between stages
    F3 o, d; float beta[NL]; float lambda; uint32_t slot; float prev_pdf; F3 prev_n, prev_p;
};
template <int NL>
PT_HD PathVertexT<NL> load_path(const Queue& q, uint32_t i) {
    PathVertexT<NL> p;
    p.o = f3(qf(q, PS_OX, i), qf(q, PS_OY, i), qf(q, PS_OZ, i));
    p.d = f3(qf(q, PS_DX, i), qf(q, PS_DY, i), qf(q, PS_DZ, i));
    p.beta[0] = qf(q, PS_BETA, i); p.lambda = qf(q, PS_LAMBDA, i); p.slot = qu(q, PS_SLOT, i); p.prev_pdf = qf(q, PS_PREV_PDF, i);   // (its SIGN is a mark for the closest-hit kernel, PT_HDR_CONVEX_INST: every reader below squares the value — the MIS weights of a light or
                                                                                                                  // environment vertex — so the sign never reaches a result; taking the magnitude here cost the fused form an instruction it does not need)
    for (int k = 1; k < NL; ++k) p.beta[k] = qf(q, PS_FIELDS + k - 1, i);
    p.prev_n = f3(qf(q, PS_PNX, i), qf(q, PS_PNY, i), qf(q, PS_PNZ, i));
    p.prev_p = f3(qf(q, PS_PPX, i), qf(q, PS_PPY, i), qf(q, PS_PPZ, i));
    return p;
}
// The camera vertex' record (round 5): of the sixteen words stage_generate fills, nine are functions of the other seven and of the item's index — throughput 1, slot = the
// index (k_generate stores slot s at index s), previous pdf 100, previous normal = the direction, previous point = the origin (pt.rs:430-446).  k_generate writes the ray and the
// wavelength alone (28 of 64 bytes: it is bound by those writes, 5.7 TB/s) and the first bounce's vertex kernel rebuilds the rest instead of reading it: the same values.
#ifndef PT_STORED_WAVELENGTH
#define PT_STORED_WAVELENGTH 1   /* k_generate leaves each sample's wavelength sample behind the energy planes and k_accumulate reads it (round 5); 0: k_accumulate draws it again */
#endif
#ifndef PT_CAMERA_RECORD
#define PT_CAMERA_RECORD 1   /* 0: the full record written and read at every bounce (rounds 1-4) */
#endif
template <int NL>
PT_HD void store_path_camera(const Queue& q, uint32_t i, const PathVertexT<NL>& p) {
    qsf(q, PS_OX, i, p.o.x); qsf(q, PS_OY, i, p.o.y); qsf(q, PS_OZ, i, p.o.z);
    qsf(q, PS_DX, i, p.d.x); qsf(q, PS_DY, i, p.d.y); qsf(q, PS_DZ, i, p.d.z);
    qsf(q, PS_LAMBDA, i, p.lambda);
}
// `first`: the launch shades camera vertices (bounce 0, wave-uniform)
template <int NL>
PT_HD PathVertexT<NL> load_path(const Queue& q, uint32_t i, bool first) {
    if (!(PT_CAMERA_RECORD && first)) return load_path<NL>(q, i);
    PathVertexT<NL> p;
    p.o = f3(qf(q, PS_OX, i), qf(q, PS_OY, i), qf(q, PS_OZ, i));
    p.d = f3(qf(q, PS_DX, i), qf(q, PS_DY, i), qf(q, PS_DZ, i));
    p.lambda = qf(q, PS_LAMBDA, i);
    for (int k = 0; k < NL; ++k) p.beta[k] = 1.0f;
    p.slot = i; p.prev_pdf = 100.0f; p.prev_n = p.d; p.prev_p = p.o;
    return p;
}
template <int NL>
PT_HD void store_path(const Queue& q, uint32_t i, const PathVertexT<NL>& p) {
    qsf(q, PS_OX, i, p.o.x); qsf(q, PS_OY, i, p.o.y); qsf(q, PS_OZ, i, p.o.z);
    qsf(q, PS_DX, i, p.d.x); qsf(q, PS_DY, i, p.d.y); qsf(q, PS_DZ, i, p.d.z);
    qsf(q, PS_BETA, i, p.beta[0]); qsf(q, PS_LAMBDA, i, p.lambda); qsu(q, PS_SLOT, i, p.slot); qsf(q, PS_PREV_PDF, i, p.prev_pdf);
    for (int k = 1; k < NL; ++k) qsf(q, PS_FIELDS + k - 1, i, p.beta[k]);
    qsf(q, PS_PNX, i, p.prev_n.x); qsf(q, PS_PNY, i, p.prev_n.y); qsf(q, PS_PNZ, i, p.prev_n.z);
    qsf(q, PS_PPX, i, p.prev_p.x); qsf(q, PS_PPY, i, p.prev_p.y); qsf(q, PS_PPZ, i, p.prev_p.z);
}
PT_HD void store_hit(const Queue& q, uint32_t i, const Hit& h) {
    qsf(q, HS_T, i, h.valid ? h.t : -1.0f);
    if (!h.valid) return;
    qsf(q, HS_PX, i, h.p.x); qsf(q, HS_PY, i, h.p.y); qsf(q, HS_PZ, i, h.p.z);
    qsf(q, HS_NX, i, h.n.x); qsf(q, HS_NY, i, h.n.y); qsf(q, HS_NZ, i, h.n.z);
    qsf(q, HS_U, i, h.u); qsf(q, HS_V, i, h.v); qsu(q, HS_MAT, i, h.material); qsu(q, HS_INST, i, h.instance);
}
// EAGER: every field is read at once, whatever the first word says (a miss's other fields hold stale words that nothing uses) — the reads
// overlap instead of waiting for the first; for scenes where nearly every segment hits something.
template <bool EAGER = false>
PT_HD Hit load_hit(const Queue& q, uint32_t i) {
    Hit h; h.t = qf(q, HS_T, i); h.valid = h.t >= 0.0f;
    if (!EAGER && !h.valid) return h;
    h.p = f3(qf(q, HS_PX, i), qf(q, HS_PY, i), qf(q, HS_PZ, i));
    h.n = f3(qf(q, HS_NX, i), qf(q, HS_NY, i), qf(q, HS_NZ, i));
    h.u = qf(q, HS_U, i); h.v = qf(q, HS_V, i); h.material = qu(q, HS_MAT, i); h.instance = qu(q, HS_INST, i);
    return h;
}

// ------------------------------------------------------------------------------------------------ generate
template <int NL>
PT_HD PathVertexT<NL> stage_generate(const RenderParams& rp, uint32_t slot, uint32_t pixel, float* wavelength_sample = nullptr) {
    uint32_t s_local = slot / rp.chunk_pixels;
    uint32_t sample = rp.first_sample + s_local;
    uint32_t x = pixel % rp.width, y = pixel / rp.width;
    pt_f32x4 fs = pt_draw4(rp.seed, pixel, sample, PT_DIM_FILM);
    float cu = ((float)x + fs.x) / (float)rp.width, cv = ((float)y + fs.y) / (float)rp.height;  // box filter, tiled.rs:372-375
    PathVertexT<NL> p;
    // pt.rs:406.  (round 5) A hero-wavelength path carries the wavelength SAMPLE instead — its four wavelengths are functions of it (hero_lambdas), and a vertex that
    // holds it need not draw the film block again for them (a Philox draw per vertex: C5 k_shade).
    p.lambda = NL > 1 ? fs.z : rp.wavelength_lo + fs.z * rp.wavelength_span;
    if (wavelength_sample != nullptr) *wavelength_sample = fs.z;
    float fu = pt_clamp(cu, 0.0f, 1.0f - PT_F32_EPSILON), fv = pt_clamp(cv, 0.0f, 1.0f - PT_F32_EPSILON);  // pt.rs:411-414
    camera_ray(rp.camera, rp.seed, pixel, sample, fu, fv, &p.o, &p.d);
    for (int k = 0; k < NL; ++k) p.beta[k] = 1.0f;
    p.slot = slot;
    p.prev_pdf = 100.0f; p.prev_n = p.d; p.prev_p = p.o;  // the camera vertex (pt.rs:430-446)
    return p;
}

// ------------------------------------------------------------------------------------------------ shade
template <int NL> struct ShadowRayT { F3 o, d; float factor[NL]; };
template <int NL> PT_HD bool ray_is_live(const ShadowRayT<NL>& r) { bool live = false; for (int k = 0; k < NL; ++k) live = live || (r.factor[k] != 0.0f); return live; }
template <int NL>
struct ShadeOutT {
    bool survives;          // path continues with `next`
    PathVertexT<NL> next;
    float energy_add[NL];   // light / environment vertex contribution
    bool add_energy;
    bool vertex_pushed;     // counts towards Profile::bounce_rays
    bool env_hit;
    uint32_t shadow_count;  // valid sub-rays (Profile::shadow_rays)
    uint32_t env_mask;      // the item's flag word: bit l: sub-ray l is an environment sample; bit 8 + l: sub-ray l cannot hit instance (word >> 16) again (PT_INST_CONVEX_OUT)
    bool has_item;
};

// Whether a hit vertex gets a light-sample work item (known before shading, so the kernel can reserve the queue
// position first and stream the rays straight to HBM): non-light material, L > 0, something to sample (pt.rs:346-348, 584).
PT_HD bool shade_wants_item(const SceneView& s, const RenderParams& rp, const Hit& hit) {
    if (!hit.valid || rp.light_samples == 0) return false;
    if (PT_MATERIAL_TAG(hit.material) == PT_TAG_LIGHT) return false;
    return !(bu(s, PT_HDR_LIGHT_COUNT) == 0 && bf(s, PT_HDR_ENV_PROB) == 0.0f);
}

// One vertex of random_walk (utils.rs:170-373) + the matching iteration of color()'s second pass (pt.rs:481-604).
// `sink(l, ray)` receives every light-sample ray (all factors 0: nothing to trace) when the vertex has an item.
// ENV = false: the caller knows env_sampling_probability == 0 (Cornell-type scenes) — light samples never pick the environment, and
// the whole estimate_direct_illumination_from_world branch is compiled out of the kernel (registers, not results: choose_first with
// probability 0 leaves the sample as it is).
// GGX = false: the scene holds no GGX material (material_prepare<false> and friends).
#ifndef PT_ONE_LIGHT_FORMS
#define PT_ONE_LIGHT_FORMS (!ENV && !GGX)   /* which vertex forms test their light-sample rays against a scene's only light (below) */
#endif
template <int NL, bool ENV = true, bool GGX = true, typename RaySink>
PT_HD ShadeOutT<NL> stage_shade(const SceneView& s, const RenderParams& rp, uint32_t bounce, const PathVertexT<NL>& pv, const Hit& hit,
                                uint32_t pixel, RaySink&& sink) {
    ShadeOutT<NL> out;
    out.survives = false; out.add_energy = false; out.vertex_pushed = false; out.env_hit = false;
    out.shadow_count = 0; out.env_mask = 0; out.has_item = false;
    for (int k = 0; k < NL; ++k) out.energy_add[k] = 0.0f;
    uint32_t sample = rp.first_sample + pv.slot / rp.chunk_pixels;
    bool prev_is_camera = bounce == 0;
    float lam[NL];
    lam[0] = pv.lambda;
    if (NL > 1) hero_lambdas<NL>(rp, pv.lambda, lam);   // (the record holds the wavelength sample, stage_generate)
    const float lambda = lam[0];
    if (!hit.valid) {
        // environment vertex (utils.rs:344-372) and its MIS-weighted emission (pt.rs:487-511)
        F3 wo = pv.d;
        float u = 0.0f, v = 0.0f;
        if (bu(s, PT_HDR_ENV_KIND) != PT_ENV_CONSTANT) direction_to_uv(wo, &u, &v);
        float cos_i = pt_abs(dot(pv.prev_n, wo));
        const EnvPoint ep = env_point(s, u, v);   // (the direction of (u, v) and its texture coordinates: once for the pdf and the emission)
        float nee_psa_pdf = env_pdf_for(s, ep) / pt_abs(cos_i);
        float bsdf_psa_pdf = pv.prev_pdf / pt_abs(cos_i);
        float weight = (bsdf_psa_pdf * bsdf_psa_pdf) / (bsdf_psa_pdf * bsdf_psa_pdf + nee_psa_pdf * nee_psa_pdf);
        PT_ROLLED for (int k = 0; k < NL; ++k) pl_set<NL>(out.energy_add, k, weight * pl_get<NL>(pv.beta, k) * env_emission(s, ep, pl_get<NL>(lam, k)));
        out.add_energy = true;
        out.vertex_pushed = true; out.env_hit = true;
        return out;
    }
    Frame frame = frame_from_normal(hit.n);
    F3 wi = normalize(to_local(frame, neg(pv.d)));
    uint32_t m = material_record(s, hit.material);
    bool is_light = !(s.lacks & PT_SCENE_NO_LIGHTS) && PT_MATERIAL_TAG(hit.material) == PT_TAG_LIGHT;
    // (round 6) the vertex lies on an instance the host certified convex and closed (pt_blob.h PT_INST_CONVEX_*): its outward light-sample rays are marked
    // "cannot hit this instance again" (ShadeOutT::env_mask, bits 8.. and 16..), its inward light rays are dead here, and so is marked the path's next segment if it
    // leaves outward.  A scene without such an instance: one scalar test.
    const bool certs = scene_has_certificates(s);   // (wave-uniform, a scalar: everything the certificates add stands behind it)
    // (three lane predicates made here, so that neither the instance's flag word nor the hit's instance word stays alive through the sampling code below: the NO_ENV form
    // runs at its register cap)
    bool cv_out = false, cv_in = false, cv_path = false, cv_only = false;   // outward rays are marked; inward light rays from this face are dead; the path's next segment is marked too
    float cv_tau = 2.0f;                                    // the face's outward threshold: a direction's cosine to the hit normal must exceed it
    if (certs) {
        PT_KEEP_BRANCH_NOFENCE();
        const uint32_t id = hit.instance & PT_HIT_INDEX_MASK, tq = (hit.instance >> PT_HIT_OUT_SHIFT) & 0x7fffu;
        const uint32_t cf = bu(s, bu(s, PT_HDR_INSTANCE_OFF) + id * PT_INST_WORDS + PT_INST_FLAGS);
        cv_out = (cf & PT_INST_CONVEX_OUT) != 0u && tq != 0u;
        cv_tau = (float)tq * (1.0f / 32768.0f);
        cv_in = (cf & PT_INST_CONVEX_IN) != 0u && (hit.instance & PT_HIT_IN_SAFE) != 0u;
        cv_only = bu(s, PT_HDR_CONVEX_INST) == id + 1u;   // (the scene's one certified body: path marks name no instance)
        cv_path = cv_out && cv_only;
    }
    pt_f32x4 r = pt_draw4(rp.seed, pixel, sample, pt_dim_bounce(bounce, rp.light_samples));
    float f, pdf; F3 wo;
    // (per-wavelength loops stay rolled, their arrays in registers: pl_get / pl_set, pt_device.h)
    const MatEvalN<NL> me = material_prepare_n<NL, GGX>(s, m, lam, hit.u, hit.v);
    const MatEval me0 = material_at<NL>(me, 0);
    material_sample_p<GGX>(me0, r.x, r.y, wi, &f, &wo, &pdf);
    float cos_o = pt_abs(wo.z);
    if (pt_isnan(pdf)) return out;  // utils.rs:261-263: the vertex is never pushed
    float rr = (bounce >= rp.min_bounces) ? pt_min(f / pdf, 1.0f) : 1.0f;
    float pdf_forward = pdf * (rr / cos_o);
    out.vertex_pushed = true;

    if (is_light) {
        // pt.rs:512-561
        float emission = material_emission(s, m, lambda, wi);
        if (emission > 0.0f) {
            if (rp.light_samples == 0 || prev_is_camera) {
                out.energy_add[0] = pv.beta[0] * emission;
                PT_ROLLED for (int k = 1; k < NL; ++k) pl_set<NL>(out.energy_add, k, pl_get<NL>(pv.beta, k) * material_emission(s, m, pl_get<NL>(lam, k), wi));
                out.add_energy = true;
            } else if (!rp.only_direct) {
                F3 nee_dir = normalize(sub(hit.p, pv.prev_p));
                uint32_t inst = bu(s, PT_HDR_INSTANCE_OFF) + hit_instance_index(s, hit.instance) * PT_INST_WORDS;
                float pdfh = light_psa_pdf(s, inst, dot(pv.prev_n, nee_dir), dot(hit.n, nee_dir), pv.prev_p, hit.p);
                float a = pv.prev_pdf;
                float weight = (a * a) / (a * a + pdfh * pdfh);
                out.energy_add[0] = weight * pv.beta[0] * emission;
                PT_ROLLED for (int k = 1; k < NL; ++k) pl_set<NL>(out.energy_add, k, weight * pl_get<NL>(pv.beta, k) * material_emission(s, m, pl_get<NL>(lam, k), wi));
                out.add_energy = true;
            }
        }
    } else if (rp.light_samples > 0 && PT_MATERIAL_TAG(hit.material) != PT_TAG_LIGHT) {   // (the second test: exactly the vertices shade_wants_item reserved an item for, whatever `lacks` folded above)
        // pt.rs:562-604 -> estimate_direct_illumination_with_loop (pt.rs:333-393)
        uint32_t n_lights = (s.lacks & PT_SCENE_NO_LIGHTS) ? 0u : bu(s, PT_HDR_LIGHT_COUNT);
        float env_p = bf(s, PT_HDR_ENV_PROB);
        if (!(n_lights == 0 && env_p == 0.0f)) {
            F3 hn = normalize(hit.n);  // HitRecord::from(vertex) renormalises (utils.rs:117-134)
            Frame fr2 = frame_from_normal(hn);
            F3 wi2 = to_local(fr2, normalize(sub(pv.prev_p, hit.p)));
            // (the light-sample rays of a scene with one light are tested against it here — below — unless the scene forbids the light bound)
            // (the shortcut below is legal exactly where shadow_light_bound takes its bound from nearest_light_hit: the same flags)
            const bool one_light = n_lights == 1u && !(bu(s, PT_HDR_FLAGS) & (PT_FLAG_NO_SHADOW_BOUND | PT_FLAG_NO_LIGHT_PREPASS | PT_FLAG_NO_CULL | PT_FLAG_NO_ONE_LIGHT));
            EnvCurves ec[NL];  // the environment's spectral weights at this vertex' wavelengths, for all its light samples
            for (int k = 0; k < NL; ++k) ec[k] = (ENV && env_p > 0.0f) ? env_curves(s, lam[k]) : EnvCurves{{0.0f, 0.0f, 0.0f, 0.0f}, false};
            for (uint32_t l = 0; l < rp.light_samples; ++l) {
                ShadowRayT<NL> ray; ray.o = f3(0, 0, 0); ray.d = f3(0, 0, 0);
                for (int k = 0; k < NL; ++k) ray.factor[k] = 0.0f;
                pt_f32x4 q = pt_draw4(rp.seed, pixel, sample, pt_dim_bounce(bounce, rp.light_samples) + 1u + l);
                float x = q.x;
                const bool sample_world = choose_first(&x, env_p) && ENV;
                if (ENV && sample_world) {
                    // estimate_direct_illumination_from_world, pt.rs:224-331
                    float eu, ev, light_pdf;
                    env_sample_uv(s, q.y, q.z, &eu, &ev, &light_pdf);
                    F3 direction = uv_to_direction(eu, ev);
                    F3 local_wo = to_local(fr2, direction);
                    if (local_wo.z > 0.0f) {
                        const EnvPoint ep = env_point_of(s, direction);   // (the sample's emission starts from the same direction: not taken twice)
                        float refl, spdf;
                        material_bsdf_p<GGX>(me0, wi2, local_wo, &refl, &spdf);
                        float weight = rp.only_direct ? 1.0f : light_pdf / (light_pdf + spdf);
                        ray.o = add(hit.p, mul(mul(hn, 0.001f), pt_signum(direction.z)));
                        ray.d = direction;
                        ray.factor[0] = pv.beta[0] * weight * refl * env_emission(s, ep, lambda, ec[0]) * pt_abs(local_wo.z) * (1.0f / light_pdf);
                        for (int k = 1; k < NL; ++k) {   // (hero wavelengths under an environment that light samples pick: no BASELINE configuration; unrolled)
                            float rk, pk; material_bsdf_p<GGX>(material_at<NL>(me, k), wi2, local_wo, &rk, &pk);
                            ray.factor[k] = pv.beta[k] * weight * rk * env_emission(s, ep, lam[k], ec[k]) * pt_abs(local_wo.z) * (1.0f / light_pdf);
                        }
                        out.shadow_count += 1;
                        if (certs && (cv_out | cv_in)) {
                            PT_KEEP_BRANCH_NOFENCE();
                            // An environment ray always LEAVES on the normal's side (local_wo.z > 0) but starts on the side of the WORLD z of its direction (pt.rs:256, a kept
                            // quirk): with direction.z < 0 it starts 1e-3 INSIDE a certified body and must cross its closed surface — any hit blocks an environment ray
                            // (pt.rs:300-330): dead here; with direction.z > 0 it starts outside and cannot hit the body again.
                            if (cv_in && direction.z < 0.0f) for (int k = 0; k < NL; ++k) ray.factor[k] = 0.0f;
                            if (cv_out && local_wo.z > cv_tau && direction.z > 0.0f) out.env_mask |= 0x100u << l;
                        }
                        // a contribution of exactly 0 adds 0 whether or not the ray is occluded: not traced
                        if (ray_is_live<NL>(ray)) out.env_mask |= 1u << l;
                    }
                } else if (n_lights != 0) {
                    // estimate_direct_illumination, pt.rs:146-218
                    float fi = pt_clamp((float)n_lights * x, 0.0f, (float)n_lights - 1.0f);
                    uint32_t light_id = bu(s, bu(s, PT_HDR_LIGHT_OFF) + (uint32_t)fi);
                    float pick_pdf = 1.0f / (float)n_lights;
                    F3 ldir; float light_pdf;
                    light_sample(s, bu(s, PT_HDR_INSTANCE_OFF) + light_id * PT_INST_WORDS, q.y, q.z, hit.p, &ldir, &light_pdf);
                    light_pdf = light_pdf * pick_pdf;
                    if (light_pdf != 0.0f) {
                        F3 bsdf_wo = to_local(fr2, ldir);
                        float refl, bpdf;
                        material_bsdf_p<GGX>(me0, wi2, bsdf_wo, &refl, &bpdf);
                        float weight = rp.only_direct ? 1.0f : light_pdf / (light_pdf + bpdf);
                        ray.o = add(hit.p, mul(mul(hn, 0.001f), pt_signum(bsdf_wo.z)));
                        ray.d = ldir;
                        // pt.rs:196-202: reflectance * throughput * cos_i * cos_o * emission * weight / light_pdf; cos_i and the
                        // emission are only known at the shadow hit and are multiplied in there.
                        ray.factor[0] = refl * pv.beta[0] * pt_abs(bsdf_wo.z) * weight / light_pdf;
                        auto passenger = [&](int k) {
                            float rk, pk; material_bsdf_p<GGX>(material_at<NL>(me, k), wi2, bsdf_wo, &rk, &pk);
                            pl_set<NL>(ray.factor, k, rk * pl_get<NL>(pv.beta, k) * pt_abs(bsdf_wo.z) * weight / light_pdf);
                        };
                        // (the microfacet evaluation is a big body: rolled; the Lambertian one a few instructions: the compiler's choice)
                        if (GGX) { PT_ROLLED for (int k = 1; k < NL; ++k) passenger(k); } else { for (int k = 1; k < NL; ++k) passenger(k); }
                        out.shadow_count += 1;
                        if (certs && (cv_out | cv_in)) {
                            PT_KEEP_BRANCH_NOFENCE();
                            // inward from a certified body: the reference's closest hit is the body's own surface, or something inside it — no light (pt.rs:177-189): the sample adds 0
                            if (cv_in && bsdf_wo.z < 0.0f) for (int k = 0; k < NL; ++k) ray.factor[k] = 0.0f;
                            if (cv_out && bsdf_wo.z > cv_tau) out.env_mask |= 0x100u << l;
                        }
                        // The scene's ONLY light: a ray that misses it meets no light at all — the light-sample kernel's search bound (shadow_light_bound:
                        // nearest_light_hit = +inf) would drop it untraced, with the same test on the same ray.  Found here it makes the ray dead, and an
                        // item whose rays are all dead is never read (Layout::shadow_live_field).  In the Cornell box that is every vertex on the
                        // ceiling: its rays start, after the normal offset, BELOW the lamp that hangs 1e-4 under it — 16 % of C2's items.
                        if (PT_ONE_LIGHT_FORMS && one_light) {
                            // (light_shape_hit: the very call nearest_light_hit makes per light — it kills a SUBSET of what the light-sample kernel
                            // would drop, which runs the light's box test first; PT_AMD_NO_ONE_LIGHT switches this off and the GPU tests compare the two)
                            Hit lh;
                            if (!light_shape_hit(s, bu(s, PT_HDR_INSTANCE_OFF) + light_id * PT_INST_WORDS, ray.o, ray.d, &lh)) for (int k = 0; k < NL; ++k) ray.factor[k] = 0.0f;
                        }
                    }
                }
                sink(l, ray);
            }
            if (certs && (out.env_mask & 0xff00u)) out.env_mask |= hit.instance << 16;   // (the hit's face claims fall off the top: the instance's index is its low 16 bits)   // (the instance the marked rays may skip; certified instances are numbered below 65536)
            out.has_item = true;
        }
    }

    // continue the walk (utils.rs:301-329); passengers are divided by the hero's pdf (sketch utils.rs:493)
    float beta[NL];
    for (int k = 1; k < NL; ++k) beta[k] = 0.0f;
    beta[0] = pv.beta[0] * (f / pdf_forward);
    {
        auto passenger = [&](int k) { float fk, pk; material_bsdf_p<GGX>(material_at<NL>(me, k), wi, wo, &fk, &pk); pl_set<NL>(beta, k, pl_get<NL>(pv.beta, k) * (fk / pdf_forward)); };
        if (GGX) { PT_ROLLED for (int k = 1; k < NL; ++k) passenger(k); } else { for (int k = 1; k < NL; ++k) passenger(k); }
    }
    if (pdf_forward == 0.0f) for (int k = 0; k < NL; ++k) beta[k] = 0.0f;
    if (beta[0] == 0.0f) return out;
    if (r.z > rr) return out;
    if (bounce + 1 >= (rp.only_direct ? 1u : rp.max_bounces)) return out;  // `for bounce in 0..bounce_limit`
    out.survives = true;
    out.next.o = add(hit.p, mul(mul(hit.n, 0.001f), pt_signum(wo.z)));
    out.next.d = normalize(to_world(frame, wo));
    for (int k = 0; k < NL; ++k) out.next.beta[k] = beta[k];
    out.next.lambda = pv.lambda; out.next.slot = pv.slot;
    out.next.prev_pdf = pdf_forward; out.next.prev_n = hit.n; out.next.prev_p = hit.p;
    // The next segment leaves the scene's one certified convex body outward (it starts 1e-3 outside the face it left, hit.n being that face's normal, and moves away): it
    // cannot hit that instance again.  Marked in the sign of the previous-pdf word — every reader takes its magnitude (load_path) — for the parked closest-hit kernel.
    if (certs && cv_path && wo.z > cv_tau) out.next.prev_pdf = -pdf_forward;
    // ... and one that leaves it INWARD from an inward-safe face starts inside the closed body: marked in the slot word's top bit (PT_PATH_INSIDE_MARK; the vertex kernel of
    // the next bounce takes it off again), for mesh_walk's `inside`
    if (certs && cv_in && cv_only && wo.z < 0.0f) out.next.slot |= PT_PATH_INSIDE_MARK;
    return out;
}

// ------------------------------------------------------------------------------------------------ shade, medium-aware
// One vertex of random_walk_medium (utils.rs:708-1103) + the matching iteration of color()'s second pass, which only looks at pairs of
// SURFACE vertices (pt.rs:481-611): a surface vertex behind a medium vertex takes no light samples, an environment vertex behind one
// adds nothing.  Differences from the plain walk that matter: the vertex carries the throughput from BEFORE this segment's
// attenuation (utils.rs:733-749); the direction comes from Material::generate and f, pdf from Material::bsdf; a vertex whose pdf is 0 or
// NaN, or that the roulette stops, is never pushed (:858-869); the walk does not stop when the throughput reaches 0.  The free-flight
// and phase samples are counter-based (PT_TAG_MEDIUM_*; the reference takes them from the thread RNG).  A light-tagged vertex adds
// nothing and takes no light samples (the reference panics there, pt.rs:575-582).  Single wavelength.
// Path state beyond PathVertexT: the tracked mediums (mediums_add / mediums_remove) and whether the previous vertex was a medium vertex.
enum { PS_MEDIUMS = PS_FIELDS, PS_PREV_MEDIUM = PS_FIELDS + 1 };   // (the passenger-throughput fields of the hero layout: the two variants exclude each other)
struct MediumState { uint32_t mediums, prev_medium; uint32_t dropped = 0u; };   // `dropped`: out only — mediums a fifth nesting level lost at this vertex (not part of the path record)
PT_HD bool shade_medium_wants_item(const SceneView& s, const RenderParams& rp, const Hit& hit, const MediumState& ms) {
    return ms.prev_medium == 0u && shade_wants_item(s, rp, hit);
}
// In two steps (round 5, for the kernel: k_shade_medium runs the second for whole waves of surface vertices): stage_medium_flight — the environment vertex, or the free
// flights through the tracked mediums and, when one of them scatters in front of the hit, the medium vertex; it returns true when the vertex is the SURFACE hit, with the
// throughput the segment's attenuation left — and stage_medium_surface, the surface vertex from there.  stage_shade_medium is the two in sequence.
PT_HD void shade_out_clear(ShadeOutT<1>* out) {
    out->survives = false; out->add_energy = false; out->vertex_pushed = false; out->env_hit = false;
    out->shadow_count = 0; out->env_mask = 0; out->has_item = false; out->energy_add[0] = 0.0f;
}
PT_HD bool stage_medium_flight(const SceneView& s, const RenderParams& rp, uint32_t bounce, const PathVertexT<1>& pv, const Hit& hit,
                               uint32_t pixel, const MediumState& ms, MediumState* ms_out, ShadeOutT<1>* out_p, float* beta_out) {
    ShadeOutT<1>& out = *out_p;
    shade_out_clear(&out);
    *ms_out = ms; *beta_out = 0.0f;
    const uint32_t sample = rp.first_sample + pv.slot / rp.chunk_pixels;
    const float lambda = pv.lambda;
    const bool more = bounce + 1 < (rp.only_direct ? 1u : rp.max_bounces);
    if (!hit.valid) {
        out.vertex_pushed = true;   // the environment vertex (utils.rs:1069-1096); looked at only behind a surface vertex
        if (ms.prev_medium == 0u) {
            F3 wo = pv.d;
            float u = 0.0f, v = 0.0f;
            if (bu(s, PT_HDR_ENV_KIND) != PT_ENV_CONSTANT) direction_to_uv(wo, &u, &v);
            float cos_i = pt_abs(dot(pv.prev_n, wo));
            const EnvPoint ep = env_point(s, u, v);
            float nee_psa_pdf = env_pdf_for(s, ep) / pt_abs(cos_i);
            float bsdf_psa_pdf = pv.prev_pdf / pt_abs(cos_i);
            float weight = (bsdf_psa_pdf * bsdf_psa_pdf) / (bsdf_psa_pdf * bsdf_psa_pdf + nee_psa_pdf * nee_psa_pdf);
            out.energy_add[0] = weight * pv.beta[0] * env_emission(s, ep, lambda);
            out.add_energy = true; out.env_hit = true;
        }
        return false;
    }
    // the nearest scattering event of the tracked mediums in front of the hit (utils.rs:766-793), then the segment's attenuation (:794-806)
    // (the four slots as unrolled, guarded steps: every index is a constant, so the mediums' values and the flight samples stay in registers — as a run-time-indexed
    // array they were 80 bytes of stack per lane; the scattering medium's values are kept where they are chosen instead of being looked up again)
    MediumEval me[4];
    uint32_t n_tracked = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t id = (ms.mediums >> (8u * k)) & 0xffu;
        if (n_tracked == k && id != 0u) { me[k] = medium_prepare(s, medium_record(s, id), lambda); n_tracked = k + 1u; }
    }
    float medium_time = hit.t; F3 medium_point = hit.p; uint32_t medium_slot = 4u;
    float hero_weight = 1.0f, hero_tr = 1.0f;
    MediumEval scattering = {0u, 0.0f, 0.0f, 0.0f};
    if (n_tracked != 0u) {
        const pt_f32x4 fd = pt_draw4_tagged(rp.seed, pixel, sample, bounce, PT_TAG_MEDIUM_DISTANCE);
        const float flight[4] = {fd.x, fd.y, fd.z, fd.w};
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            if (k < n_tracked) {
                F3 p; float w;
                medium_sample(me[k], pv.o, pv.d, flight[k], &p, &w);
                const float t = norm(sub(p, pv.o));
                if (t < medium_time) { medium_time = t; medium_point = p; hero_weight = w; hero_tr = medium_tr(me[k], pv.o, p); medium_slot = k; scattering = me[k]; }
            }
        }
    }
    float beta = pv.beta[0] * hero_weight;
    float combined = 1.0f;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) if (k < n_tracked) combined *= medium_tr(me[k], pv.o, medium_point);
    beta *= combined / hero_tr;
    if (medium_slot != 4u) {
        // Vertex::Medium (utils.rs:1031-1066): a new direction from the phase function, the throughput untouched
        const pt_f32x4 ph = pt_draw4_tagged(rp.seed, pixel, sample, bounce, PT_TAG_MEDIUM_PHASE);
        float phase;
        const F3 wo = medium_sample_p(scattering, neg(pv.d), ph.x, ph.y, &phase);
        out.vertex_pushed = true;
        ms_out->prev_medium = 1u;
        out.survives = more;
        out.next.o = medium_point; out.next.d = wo; out.next.beta[0] = beta; out.next.lambda = lambda; out.next.slot = pv.slot;
        out.next.prev_pdf = phase; out.next.prev_n = wo; out.next.prev_p = medium_point;
        return false;
    }
    *beta_out = beta;
    return true;
}
template <typename RaySink>
PT_HD ShadeOutT<1> stage_medium_surface(const SceneView& s, const RenderParams& rp, uint32_t bounce, const PathVertexT<1>& pv, const Hit& hit,
                                        uint32_t pixel, const MediumState& ms, float beta, MediumState* ms_out, RaySink&& sink) {
    ShadeOutT<1> out;
    shade_out_clear(&out);
    *ms_out = ms;
    const uint32_t sample = rp.first_sample + pv.slot / rp.chunk_pixels;
    const float lambda = pv.lambda;
    const bool more = bounce + 1 < (rp.only_direct ? 1u : rp.max_bounces);
    const Frame frame = frame_from_normal(hit.n);
    const F3 wi = normalize(to_local(frame, neg(pv.d)));
    if (PT_MATERIAL_TAG(hit.material) == PT_TAG_CAMERA) return out;
    const uint32_t m = material_record(s, hit.material);
    const pt_f32x4 r = pt_draw4(rp.seed, pixel, sample, pt_dim_bounce(bounce, rp.light_samples));
    const MatEval mev = material_prepare(s, m, lambda, hit.u, hit.v);
    float f0, pdf0, f, pdf; F3 wo;
    material_sample_p(mev, r.x, r.y, wi, &f0, &wo, &pdf0);   // Material::generate: the direction of generate_and_evaluate (materials/mod.rs:76-86)
    material_bsdf_p(mev, wi, wo, &f, &pdf);
    const float cos_i = pt_abs(wo.z);
    if (pdf == 0.0f || pt_isnan(pdf)) return out;            // never pushed (:858-860)
    const float rr = (bounce >= rp.min_bounces) ? pt_min(f / pdf, 1.0f) : 1.0f;
    if (r.z > rr) return out;                                // nor here (:866-869)
    beta *= f * pt_abs(cos_i) * (1.0f / (rr * pdf));
    const float pdf_forward = pdf * (rr / cos_i);
    out.vertex_pushed = true;
    // the second pass at this vertex (pt.rs:562-604): light samples with the vertex's own throughput, unless the pair is not (Surface, Surface)
    if (ms.prev_medium == 0u && PT_MATERIAL_TAG(hit.material) != PT_TAG_LIGHT && rp.light_samples > 0) {
        const uint32_t n_lights = bu(s, PT_HDR_LIGHT_COUNT);
        const float env_p = bf(s, PT_HDR_ENV_PROB);
        if (!(n_lights == 0 && env_p == 0.0f)) {
            const F3 hn = normalize(hit.n);
            const Frame fr2 = frame_from_normal(hn);
            const F3 wi2 = to_local(fr2, normalize(sub(pv.prev_p, hit.p)));
            const EnvCurves ec = env_p > 0.0f ? env_curves(s, lambda) : EnvCurves{{0.0f, 0.0f, 0.0f, 0.0f}, false};
            for (uint32_t l = 0; l < rp.light_samples; ++l) {
                ShadowRayT<1> ray; ray.o = f3(0, 0, 0); ray.d = f3(0, 0, 0); ray.factor[0] = 0.0f;
                const pt_f32x4 q = pt_draw4(rp.seed, pixel, sample, pt_dim_bounce(bounce, rp.light_samples) + 1u + l);
                float x = q.x;
                if (choose_first(&x, env_p)) {   // estimate_direct_illumination_from_world, pt.rs:224-331
                    float eu, ev, light_pdf;
                    env_sample_uv(s, q.y, q.z, &eu, &ev, &light_pdf);
                    const F3 direction = uv_to_direction(eu, ev);
                    const F3 local_wo = to_local(fr2, direction);
                    if (local_wo.z > 0.0f) {
                        const EnvPoint ep = env_point_of(s, direction);
                        float refl, spdf;
                        material_bsdf_p(mev, wi2, local_wo, &refl, &spdf);
                        const float weight = rp.only_direct ? 1.0f : light_pdf / (light_pdf + spdf);
                        ray.o = add(hit.p, mul(mul(hn, 0.001f), pt_signum(direction.z)));
                        ray.d = direction;
                        ray.factor[0] = pv.beta[0] * weight * refl * env_emission(s, ep, lambda, ec) * pt_abs(local_wo.z) * (1.0f / light_pdf);
                        out.shadow_count += 1;
                        if (ray_is_live<1>(ray)) out.env_mask |= 1u << l;
                    }
                } else if (n_lights != 0) {      // estimate_direct_illumination, pt.rs:146-218
                    const float fi = pt_clamp((float)n_lights * x, 0.0f, (float)n_lights - 1.0f);
                    const uint32_t light_id = bu(s, bu(s, PT_HDR_LIGHT_OFF) + (uint32_t)fi);
                    F3 ldir; float light_pdf;
                    light_sample(s, bu(s, PT_HDR_INSTANCE_OFF) + light_id * PT_INST_WORDS, q.y, q.z, hit.p, &ldir, &light_pdf);
                    light_pdf = light_pdf * (1.0f / (float)n_lights);
                    if (light_pdf != 0.0f) {
                        const F3 bsdf_wo = to_local(fr2, ldir);
                        float refl, bpdf;
                        material_bsdf_p(mev, wi2, bsdf_wo, &refl, &bpdf);
                        const float weight = rp.only_direct ? 1.0f : light_pdf / (light_pdf + bpdf);
                        ray.o = add(hit.p, mul(mul(hn, 0.001f), pt_signum(bsdf_wo.z)));
                        ray.d = ldir;
                        ray.factor[0] = refl * pv.beta[0] * pt_abs(bsdf_wo.z) * weight / light_pdf;
                        out.shadow_count += 1;
                    }
                }
                sink(l, ray);
            }
            out.has_item = true;
        }
    }
    // medium transitions (utils.rs:925-991): only on transmission through a boundary whose two sides differ
    const uint32_t mm = bu(s, m + PT_MAT_MEDIUMS), outer = mm & 0xffu, inner = (mm >> 8) & 0xffu;
    uint32_t list = ms.mediums;
    if (!(wi.z * wo.z > 0.0f) && inner != outer) {
        if (wo.z < 0.0f) { if (outer != 0u) list = mediums_remove(list, outer); if (inner != 0u) list = mediums_add(list, inner, &ms_out->dropped); }
        else { if (inner != 0u) list = mediums_remove(list, inner); if (outer != 0u) list = mediums_add(list, outer, &ms_out->dropped); }
    }
    ms_out->mediums = list; ms_out->prev_medium = 0u;
    out.survives = more;
    out.next.o = add(hit.p, mul(mul(hit.n, 0.001f), wo.z > 0.0f ? 1.0f : -1.0f));
    out.next.d = normalize(to_world(frame, wo));
    out.next.beta[0] = beta; out.next.lambda = lambda; out.next.slot = pv.slot;
    out.next.prev_pdf = pdf_forward; out.next.prev_n = hit.n; out.next.prev_p = hit.p;
    return out;
}
template <typename RaySink>
PT_HD ShadeOutT<1> stage_shade_medium(const SceneView& s, const RenderParams& rp, uint32_t bounce, const PathVertexT<1>& pv, const Hit& hit,
                                      uint32_t pixel, const MediumState& ms, MediumState* ms_out, RaySink&& sink) {
    ShadeOutT<1> out; float beta;
    if (!stage_medium_flight(s, rp, bounce, pv, hit, pixel, ms, ms_out, &out, &beta)) return out;
    return stage_medium_surface(s, rp, bounce, pv, hit, pixel, ms, beta, ms_out, sink);
}

// queue I/O of one light-sample ray
// The sample number l is a run-time value (a loop counter, or per lane in the kernels that list their rays): one address is formed for the ray's first
// field and its other fields are constant offsets from it.  Formed field by field — qindex(q, f0 + field, item) — every field is a 64-bit address of its
// own ((f0 + field) is a 32-bit sum the compiler may not split), seven of them alive at once in k_shade's light-sample loop.
template <int NL> PT_HD uint32_t* shadow_ray_base(const Queue& q, uint32_t item, uint32_t l) { return q.base + qindex(q, Layout<NL>::sh_head + l * Layout<NL>::sr_fields, item); }
PT_HD float rayf(const Queue& q, const uint32_t* rb, uint32_t field) { return pt_u2f(PT_QLOAD(&rb[field * qstride(q)])); }
PT_HD void rayfs(const Queue& q, uint32_t* rb, uint32_t field, float v) { PT_QSTORE(pt_f2u(v), &rb[field * qstride(q)]); }
template <int NL>
PT_HD void store_shadow_ray(const Queue& q, uint32_t item, uint32_t l, const ShadowRayT<NL>& ray) {
    uint32_t* const rb = shadow_ray_base<NL>(q, item, l);
    for (int k = 0; k < NL; ++k) rayfs(q, rb, SR_FACTOR + k, ray.factor[k]);
    if (ray_is_live<NL>(ray)) {
        rayfs(q, rb, SR_OX, ray.o.x); rayfs(q, rb, SR_OY, ray.o.y); rayfs(q, rb, SR_OZ, ray.o.z);
        rayfs(q, rb, SR_DX, ray.d.x); rayfs(q, rb, SR_DY, ray.d.y); rayfs(q, rb, SR_DZ, ray.d.z);
    }
}
// (EAGER: origin and direction are read along with the factors instead of after them — a dead ray's are stale words, unused)
template <int NL, bool EAGER = false, bool NT_RAY = false>
PT_HD bool load_shadow_ray(const Queue& q, uint32_t item, uint32_t l, ShadowRayT<NL>* ray) {
    const uint32_t* const rb = shadow_ray_base<NL>(q, item, l);
    for (int k = 0; k < NL; ++k) ray->factor[k] = rayf(q, rb, SR_FACTOR + k);
    if (!EAGER && !ray_is_live<NL>(*ray)) return false;
    auto rd = [&](uint32_t field) { return NT_RAY ? pt_u2f(PT_QLOAD_NT(&rb[field * qstride(q)])) : rayf(q, rb, field); };
    ray->o = f3(rd(SR_OX), rd(SR_OY), rd(SR_OZ));
    ray->d = f3(rd(SR_DX), rd(SR_DY), rd(SR_DZ));
    return ray_is_live<NL>(*ray);
}
template <int NL>
PT_HD void clear_shadow_item(const Queue& q, uint32_t item, uint32_t light_samples) {
    for (uint32_t l = 0; l < light_samples; ++l) {
        uint32_t* const rb = shadow_ray_base<NL>(q, item, l);
        for (int k = 0; k < NL; ++k) rayfs(q, rb, SR_FACTOR + k, 0.0f);
    }
}

// ------------------------------------------------------------------------------------------------ shadow
// The light-sample ray of pt.rs:171-217: nearest hit must be *a* light; emission is evaluated at that hit.
// Only a light that is the closest hit contributes, so the walk is bounded by the nearest light hit (nothing beyond it
// can be the closest hit) and may stop at the first accepted non-light hit in front of it.  If no light is hit at all
// the ray contributes nothing and is not walked.  The result is the reference's in every case.
// What a traced light-sample ray adds, given its closest hit: a light ray contributes only if that hit is a light, with
// the emission evaluated there (pt.rs:177-217); an environment ray only if nothing was hit (pt.rs:300-330).
// `lambda_of(k)`: the item's k-th wavelength — a register of the caller, or (hero wavelengths, where four of them held across two searches
// cost the six-wave kernel registers it does not have) a read of the item's record at the moment it is needed.
template <int NL, typename LambdaOf, typename FactorOf>
PT_HD void shadow_ray_contribution(const SceneView& s, LambdaOf&& lambda_of, FactorOf&& factor_of, F3 ray_d, bool env, bool hit, const Hit& sh, float (&contribution)[NL]) {
    for (int k = 0; k < NL; ++k) contribution[k] = 0.0f;
    if (env) { if (!hit) for (int k = 0; k < NL; ++k) contribution[k] = factor_of(k); return; }
    if (!hit || PT_MATERIAL_TAG(sh.material) != PT_TAG_LIGHT) return;
    Frame lf = frame_from_normal(sh.n);
    F3 lwi = to_local(lf, neg(ray_d));
    uint32_t lm = material_record(s, sh.material);
    PT_ROLLED for (int k = 0; k < NL; ++k) pl_set<NL>(contribution, k, factor_of(k) * pt_abs(lwi.z) * material_emission(s, lm, lambda_of(k), lwi));
}
template <int NL, typename LambdaOf>
PT_HD void shadow_ray_contribution(const SceneView& s, LambdaOf&& lambda_of, const ShadowRayT<NL>& ray, bool env, bool hit, const Hit& sh, float (&contribution)[NL]) {
    shadow_ray_contribution<NL>(s, lambda_of, [&](int k) { return pl_get<NL>(ray.factor, k); }, ray.d, env, hit, sh, contribution);
}
// The bound of a light ray's search: the nearest light hit (+inf: no light on the ray, nothing to trace), or "unbounded"
// when the scene forbids the shortcut.  Returns false when the ray cannot contribute.
PT_HD bool shadow_light_bound(const SceneView& s, F3 o, F3 d, float* bound, int* stop, uint32_t* light = nullptr) {
    if (light != nullptr) *light = 0xffffffffu;
    if (bu(s, PT_HDR_FLAGS) & (PT_FLAG_NO_SHADOW_BOUND | PT_FLAG_NO_LIGHT_PREPASS | PT_FLAG_NO_CULL)) { *bound = PT_INF; *stop = PT_STOP_NONE; return true; }
    float t_light = nearest_light_hit(s, o, d, light);
    if (light != nullptr && (bu(s, PT_HDR_FLAGS) & PT_FLAG_NO_KNOWN_LIGHT)) *light = 0xffffffffu;
    *bound = t_light; *stop = PT_STOP_NONLIGHT;
    return t_light < PT_INF;
}
PT_HD int shadow_light_stop(const SceneView& s) { return (bu(s, PT_HDR_FLAGS) & (PT_FLAG_NO_SHADOW_BOUND | PT_FLAG_NO_LIGHT_PREPASS | PT_FLAG_NO_CULL)) ? PT_STOP_NONE : PT_STOP_NONLIGHT; }   // (the stop rule shadow_light_bound sets)
PT_HD int shadow_env_stop(const SceneView& s) { return (bu(s, PT_HDR_FLAGS) & PT_FLAG_NO_CULL) ? PT_STOP_NONE : PT_STOP_ANY; }
// One light-sample ray: a light ray (pt.rs:171-217) or an environment ray (pt.rs:252-330, contributes only if nothing is hit: any hit
// blocks it, so its search ends at the first one; PT_AMD_NO_CULL keeps the full search).  One search call site for both kinds — a
// wave whose lanes hold both kinds traces them together.  ENV = false: the caller knows the scene produces no environment rays
// (env_sampling_probability = 0) and that half is compiled out.
template <int NL, int TRAV = PT_TRAV_ANY, bool ENV = true, typename LambdaOf, typename FactorOf>
PT_HD void stage_shadow_ray(const SceneView& s, LambdaOf&& lambda_of, FactorOf&& factor_of, F3 o, F3 d, bool env, float (&contribution)[NL], uint32_t skip_inst = 0xffffffffu) {
    for (int k = 0; k < NL; ++k) contribution[k] = 0.0f;
    float bound = PT_INF; int stop = PT_STOP_NONE;
    uint32_t light = 0xffffffffu;   // the light whose hit bounds the search: its test has been run, phase 3 takes the distance (sweep_run)
    if (ENV && env) stop = shadow_env_stop(s);
    else if (!shadow_light_bound(s, o, d, &bound, &stop, &light)) { PT_STAT(rays_without_light); return; }
    Hit sh;
    bool hit = world_hit<TRAV, true>(s, o, d, &sh, bound, stop, light, bound, skip_inst);
    shadow_ray_contribution<NL>(s, lambda_of, factor_of, d, ENV && env, hit, sh, contribution);
}
// (PT_SHADOW_EAGER true: the pure sweep form reads a ray's origin and direction along with its factor — measured: k_shadow 3525 -> 3600 us on C2, not used)
#ifndef PT_SHADOW_EAGER
#define PT_SHADOW_EAGER false
#endif
// One light-sample item: L rays, summed in order, divided by L (pt.rs:349-392, 596)
template <int NL, int TRAV = PT_TRAV_ANY, bool ENV = true>
PT_HD void stage_shadow_item(const SceneView& s, uint32_t light_samples, const Queue& shadow, uint32_t item, float* energy, uint32_t energy_stride) {
    // (the pure sweep kernels' tables hold no walked mesh: a mark "cannot hit its instance again" — bit 8 + l, PT_INST_CONVEX_OUT — would save them a triangle leaf or two; not read there)
    const bool marks = TRAV != PT_TRAV_SWEEP && scene_has_certificates(s);
    uint32_t slot = qu(shadow, Layout<NL>::sh_slot, item), flags = (ENV || marks) ? qu(shadow, Layout<NL>::sh_flags, item) : 0u;
    float lambda0 = 0.0f, lc[NL];
    if (NL == 1) lambda0 = qf(shadow, Layout<NL>::sh_lambda, item);
    for (int k = 0; k < NL; ++k) lc[k] = 0.0f;
    auto lambda_of = [&](int k) { return NL == 1 ? lambda0 : qf(shadow, Layout<NL>::sh_lambda + (uint32_t)k, item); };
    for (uint32_t l = 0; l < light_samples; ++l) {
        ShadowRayT<NL> ray;
        if (!load_shadow_ray<NL, PT_SHADOW_EAGER && TRAV == PT_TRAV_SWEEP>(shadow, item, l, &ray)) { PT_STAT(rays_dead); continue; }
        PT_STAT(rays_live);
        float c[NL];
        // (hero wavelengths: the factors too are read again when the ray contributes, not held across its search)
        const uint32_t ff = Layout<NL>::sh_head + l * Layout<NL>::sr_fields + SR_FACTOR;
        auto factor_of = [&](int k) { return NL == 1 ? ray.factor[0] : qf(shadow, ff + (uint32_t)k, item); };
        stage_shadow_ray<NL, TRAV, ENV>(s, lambda_of, factor_of, ray.o, ray.d, ENV && ((flags >> l) & 1u) != 0u, c, (marks && ((flags >> (8u + l)) & 1u) != 0u) ? flags >> 16 : 0xffffffffu);
        for (int k = 0; k < NL; ++k) lc[k] += c[k];
    }
    for (int k = 0; k < NL; ++k) energy[(size_t)k * energy_stride + slot] += lc[k] / (float)light_samples;
}

// ------------------------------------------------------------------------------------------------ accumulate
// XYZColor::from(SingleWavelength) (math crate; pt.rs:614) and the film sums of tiled.rs:366-398.  With hero wavelengths
// the sample's colour is the mean of the four wavelengths' XYZ.
template <int NL>
PT_HD void stage_accumulate_pixel(const RenderParams& rp, const float* energy, uint32_t p, uint32_t pixel, float* film_px) {
    float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
    float f0 = film_px[0], f1 = film_px[1], f2 = film_px[2];
    for (uint32_t s_local = 0; s_local < rp.pass_samples; ++s_local) {
        uint32_t sample = rp.first_sample + s_local;
        size_t slot = (size_t)s_local * rp.chunk_pixels + p;
        // (round 5) the sample's wavelength sample u as k_generate left it in the plane behind the energies (PT_STORED_WAVELENGTH) — not the film block's Philox draw again,
        // which was 40 % of this loop; hero_lambdas(u)[0] is stage_generate's own expression for the wavelength
        float lam[NL];
        if (PT_STORED_WAVELENGTH) hero_lambdas<NL>(rp, energy[(size_t)NL * rp.energy_stride + slot], lam);
        else hero_lambdas<NL>(rp, pt_draw4(rp.seed, pixel, sample, PT_DIM_FILM).z, lam);
        if (NL == 1) {
            float e = energy[slot];
            float xb, yb, zb;
            xyz_bar(lam[0] * 10.0f, &xb, &yb, &zb);
            t0 += e * xb; t1 += e * yb; t2 += e * zb;
        } else {
            float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
            for (int k = 0; k < NL; ++k) {
                float e = energy[(size_t)k * rp.energy_stride + slot];
                float xb, yb, zb;
                xyz_bar(lam[k] * 10.0f, &xb, &yb, &zb);
                c0 += e * xb; c1 += e * yb; c2 += e * zb;
            }
            t0 += c0 / 4.0f; t1 += c1 / 4.0f; t2 += c2 / 4.0f;
        }
        if ((sample + 1) % rp.phase == 0 || sample + 1 == rp.spp || sample + 1 == rp.range_end) {  // phases of 10, tiled.rs:347-361 (or of everything, naive.rs:82-103)
            f0 += t0; f1 += t1; f2 += t2;
            t0 = t1 = t2 = 0.0f;
        }
    }
    if (rp.normalize && rp.first_sample + rp.pass_samples == rp.range_end) {
        float n = (float)rp.spp;
        f0 /= n; f1 /= n; f2 /= n;
    }
    film_px[0] = f0; film_px[1] = f1; film_px[2] = f2; film_px[3] = 0.0f;
}

}  // namespace ptd
#endif
