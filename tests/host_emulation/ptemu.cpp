// ptemu.cpp — TEST HARNESS: runs the HIP engine's per-lane stage functions (csrc/pt_stages.h, pt_device.h)
// and its host-side scene flattening / pass planning on the CPU, one "lane" at a time, with sequential queue
// compaction in place of the wave64 ballot.  It exists so that the wavefront logic can be compared with the
// oracle bit for bit in the GPU-less build container (tests/test_emulation.py); it is not part of the product
// (nothing in rust-pathtracer_amd/ builds, links or loads it) and exports the boundary with the prefix ptemu_.
#include <cstring>
#include <string>
#include <vector>

static unsigned long long g_inside_stops = 0;   // sweeps ended by mesh_walk's `inside` rule since the library was loaded (ptemu_debug_scene_info 18)
#define PT_STAT_INSIDE_STOP() (++g_inside_stops)

#include "../../rust-pathtracer_amd/csrc/pt_plan.h"
#include "../../rust-pathtracer_amd/csrc/pt_scene_host.h"
#include "../../rust-pathtracer_amd/csrc/pt_stages.h"

using namespace ptd;

static thread_local std::string g_error;
struct pt_scene { pth::HostScene host; };

extern "C" {

const char* ptemu_last_error(void) { return g_error.c_str(); }

pt_status ptemu_scene_create(const pt_scene_desc* d, pt_scene** out) {
    pt_scene* sc = new pt_scene();
    if (!pth::build_host_scene(*d, &sc->host, &g_error)) { delete sc; return PT_ERR_INVALID_ARGUMENT; }
    // the engine's diagnostic switches (pt_engine.hip: PT_AMD_EXACT_SLAB / NO_CULL / NO_SWEEP), as one flag word
    const char* f = getenv("PTEMU_FLAGS");
    if (f) sc->host.blob[PT_HDR_FLAGS] |= (uint32_t)strtoul(f, nullptr, 0);
    if (getenv("PTEMU_NO_CONVEX")) sc->host.blob[PT_HDR_FLAGS] &= ~PT_FLAG_CONVEX;   // (pt_engine.hip: PT_TUNE_NO_CONVEX)
    if (getenv("PTEMU_NO_MESH_SHORTCUTS")) {   // (the mesh records' inner ball and slab table: mesh_surely_blocks / mesh_surely_missed claim nothing without them)
        std::vector<uint32_t>& bl = sc->host.blob;
        for (uint32_t i = 0; i < bl[PT_HDR_INSTANCE_COUNT]; ++i) {
            const uint32_t inst = bl[PT_HDR_INSTANCE_OFF] + i * PT_INST_WORDS;
            if (bl[inst + PT_INST_KIND] != (uint32_t)PT_SHAPE_MESH) continue;
            bl[bl[inst + PT_INST_MESH] + PT_MESH_INNER_R] = 0u; bl[bl[inst + PT_INST_MESH] + PT_MESH_DOP_OFF] = 0u;
        }
    }   // (pt_engine.hip: PT_TUNE_NO_MESH_SHORTCUTS)
    *out = sc;
    return PT_OK;
}
void ptemu_scene_destroy(pt_scene* sc) { delete sc; }
uint32_t ptemu_debug_scene_info(pt_scene* sc, int what) {
    const std::vector<uint32_t>& w = sc->host.blob;
    if (what == 0) return (uint32_t)w.size() * 4u;          // bytes of the blob
    if (what == 7) return w[PT_HDR_CORE_WORDS] * 4u;        // bytes of its core section
    if (what >= 1000000) return (size_t)(what - 1000000) < w.size() ? w[(size_t)(what - 1000000)] : 0u;   // a word of the blob
    if (what >= 8 && what <= 14) {   // the inner ball of the first mesh instance's mesh (pt_blob.h PT_MESH_INNER_*): 8..10 centre, 11 radius, 12 reach — float bits —, 13 / 14 the further balls' list (blob word offset, count); 0 = no mesh
        for (uint32_t i = 0; i < w[PT_HDR_INSTANCE_COUNT]; ++i) {
            const uint32_t inst = w[PT_HDR_INSTANCE_OFF] + i * PT_INST_WORDS;
            if (w[inst + PT_INST_KIND] == (uint32_t)PT_SHAPE_MESH) return w[w[inst + PT_INST_MESH] + (uint32_t)what];
        }
        return 0;
    }
    if (what >= 15 && what <= 17) {   // the first mesh instance's record flags (15: PT_INST_CONVEX_* among them) and how many of its mesh's triangles carry PT_TRI_IN_SAFE (16) / PT_TRI_IN_SAFE_INNER (17)
        for (uint32_t i = 0; i < w[PT_HDR_INSTANCE_COUNT]; ++i) {
            const uint32_t inst = w[PT_HDR_INSTANCE_OFF] + i * PT_INST_WORDS;
            if (w[inst + PT_INST_KIND] != (uint32_t)PT_SHAPE_MESH) continue;
            if (what == 15) return w[inst + PT_INST_FLAGS];
            const uint32_t mesh = w[inst + PT_INST_MESH], tri = w[PT_HDR_CORE_WORDS] + w[mesh + PT_MESH_TRI_OFF];
            uint32_t n = 0;
            for (uint32_t f = 0; f < w[mesh + PT_MESH_FACE_COUNT]; ++f) n += (w[tri + f * PT_TRI_WORDS + PT_TRI_FLAGS] & (what == 16 ? PT_TRI_IN_SAFE : PT_TRI_IN_SAFE_INNER)) ? 1u : 0u;
            return n;
        }
        return 0;
    }
    if (what == 18) return (uint32_t)g_inside_stops;
    if (what == 4) return w[PT_HDR_SWEEP_OFF] != 0 && !(w[PT_HDR_FLAGS] & PT_FLAG_NO_SWEEP) ? 1u : 0u;
    if ((what == 5 || what == 6) && w[PT_HDR_SWEEP_OFF] != 0) {  // 5: mask bits in use, 6: bits whose box test is a copy
        uint32_t bits = 0, copies = 0;
        for (uint32_t j = 0; j < w[PT_HDR_SWEEP_COUNT]; ++j) {
            const uint32_t* e = &w[w[PT_HDR_SWEEP_OFF] + j * PT_SWEEP_INST_WORDS];
            bits += 1 + e[11];
            copies += e[11] - (e[1] >> 24);   // triangle leaves without a box test of their own
        }
        return what == 5 ? bits : copies;
    }
    return 0;
}

// the colour-matching fit as the kernels evaluate it (contract = 0) and as the numeric contract defines it (1), for n wavelengths in angstrom: 3 floats each
void ptemu_xyz_bar(size_t n, const float* angstrom, int contract, float* out) {
    for (size_t i = 0; i < n; ++i) {
        if (contract) xyz_bar_contract(angstrom[i], &out[3 * i], &out[3 * i + 1], &out[3 * i + 2]);
        else xyz_bar(angstrom[i], &out[3 * i], &out[3 * i + 1], &out[3 * i + 2]);
    }
}

}  // extern "C"
template <int NL>
static pt_status render_t(pt_scene* sc, const pt_render_desc& rd, float* film, pt_profile* profile) {
    SceneView s{sc->host.blob.data(), sc->host.tex.data(), sc->host.blob.data() + sc->host.blob[PT_HDR_CORE_WORDS]};
    std::vector<uint32_t> pixels = pth::shard_pixels(rd.width, rd.height, rd.tile_width, rd.tile_height, rd.shard_index, rd.shard_count);
    std::memset(film, 0, sizeof(float) * 4 * (size_t)rd.width * rd.height);
    uint32_t capacity = 1u << 16;  // small on purpose: exercises pixel chunking and phase-aligned sample passes
    const char* cap_env = getenv("PTEMU_BATCH");
    if (cap_env) capacity = (uint32_t)strtoul(cap_env, nullptr, 10);
    RenderParams rp;
    std::memset(&rp, 0, sizeof(rp));
    rp.seed = rd.seed; rp.width = rd.width; rp.height = rd.height;
    rp.min_bounces = rd.min_bounces; rp.max_bounces = rd.max_bounces; rp.light_samples = rd.light_samples; rp.only_direct = rd.only_direct;
    rp.wavelength_lo = rd.wavelength_lo; rp.wavelength_span = rd.wavelength_hi - rd.wavelength_lo;
    rp.spp = rd.spp; rp.range_end = rd.first_sample + rd.sample_count;
    rp.normalize = (rd.first_sample == 0 && rd.sample_count == rd.spp) ? 1u : 0u;
    rp.phase = rd.phase_samples;
    rp.energy_stride = capacity;
    rp.camera = pth::camera_params(sc->host.cameras[rd.camera_index], (float)rd.width / (float)rd.height);
    rp.camera_record = 1u;   // (the camera vertex' lean record, pt_stages.h: the engine takes it in the lean and full vertex forms; here wherever the plain walk runs)
    typedef Layout<NL> LY;
    const size_t cap64 = ((size_t)capacity + 63u) & ~(size_t)63u;   // queues are tiled by 64 items (pt_stages.h)
    std::vector<uint32_t> pa((size_t)(LY::path_fields + 2) * cap64), pb((size_t)(LY::path_fields + 2) * cap64), ph((size_t)HS_FIELDS * cap64),
        psh((size_t)LY::shadow_fields(PT_MAX_LIGHT_SAMPLES) * cap64);
    std::vector<float> energy((size_t)(NL + 1) * capacity);   // (+ the plane of wavelength samples, PT_STORED_WAVELENGTH)
    Queue qa{pa.data(), capacity, LY::path_fields + 2}, qb{pb.data(), capacity, LY::path_fields + 2}, qh{ph.data(), capacity, HS_FIELDS}, qs{psh.data(), capacity, LY::shadow_fields(PT_MAX_LIGHT_SAMPLES)};
    uint64_t bounce_rays = 0, shadow_rays = 0, env_hits = 0, camera_rays = 0, medium_drops = 0;
    uint32_t bounce_limit = rd.only_direct ? 1u : rd.max_bounces;
    for (const pth::Pass& pass : pth::plan_passes((uint32_t)pixels.size(), rd.first_sample, rd.sample_count, capacity, rd.phase_samples)) {
        rp.chunk_pixels = pass.pixel_count; rp.first_sample = pass.first_sample; rp.pass_samples = pass.sample_count;
        const uint32_t* px = pixels.data() + pass.pixel_begin;
        uint32_t n = pass.pixel_count * pass.sample_count;
        camera_rays += n;
        for (uint32_t i = 0; i < n; ++i) { float u = 0.0f; const PathVertexT<NL> pg = stage_generate<NL>(rp, i, px[i % rp.chunk_pixels], &u); if (PT_CAMERA_RECORD && rp.camera_record) store_path_camera<NL>(qa, i, pg); else store_path<NL>(qa, i, pg); for (int k = 0; k < NL; ++k) energy[(size_t)k * capacity + i] = 0.0f; energy[(size_t)NL * capacity + i] = u; }
        uint32_t live = n;
        bool has_ggx = false;
        for (uint32_t i = 0; i < bu(s, PT_HDR_MATERIAL_COUNT); ++i) {
            const uint32_t kind = bu(s, bu(s, PT_HDR_MATERIAL_OFF) + i * PT_MAT_WORDS + PT_MAT_KIND);
            has_ggx = has_ggx || kind == PT_MATERIAL_GGX || kind == PT_MATERIAL_PASSTHROUGH;
        }
        const int forced = getenv("PTEMU_SHADE_FORM") ? atoi(getenv("PTEMU_SHADE_FORM")) : 0;
        const bool no_inside = getenv("PTEMU_NO_INSIDE") != nullptr;   // (test switch: PT_PATH_INSIDE_MARK is made and ignored)
        const int shade_form = (bf(s, PT_HDR_ENV_PROB) != 0.0f || forced == 2) ? 2 : (has_ggx || forced == 1) ? 1 : 0;
        for (uint32_t bounce = 0; bounce < bounce_limit; ++bounce) {
            Queue qin = (bounce & 1) ? qb : qa, qout = (bounce & 1) ? qa : qb;
            for (uint32_t i = 0; i < live; ++i) {
                Hit h;
                // (k_extend_parked's path_marks: a marked segment — the sign of its previous-pdf word, from bounce 1 on — cannot hit the scene's one certified convex body)
                const bool marked = bounce > 0 && !rd.medium_aware && (bu(s, PT_HDR_FLAGS) & PT_FLAG_CONVEX) && bu(s, PT_HDR_CONVEX_INST) != 0u && qf(qin, PS_PREV_PDF, i) < 0.0f;
                const bool path_certs = bounce > 0 && !rd.medium_aware && (bu(s, PT_HDR_FLAGS) & PT_FLAG_CONVEX) && bu(s, PT_HDR_CONVEX_INST) != 0u;
                const bool inside = path_certs && !no_inside && (qu(qin, PS_SLOT, i) & PT_PATH_INSIDE_MARK) != 0u;   // (PT_PATH_INSIDE_MARK: mesh_walk's `inside`)
                world_hit(s, f3(qf(qin, PS_OX, i), qf(qin, PS_OY, i), qf(qin, PS_OZ, i)), f3(qf(qin, PS_DX, i), qf(qin, PS_DY, i), qf(qin, PS_DZ, i)), &h, PT_INF, PT_STOP_NONE, 0xffffffffu, 0.0f,
                          marked ? bu(s, PT_HDR_CONVEX_INST) - 1u : 0xffffffffu, inside ? bu(s, PT_HDR_CONVEX_INST) - 1u : 0xffffffffu);
                store_hit(qh, i, h);
            }
            uint32_t next = 0, items = 0;
            for (uint32_t i = 0; i < live; ++i) {
                PathVertexT<NL> pv = load_path<NL>(qin, i, rp.camera_record != 0u && bounce == 0u);
                if (bounce > 0) pv.slot &= ~PT_PATH_INSIDE_MARK;   // (the previous vertex' mark for the closest-hit step)
                Hit hit = load_hit(qh, i);
                bool wants = shade_wants_item(s, rp, hit);   // (the medium-aware walk overrides this below)
                uint32_t ipos = items;
                auto sink = [&](uint32_t l, const ShadowRayT<NL>& ray) { store_shadow_ray<NL>(qs, ipos, l, ray); };
                // the kernel form the engine launches (PT_SHADE_LEAN / NO_ENV / FULL): without the environment-sampling branch when
                // env_sampling_probability is 0, without the GGX code when the scene has no GGX material
                const uint32_t px_i = px[pv.slot % rp.chunk_pixels];
                ShadeOutT<NL> out;
                MediumState ms_next{0u, 0u};
                bool medium_walk = false;
                if constexpr (NL == 1) {
                    if (rd.medium_aware) {   // the medium-aware walk (k_shade_medium): its two extra path fields ride behind the plain record
                        medium_walk = true;
                        MediumState ms{0u, 0u};
                        if (bounce != 0) { ms.mediums = qu(qin, PS_MEDIUMS, i); ms.prev_medium = qu(qin, PS_PREV_MEDIUM, i); }
                        wants = shade_medium_wants_item(s, rp, hit, ms);
                        out = stage_shade_medium(s, rp, bounce, pv, hit, px_i, ms, &ms_next, sink);
                    }
                }
                if (!medium_walk)
                    out = shade_form == 2 ? stage_shade<NL, true, true>(s, rp, bounce, pv, hit, px_i, sink)
                        : shade_form == 1 ? stage_shade<NL, false, true>(s, rp, bounce, pv, hit, px_i, sink)
                                          : stage_shade<NL, false, false>(s, rp, bounce, pv, hit, px_i, sink);
                if (wants) {
                    items++;
                    float lam[NL]; lam[0] = pv.lambda;
                    if (NL > 1) hero_lambdas<NL>(rp, pv.lambda, lam);
                    qsu(qs, LY::sh_slot, ipos, pv.slot); qsu(qs, LY::sh_flags, ipos, out.env_mask);
                    for (int k = 0; k < NL; ++k) qsf(qs, LY::sh_lambda + k, ipos, lam[k]);
                    if (!out.has_item) clear_shadow_item<NL>(qs, ipos, rp.light_samples);
                }
                if (out.add_energy) for (int k = 0; k < NL; ++k) energy[(size_t)k * capacity + pv.slot] += out.energy_add[k];
                if (out.survives) {
                    if (medium_walk) { qsu(qout, PS_MEDIUMS, next, ms_next.mediums); qsu(qout, PS_PREV_MEDIUM, next, ms_next.prev_medium); }
                    store_path<NL>(qout, next++, out.next);
                }
                bounce_rays += out.vertex_pushed; env_hits += out.env_hit; shadow_rays += out.shadow_count; medium_drops += ms_next.dropped;
            }
            for (uint32_t i = 0; i < items; ++i) {
                if (shade_form == 2 || rd.medium_aware) stage_shadow_item<NL, PT_TRAV_ANY, true>(s, rp.light_samples, qs, i, energy.data(), capacity);
                else stage_shadow_item<NL, PT_TRAV_ANY, false>(s, rp.light_samples, qs, i, energy.data(), capacity);
            }
            live = next;
        }
        for (uint32_t p = 0; p < rp.chunk_pixels; ++p) stage_accumulate_pixel<NL>(rp, energy.data(), p, px[p], film + 4 * (size_t)px[p]);
    }
    if (profile) {
        std::memset(profile, 0, sizeof(*profile));
        profile->camera_rays = camera_rays; profile->bounce_rays = bounce_rays + camera_rays; profile->shadow_rays = shadow_rays; profile->env_hits = env_hits;
        profile->stage_items[5] = medium_drops;
    }
    return PT_OK;
}

extern "C" {
pt_status ptemu_render(pt_scene* sc, const pt_render_desc* rdp, float* film, pt_profile* profile) {
    pt_render_desc rd;
    if (!pth::normalize_render_desc(*rdp, (uint32_t)sc->host.cameras.size(), &rd, &g_error)) return PT_ERR_INVALID_ARGUMENT;
    return rd.hero_wavelengths == 4 ? render_t<4>(sc, rd, film, profile) : render_t<1>(sc, rd, film, profile);
}

pt_status ptemu_intersect(pt_scene* sc, size_t n, const float* o, const float* d, pt_hit* hits) {
    SceneView s{sc->host.blob.data(), sc->host.tex.data(), sc->host.blob.data() + sc->host.blob[PT_HDR_CORE_WORDS]};
    for (size_t i = 0; i < n; ++i) {
        Hit h; pt_hit r; std::memset(&r, 0, sizeof(r));
        if (world_hit(s, f3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), f3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), &h)) {
            r.valid = 1; r.t = h.t; r.point[0] = h.p.x; r.point[1] = h.p.y; r.point[2] = h.p.z; r.normal[0] = h.n.x; r.normal[1] = h.n.y; r.normal[2] = h.n.z;
            r.uv[0] = h.u; r.uv[1] = h.v; r.material = h.material; r.instance = hit_instance_index(s, h.instance);
        }
        hits[i] = r;
    }
    return PT_OK;
}
pt_status ptemu_camera_samples(pt_scene* sc, const pt_render_desc* rd, size_t n, const uint32_t* pixel, const uint32_t* sample, float* o, float* d, float* lambda) {
    RenderParams rp;
    std::memset(&rp, 0, sizeof(rp));
    rp.seed = rd->seed; rp.width = rd->width; rp.height = rd->height;
    rp.wavelength_lo = rd->wavelength_lo; rp.wavelength_span = rd->wavelength_hi - rd->wavelength_lo;
    rp.camera = pth::camera_params(sc->host.cameras[rd->camera_index], (float)rd->width / (float)rd->height);
    rp.chunk_pixels = 1;
    for (size_t i = 0; i < n; ++i) {
        const PathVertexT<1> p = stage_generate<1>(rp, sample[i], pixel[i]);
        o[3 * i] = p.o.x; o[3 * i + 1] = p.o.y; o[3 * i + 2] = p.o.z; d[3 * i] = p.d.x; d[3 * i + 1] = p.d.y; d[3 * i + 2] = p.d.z; lambda[i] = p.lambda;
    }
    return PT_OK;
}
static uint32_t mat_rec(pt_scene* sc, uint32_t m) { return sc->host.blob[PT_HDR_MATERIAL_OFF] + m * PT_MAT_WORDS; }
pt_status ptemu_bsdf_sample(pt_scene* sc, uint32_t m, size_t n, const float* lambda, const float* wi, const float* s2, float* f, float* wo, float* pdf) {
    SceneView s{sc->host.blob.data(), sc->host.tex.data(), sc->host.blob.data() + sc->host.blob[PT_HDR_CORE_WORDS]};
    for (size_t i = 0; i < n; ++i) {
        F3 w; material_sample(s, mat_rec(sc, m), lambda[i], 0.5f, 0.5f, s2[2 * i], s2[2 * i + 1], f3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]), &f[i], &w, &pdf[i]);
        wo[3 * i] = w.x; wo[3 * i + 1] = w.y; wo[3 * i + 2] = w.z;
    }
    return PT_OK;
}
pt_status ptemu_bsdf_eval(pt_scene* sc, uint32_t m, size_t n, const float* lambda, const float* wi, const float* wo, float* f, float* pdf) {
    SceneView s{sc->host.blob.data(), sc->host.tex.data(), sc->host.blob.data() + sc->host.blob[PT_HDR_CORE_WORDS]};
    for (size_t i = 0; i < n; ++i)
        material_bsdf(s, mat_rec(sc, m), lambda[i], 0.5f, 0.5f, f3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]), f3(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]), &f[i], &pdf[i]);
    return PT_OK;
}
pt_status ptemu_emission(pt_scene* sc, uint32_t m, size_t n, const float* lambda, const float* wi, float* e) {
    SceneView s{sc->host.blob.data(), sc->host.tex.data(), sc->host.blob.data() + sc->host.blob[PT_HDR_CORE_WORDS]};
    for (size_t i = 0; i < n; ++i) e[i] = material_emission(s, mat_rec(sc, m), lambda[i], f3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]));
    return PT_OK;
}
pt_status ptemu_curve_eval(pt_scene* sc, uint32_t c, size_t n, const float* lambda, float* v) {
    SceneView s{sc->host.blob.data(), sc->host.tex.data(), sc->host.blob.data() + sc->host.blob[PT_HDR_CORE_WORDS]};
    for (size_t i = 0; i < n; ++i) v[i] = curve_eval(s, sc->host.curve_offsets[c], lambda[i]);
    return PT_OK;
}
}
