// Host-side experiment (not part of the product or the tests): replays the engine's BVH walk on the emulated lane
// logic and reports what a wave64 would execute: per-ray event traces are grouped 64 at a time in call order (the
// order of queue items) and the lock-step cost of the while-while walk is computed from them.
#include <cstdio>
#include <cstdlib>
#include <vector>
struct Stats { unsigned long long box_tests, box_exact, tri_tests, instance_tests, rays_without_light, rays_dead, rays_live; };
static Stats g_stats;
// trace: per ray, list of rounds; each round = (steps, leaf kind 0 none / 3 tri / 4 inst)
struct Round { int steps; int leaf; };
static std::vector<std::vector<Round>> g_rays;
static inline void pt_event(int code) {
    if (code == 0) { g_rays.emplace_back(); return; }
    auto& r = g_rays.back();
    if (code == 1) { r.push_back(Round{0, 0}); return; }
    if (r.empty()) r.push_back(Round{0, 0});  // the sweep form has no walk rounds
    if (code == 2) { r.back().steps++; return; }
    r.back().leaf = code;
}
#define PT_STAT(counter) (g_stats.counter++)
#define PT_STAT_EVENT(code) pt_event(code)
static std::vector<float> g_raydata;
#define PT_STAT_RAY(o, d) do { g_raydata.push_back((o).x); g_raydata.push_back((o).y); g_raydata.push_back((o).z); g_raydata.push_back((d).x); g_raydata.push_back((d).y); g_raydata.push_back((d).z); } while (0)
#include "../tests/host_emulation/ptemu.cpp"
extern "C" void ptemu_set_flags(pt_scene* sc, unsigned flags) { sc->host.blob[PT_HDR_FLAGS] |= flags; }
extern "C" size_t ptemu_trace_dump(int* out, size_t cap) {
    // flat: per ray: n_rounds, then (steps, leaf) pairs
    size_t k = 0;
    for (auto& r : g_rays) { if (k + 1 + 2 * r.size() > cap) break; out[k++] = (int)r.size(); for (auto& x : r) { out[k++] = x.steps; out[k++] = x.leaf; } }
    return k;
}
extern "C" size_t ptemu_ray_dump(float* out, size_t cap) { size_t n = g_raydata.size() < cap ? g_raydata.size() : cap; for (size_t i = 0; i < n; ++i) out[i] = g_raydata[i]; g_raydata.clear(); return n; }
extern "C" void ptemu_wave_stats(double* out) {
    // out: rays, avg steps/lane, avg rounds/lane, wave: sum over rounds of max steps, rounds(max), tri rounds, inst rounds
    size_t n = g_rays.size();
    double steps = 0, rounds = 0, w_steps = 0, w_rounds = 0, w_tri = 0, w_inst = 0, waves = 0;
    for (size_t w0 = 0; w0 + 64 <= n; w0 += 64) {
        size_t maxr = 0;
        for (size_t l = 0; l < 64; ++l) { auto& r = g_rays[w0 + l]; rounds += r.size(); for (auto& x : r) steps += x.steps; if (r.size() > maxr) maxr = r.size(); }
        for (size_t k = 0; k < maxr; ++k) {
            int ms = 0; bool tri = false, inst = false;
            for (size_t l = 0; l < 64; ++l) { auto& r = g_rays[w0 + l]; if (k < r.size()) { if (r[k].steps > ms) ms = r[k].steps; tri |= r[k].leaf == 3; inst |= r[k].leaf == 4; } }
            w_steps += ms; w_tri += tri; w_inst += inst;
        }
        w_rounds += maxr; waves += 1;
    }
    double lanes = waves * 64;
    out[0] = (double)n; out[1] = steps / lanes; out[2] = rounds / lanes; out[3] = w_steps / waves; out[4] = w_rounds / waves; out[5] = w_tri / waves; out[6] = w_inst / waves;
    g_rays.clear(); g_stats = Stats{};
}
extern "C" void ptemu_counter_stats(unsigned long long* out) {
    out[0] = g_stats.box_tests; out[1] = g_stats.box_exact; out[2] = g_stats.tri_tests; out[3] = g_stats.instance_tests; out[4] = (unsigned long long)g_rays.size();
}
