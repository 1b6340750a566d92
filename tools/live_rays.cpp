// Host-side experiment (not part of the product or the tests): the emulated lane logic with counters of what k_shadow's lanes hold — light-sample rays
// that are traced (factor != 0), dead ones (factor 0: the sample lies below the surface's horizon or its BSDF value is 0), rays whose line meets no light —
// and, per traced ray of either kind (closest-hit segment / light sample), the leaves phase 3 of the sweep tests.  tools/live_rays.py prints the shares.
#include <string>
#include <vector>
struct Stats { unsigned long long rays_live, rays_dead, rays_without_light, box_tests, box_exact, tri_tests, instance_tests; };
static Stats g_stats;
static std::vector<int> g_leaves;   // per world_hit call: leaves tested in phase 3
static inline void pt_event(int code) { if (code == 0) g_leaves.push_back(0); else if (code == 3 || code == 4) g_leaves.back()++; }
#define PT_STAT(counter) (g_stats.counter++)
#define PT_STAT_EVENT(code) pt_event(code)
#include "../tests/host_emulation/ptemu.cpp"
extern "C" void ptemu_live_stats(unsigned long long* out) { out[0] = g_stats.rays_live; out[1] = g_stats.rays_dead; out[2] = g_stats.rays_without_light; out[3] = g_leaves.size(); g_stats = Stats{}; }
extern "C" size_t ptemu_leaves_dump(int* out, size_t cap) { size_t n = g_leaves.size() < cap ? g_leaves.size() : cap; for (size_t i = 0; i < n; ++i) out[i] = g_leaves[i]; g_leaves.clear(); return n; }
