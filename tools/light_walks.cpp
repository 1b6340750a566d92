// Host-side experiment (not part of the product or the tests): the emulated lane logic with a record of every mesh WALK (mesh_walk's while-while loop: bounded
// searches and every search the grouped sweep does not take) — the ray in the instance's own space, its bound, the closest hit it came with, the kind of search,
// the numbers of box and triangle tests, and whether the search ended at the mesh.  tools/light_walks.py asks what a cheaper certificate would have decided.
#include <vector>
static std::vector<float> g_walks;   // 12 floats a walk: o, d, bound, closest, stop, boxes, triangles, over
static inline void pt_walk_event(int code) {
    if (g_walks.empty()) return;
    if (code == 5) g_walks[g_walks.size() - 3] += 1.0f;
    if (code == 6) g_walks[g_walks.size() - 2] += 1.0f;
}
#define PT_STAT_EVENT(code) pt_walk_event(code)
#define PT_STAT_WALK(lo, ld, bound, closest, stop) do { const float r_[12] = {(lo).x, (lo).y, (lo).z, (ld).x, (ld).y, (ld).z, (bound), (closest), (float)(stop), 0.0f, 0.0f, 0.0f}; g_walks.insert(g_walks.end(), r_, r_ + 12); } while (0)
#define PT_STAT_WALK_END(over) do { g_walks[g_walks.size() - 1] = (over) ? 1.0f : 0.0f; } while (0)
#include "../tests/host_emulation/ptemu.cpp"
extern "C" size_t ptemu_walks_dump(float* out, size_t cap) {
    size_t n = g_walks.size() < cap ? g_walks.size() : cap;
    for (size_t i = 0; i < n; ++i) out[i] = g_walks[i];
    g_walks.clear();
    return n;
}
