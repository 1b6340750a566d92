// Host-side experiment (not part of the product or the tests): the emulated lane logic with a trace of every mesh walk (mesh_walk's
// while-while loop) — per walk its kind (C = closest hit wanted, L = light ray, stops at the first opaque hit, E = environment ray,
// stops at any hit) and, in order, b = a node's box test, T = a triangle test.  tools/walk_stats.py groups the walks 64 at a time and
// prices the loop policies a wave could follow.
#include <string>
#include <vector>
static std::vector<std::string> g_seq;
static inline void pt_event(int code) {
    if (code >= 7) { g_seq.emplace_back(1, "CLE"[code - 7]); return; }
    if (g_seq.empty()) return;
    if (code == 5) g_seq.back().push_back('b');
    if (code == 6) g_seq.back().push_back('T');
}
#define PT_STAT_EVENT(code) pt_event(code)
#include "../tests/host_emulation/ptemu.cpp"
extern "C" size_t ptemu_seq_dump(char* out, size_t cap) {
    size_t k = 0;
    for (auto& s : g_seq) { if (k + s.size() + 1 > cap) break; for (char c : s) out[k++] = c; out[k++] = '\n'; }
    g_seq.clear();
    return k;
}
