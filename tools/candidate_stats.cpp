// Host-side experiment (not part of the product or the tests): the emulated lane logic with a trace of phase 3 of the leaf sweep —
// per ray, the kinds of the leaves it tests, in order (T = triangle, A = analytic shape).  tools/candidate_stats.py groups the rays
// 64 at a time and prices the loop policies a wave could follow.
#include <string>
#include <vector>
static std::vector<std::string> g_seq;
static inline void pt_event(int code) {
    if (code == 0) { g_seq.emplace_back(); return; }
    if (code == 3) g_seq.back().push_back('T');
    if (code == 4) g_seq.back().push_back('A');
}
#define PT_STAT_EVENT(code) pt_event(code)
#include "../tests/host_emulation/ptemu.cpp"
extern "C" size_t ptemu_seq_dump(char* out, size_t cap) {
    size_t k = 0;
    for (auto& s : g_seq) { if (k + s.size() + 1 > cap) break; for (char c : s) out[k++] = c; out[k++] = '\n'; }
    g_seq.clear();
    return k;
}
