// ptcompare — the reference's src/bin/compare_exr.rs: two EXR images in, their difference out.
//
//   ptcompare --compare-file a.exr --ground-truth-file b.exr --output-file out [--mode absolute_difference|rmse|relative]
//
// absolute_difference (default) and relative write <output>.exr (RGB of the per-channel result); rmse prints the
// "minmax" line and writes <output>.png through the viridis gradient, exactly as the reference does (compare_exr.rs:70-170).
// The arithmetic runs on the GPU (pt_compare_films); the EXR reader is the texture parser's (pt_image_read).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/pt_scene_file.h"

int main(int argc, char** argv) {
    std::string a, b, out, mode = "absolute_difference";
    for (int i = 1; i < argc; ++i) {
        std::string k = argv[i];
        auto val = [&](std::string* dst) { if (i + 1 >= argc) return false; *dst = argv[++i]; return true; };
        bool ok = k == "--compare-file" ? val(&a) : k == "--ground-truth-file" ? val(&b) : k == "--output-file" ? val(&out) : k == "--mode" ? val(&mode) : false;
        if (!ok) { fprintf(stderr, "usage: ptcompare --compare-file A.exr --ground-truth-file B.exr --output-file OUT [--mode absolute_difference|rmse|relative]\n"); return 2; }
    }
    if (a.empty() || b.empty() || out.empty()) { fprintf(stderr, "ptcompare: --compare-file, --ground-truth-file and --output-file are required\n"); return 2; }
    const int m = mode == "rmse" ? PT_COMPARE_RMSE : (mode == "relative" ? PT_COMPARE_RELATIVE : PT_COMPARE_ABSOLUTE);  // Mode::new, compare_exr.rs:45-52
    uint32_t w0, h0, c0, w1, h1, c1; float *d0 = nullptr, *d1 = nullptr;
    if (pt_image_read(a.c_str(), PT_IMAGE_EXR, 0.0f, &w0, &h0, &c0, &d0) != PT_OK || pt_image_read(b.c_str(), PT_IMAGE_EXR, 0.0f, &w1, &h1, &c1, &d1) != PT_OK) {
        printf("failed to parse images for some reason. check whether the paths exist (%s)\n", pt_scene_file_last_error());
        return 1;
    }
    if (w0 != w1 || h0 != h1) { fprintf(stderr, "image dimensions must match\n"); return 1; }   // the reference asserts
    std::vector<float> res((size_t)w0 * h0 * 4);
    pt_compare_stats st;
    if (pt_compare_films(w0, h0, d0, d1, m, res.data(), &st) != PT_OK) { fprintf(stderr, "pt_compare_films: %s\n", pt_last_error()); return 1; }
    std::string base = out;
    if (base.size() > 4 && base.compare(base.size() - 4, 4, ".exr") == 0) base.resize(base.size() - 4);
    pt_status ws;
    if (m == PT_COMPARE_RMSE) {
        printf("minmax: %g -> %g\n", st.pixel_min, st.pixel_max);
        std::vector<uint8_t> rgba((size_t)w0 * h0 * 4);
        for (size_t i = 0; i < (size_t)w0 * h0; ++i) {
            for (int c = 0; c < 3; ++c) rgba[4 * i + c] = (uint8_t)(res[4 * i + c] * 255.0f);   // (r * 255.0) as u8
            rgba[4 * i + 3] = 255;
        }
        ws = pt_write_png((base + ".png").c_str(), w0, h0, rgba.data(), PT_COLORSPACE_SRGB);
    } else {
        std::vector<float> rgb((size_t)w0 * h0 * 3);
        for (size_t i = 0; i < (size_t)w0 * h0; ++i) for (int c = 0; c < 3; ++c) rgb[3 * i + c] = res[4 * i + c];
        ws = pt_write_exr((base + ".exr").c_str(), w0, h0, rgb.data(), PT_COLORSPACE_SRGB);
    }
    pt_image_free(d0); pt_image_free(d1);
    if (ws != PT_OK) { fprintf(stderr, "failed to write the result: %s\n", pt_last_error()); return 1; }
    printf("linf %g %g %g  rmse %g\n", st.linf[0], st.linf[1], st.linf[2], st.rmse);
    printf("saved, exiting\n");
    return 0;
}
