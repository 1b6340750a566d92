// ptcli — the reference's command line (src/bin/main.rs:30-199) on top of libptscene.so (TOML front end) and libptamd.so
// (HIP engine): read the config, build the scene, render every [[render_settings]] entry on the GPU, write
// output/<filename>.exr and .png through the film output stage (src/renderer/mod.rs:24-80).
//
//   ptcli [--config data/config.toml] [--scene FILE] [-n|--dry-run] [--stdout-log-level L] [--write-log-level L]
//         [--root DIR] [--output-dir DIR] [--seed N] [--write-film]
//
// --config / --scene / --dry-run / the two log-level options are the reference's (the log levels only select how much
// this program prints: warnings are shown from "warn" up).  --root is where relative file names inside the TOML files
// are looked up when they are not found from the working directory; --write-film also stores the raw XYZ film as
// <filename>.npy for tools/compare_films.py.
#include <sys/stat.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/pt_scene_file.h"

namespace {

struct Options {
    std::string config = "data/config.toml", scene, root, output_dir = "output", stdout_log_level = "warn";
    bool has_scene = false, dry_run = false, write_film = false;
    uint32_t hero = 0;   // --hero-wavelengths: 0 = as the render settings say (1)
    uint64_t seed = 1;
};

int usage(const char* msg) {
    if (msg) fprintf(stderr, "error: %s\n", msg);
    fprintf(stderr, "usage: ptcli [--config FILE] [--scene FILE] [-n|--dry-run] [--stdout-log-level LEVEL] [--write-log-level LEVEL]\n"
                    "             [--root DIR] [--output-dir DIR] [--seed N] [--write-film] [--hero-wavelengths 1|4]\n");
    return 2;
}

bool write_npy(const std::string& path, const float* data, uint32_t h, uint32_t w) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    std::string dict = "{'descr': '<f4', 'fortran_order': False, 'shape': (" + std::to_string(h) + ", " + std::to_string(w) + ", 4), }";
    while ((10 + dict.size() + 1) % 64) dict += ' ';
    dict += '\n';
    uint8_t head[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (uint8_t)(dict.size() & 255), (uint8_t)(dict.size() >> 8)};
    bool ok = fwrite(head, 1, 10, f) == 10 && fwrite(dict.data(), 1, dict.size(), f) == dict.size() &&
              fwrite(data, sizeof(float), (size_t)w * h * 4, f) == (size_t)w * h * 4;
    fclose(f);
    return ok;
}

}  // namespace

int main(int argc, char** argv) {
    Options o;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto value = [&](std::string* dst) { if (i + 1 >= argc) return false; *dst = argv[++i]; return true; };
        std::string v;
        if (a == "--config") { if (!value(&o.config)) return usage("--config needs a value"); }
        else if (a == "--scene") { if (!value(&o.scene)) return usage("--scene needs a value"); o.has_scene = true; }
        else if (a == "-n" || a == "--dry-run") o.dry_run = true;
        else if (a == "--stdout-log-level") { if (!value(&o.stdout_log_level)) return usage("--stdout-log-level needs a value"); }
        else if (a == "--write-log-level") { if (!value(&v)) return usage("--write-log-level needs a value"); }
        else if (a == "--root") { if (!value(&o.root)) return usage("--root needs a value"); }
        else if (a == "--output-dir") { if (!value(&o.output_dir)) return usage("--output-dir needs a value"); }
        else if (a == "--seed") { if (!value(&v)) return usage("--seed needs a value"); o.seed = strtoull(v.c_str(), nullptr, 10); }
        else if (a == "--write-film") o.write_film = true;
        else if (a == "--hero-wavelengths") { if (!value(&v)) return usage("--hero-wavelengths needs a value"); o.hero = (uint32_t)strtoul(v.c_str(), nullptr, 10); }
        else if (a == "-h" || a == "--help") { usage(nullptr); return 0; }
        else return usage(("unknown option " + a).c_str());
    }
    const bool verbose = o.stdout_log_level == "info" || o.stdout_log_level == "debug" || o.stdout_log_level == "trace";
    const bool warnings = verbose || o.stdout_log_level == "warn";
    if (!o.root.empty()) pt_scene_file_set_root(o.root.c_str());

    pt_config* config = nullptr;
    if (pt_config_load(o.config.c_str(), &config) != PT_OK) {
        fprintf(stderr, "couldn't read config.toml, %s\n", pt_scene_file_last_error());   // main.rs:115-121
        return 1;
    }
    std::string scene_path = o.has_scene ? o.scene : pt_config_scene_file(config);        // main.rs:133
    pt_scene_file* scene_file = nullptr;
    if (pt_scene_file_load(scene_path.c_str(), config, &scene_file) != PT_OK) {
        fprintf(stderr, "failed to construct the scene: %s\n", pt_scene_file_last_error());  // main.rs:139-149
        pt_config_free(config);
        return 1;
    }
    if (warnings) for (uint32_t k = 0; k < pt_scene_file_warning_count(scene_file); ++k) fprintf(stderr, "warning: %s\n", pt_scene_file_warning(scene_file, k));
    const pt_scene_desc* desc = pt_scene_file_desc(scene_file);
    if (verbose) printf("scene %s: %u instances, %u meshes, %zu triangles, %u materials, %u curves\n", scene_path.c_str(), desc->instance_count, desc->mesh_count,
                        desc->index_count / 3, desc->material_count, desc->curve_count);

    printf("constructing renderer\n");
    mkdir(o.output_dir.c_str(), 0777);                                                     // main.rs:155-158
    int rc = 0;
    if (!o.dry_run) {
        pt_scene* scene = nullptr;
        if (pt_scene_create(desc, &scene) != PT_OK) { fprintf(stderr, "pt_scene_create: %s\n", pt_last_error()); rc = 1; }
        const uint32_t n = pt_config_render_settings_count(config);
        for (uint32_t i = 0; rc == 0 && i < n; ++i) {
            pt_render_settings rs; pt_render_desc rd; pt_output_desc od;
            pt_config_render_settings(config, i, &rs);
            if (pt_config_render_desc(config, i, o.seed, &rd) != PT_OK) {
                // the reference skips render settings whose integrator it cannot construct (src/renderer/tiled.rs:560-566)
                fprintf(stderr, "skipping render settings %u: %s\n", i, pt_scene_file_last_error());
                continue;
            }
            if (o.hero) rd.hero_wavelengths = o.hero;   // engine extension (not in the reference's files): 4 wavelengths per path
            std::vector<float> film((size_t)rd.width * rd.height * 4);
            pt_profile prof;
            printf("rendering %ux%u, %u spp, max_bounces %u, light_samples %u\n", rd.width, rd.height, rd.spp, rd.max_bounces, rd.light_samples);
            if (pt_render(scene, &rd, film.data(), &prof) != PT_OK) { fprintf(stderr, "pt_render: %s\n", pt_last_error()); rc = 1; break; }
            // Profile::pretty_print (src/profile.rs:20-34)
            const double total = (double)(prof.camera_rays + prof.bounce_rays + prof.shadow_rays + prof.light_rays);
            printf("took %.3fs\n", prof.seconds);
            printf("%llu camera rays, %llu bounce rays, %llu shadow rays, %llu light rays, %llu environment hits\n", (unsigned long long)prof.camera_rays,
                   (unsigned long long)prof.bounce_rays, (unsigned long long)prof.shadow_rays, (unsigned long long)prof.light_rays, (unsigned long long)prof.env_hits);
            printf("%.1f rays per second, %.3f Msamples/s\n", total / prof.seconds, (double)rd.width * rd.height * rd.spp / prof.seconds * 1e-6);
            pt_config_output_desc(config, i, 1.0f, &od);
            std::vector<uint8_t> rgba((size_t)rd.width * rd.height * 4);
            std::vector<float> linear((size_t)rd.width * rd.height * 3);
            if (pt_output_film(&od, film.data(), rgba.data(), linear.data()) != PT_OK) { fprintf(stderr, "pt_output_film: %s\n", pt_last_error()); rc = 1; break; }
            const std::string base = o.output_dir + "/" + (rs.filename ? rs.filename : "beauty");  // src/renderer/mod.rs:27-31
            if (pt_write_exr((base + ".exr").c_str(), rd.width, rd.height, linear.data(), od.colorspace) != PT_OK ||
                pt_write_png((base + ".png").c_str(), rd.width, rd.height, rgba.data(), od.colorspace) != PT_OK) {
                fprintf(stderr, "failed to write files: %s\n", pt_last_error()); rc = 1; break;  // the reference panics here (mod.rs:45-48)
            }
            if (o.write_film && !write_npy(base + ".npy", film.data(), rd.height, rd.width)) { fprintf(stderr, "failed to write %s.npy\n", base.c_str()); rc = 1; break; }
            printf("wrote %s.exr and %s.png\n", base.c_str(), base.c_str());
        }
        if (scene) pt_scene_destroy(scene);
        if (rc == 0) printf("render done\n");
    }
    pt_scene_file_free(scene_file);
    pt_config_free(config);
    return rc;
}
