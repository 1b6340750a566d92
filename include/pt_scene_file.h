/*
 * pt_scene_file.h — the reference's TOML front end for the PT path, as a C ABI (libptscene.so, no GPU code).
 *
 * Replaces, for the data the PT path consumes:
 *   get_config / TOMLConfig / RenderSettings      src/parsing/mod.rs:565-582, src/parsing/config.rs:9-164
 *   construct_world (scene + libraries + scan)    src/parsing/mod.rs:145-563
 *   CurveData / CurveDataOrReference              src/parsing/curves.rs:43-72,298-400
 *   TextureData / parse_texture_stack             src/parsing/texture.rs:19-326
 *   MaterialData::resolve                         src/parsing/material.rs:55-153
 *   MeshData / tobj loading                       src/parsing/meshes.rs:10-157
 *   InstanceData / Transform3Data / AggregateData src/parsing/instance.rs:17-118, src/parsing/primitives.rs:10-79
 *   CameraData / parse_cameras                    src/parsing/cameras.rs:69-204
 *   EnvironmentData / parse_environment           src/parsing/environment.rs:19-181
 *   TonemapSettings                               src/parsing/tonemap.rs:5-31
 * The result is a pt_scene_desc (include/pt_api.h) ready for pt_scene_create, and pt_render_desc / pt_output_desc per
 * [[render_settings]] entry.  Like serde's deny_unknown_fields, unknown keys are errors.  Not on this path and rejected
 * with PT_ERR_UNSUPPORTED: mediums, LT integrator, realistic cameras.
 *
 * File names inside the TOML files are used as written (the reference resolves them against the working directory);
 * when a file is not found there it is looked up under the root given to pt_scene_file_set_root.
 */
#ifndef PT_SCENE_FILE_H
#define PT_SCENE_FILE_H
#include "pt_api.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_config pt_config;
typedef struct pt_scene_file pt_scene_file;

enum { PT_RENDERER_NAIVE = 0, PT_RENDERER_TILED = 1 };       /* RendererType, src/parsing/config.rs:109-121 */
enum { PT_INTEGRATOR_PT = 0, PT_INTEGRATOR_LT = 1 };         /* IntegratorKind, src/parsing/config.rs:16-31 */

typedef struct pt_render_settings {   /* RenderSettings, src/parsing/config.rs:43-63; absent Options are -1 / has_* = 0 */
    const char* filename;             /* NULL when absent (output_film then uses "beauty", src/renderer/mod.rs:28) */
    uint32_t width, height;
    int32_t integrator;
    uint32_t light_samples;           /* PT */
    int32_t medium_aware;             /* PT */
    uint32_t camera_samples;          /* LT */
    int32_t min_bounces, max_bounces;
    int32_t hwss;
    int32_t threads;
    uint32_t min_samples;
    int32_t max_samples;
    const char* camera_id;
    int32_t russian_roulette, only_direct;
    int32_t has_wavelength_bounds;
    float wavelength_lo, wavelength_hi;
    int32_t has_premultiply;
    float premultiply;
    int32_t colorspace;               /* PT_COLORSPACE_* */
    int32_t tonemap;                  /* PT_TONEMAP_* */
    int32_t has_exposure;
    float exposure, key_value, white_point;
    int32_t luminance_only, silenced;
} pt_render_settings;

const char* pt_scene_file_last_error(void);
void pt_scene_file_set_root(const char* directory);

/* get_config (src/parsing/mod.rs:565-582) */
pt_status pt_config_load(const char* path, pt_config** out);
void pt_config_free(pt_config* config);
const char* pt_config_scene_file(const pt_config* config);                       /* default_scene_file */
int32_t pt_config_renderer(const pt_config* config, uint32_t* tile_width, uint32_t* tile_height);
uint32_t pt_config_render_settings_count(const pt_config* config);
pt_status pt_config_render_settings(const pt_config* config, uint32_t index, pt_render_settings* out);
/* What HipRenderer::render passes per render-settings entry: defaults of
 * src/integrator/mod.rs:59-105 (wavelength bounds [380,750], min_bounces 4), max_bounces required, camera 0 (tiled.rs:378). */
pt_status pt_config_render_desc(const pt_config* config, uint32_t index, uint64_t seed, pt_render_desc* out);
/* output_film's parameters for that entry (src/renderer/mod.rs:24-35): factor *= premultiply */
pt_status pt_config_output_desc(const pt_config* config, uint32_t index, float factor, pt_output_desc* out);

/* construct_world (src/parsing/mod.rs:145-563).  `config` supplies the cameras in use and their aspect ratios; may be NULL
 * (every camera of the scene file is loaded, in file order). */
pt_status pt_scene_file_load(const char* scene_path, const pt_config* config, pt_scene_file** out);
void pt_scene_file_free(pt_scene_file* scene);
const pt_scene_desc* pt_scene_file_desc(const pt_scene_file* scene);
/* name lookups, for tools and tests: packed MaterialId / indices, -1 when unknown */
int64_t pt_scene_file_material(const pt_scene_file* scene, const char* name);
int32_t pt_scene_file_curve(const pt_scene_file* scene, const char* name);
int32_t pt_scene_file_texture(const pt_scene_file* scene, const char* name);
int32_t pt_scene_file_camera(const pt_scene_file* scene, const char* camera_id);
uint32_t pt_scene_file_warning_count(const pt_scene_file* scene);
const char* pt_scene_file_warning(const pt_scene_file* scene, uint32_t index);

/* The image readers of the texture parser (src/parsing/texture.rs:48-153), for tools: PT_IMAGE_GREY8 = parse_bitmap
 * (1 channel), PT_IMAGE_RGBA8 = parse_rgba, PT_IMAGE_HDR = parse_hdr (alpha = alpha_fill), PT_IMAGE_EXR = parse_exr (4 channels
 * each).  `*data` is width * height * channels floats, row-major, top row first; release it with pt_image_free. */
enum { PT_IMAGE_GREY8 = 0, PT_IMAGE_RGBA8 = 1, PT_IMAGE_HDR = 2, PT_IMAGE_EXR = 3 };
pt_status pt_image_read(const char* path, int32_t kind, float alpha_fill, uint32_t* width, uint32_t* height, uint32_t* channels, float** data);
void pt_image_free(float* data);

#ifdef __cplusplus
}
#endif
#endif /* PT_SCENE_FILE_H */
