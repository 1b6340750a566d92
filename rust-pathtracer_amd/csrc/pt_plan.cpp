#include "pt_plan.h"

#include <cmath>

namespace pth {

std::vector<uint32_t> shard_pixels(uint32_t width, uint32_t height, uint32_t tw, uint32_t th, uint32_t shard_index, uint32_t shard_count) {
    struct T { uint32_t x0, x1, y0, y1; };
    std::vector<T> tiles;
    uint32_t fx = width / tw, fy = height / th, rx = width % tw, ry = height % th;
    for (uint32_t y = 0; y < fy; ++y) for (uint32_t x = 0; x < fx; ++x) tiles.push_back(T{x * tw, x * tw + tw, y * th, y * th + th});
    if (rx) for (uint32_t y = 0; y < fy; ++y) tiles.push_back(T{fx * tw, fx * tw + rx, y * th, y * th + th});
    if (ry) {
        for (uint32_t x = 0; x < fx; ++x) tiles.push_back(T{x * tw, x * tw + tw, fy * th, fy * th + ry});
        if (rx) tiles.push_back(T{fx * tw, fx * tw + rx, fy * th, fy * th + ry});
    }
    std::vector<uint32_t> mine;
    for (size_t t = 0; t < tiles.size(); ++t)
        if (!shard_count || PT_TILE_SHARD((uint32_t)t, fx, shard_count) == shard_index) mine.push_back((uint32_t)t);
    // The order of the tiles in the slot space is free (a pixel's samples are keyed by its id, its sums are its own), and it decides
    // how evenly the work falls on the workgroups, each of which takes a run of consecutive slots = a few consecutive tiles: in film
    // order those are neighbours and cost alike (a run of bright floor, a run of dark wall); taken with a stride coprime to their
    // number (about 0.382 of it), every run mixes tiles from all over the film.
    const size_t n = mine.size();
    size_t stride = (size_t)((double)n * 0.3819660112501051);
    auto gcd = [](size_t a, size_t b) { while (b) { size_t t = a % b; a = b; b = t; } return a; };
    if (stride < 1) stride = 1;
    while (gcd(stride, n ? n : 1) != 1) ++stride;
    std::vector<uint32_t> px;
    for (size_t i = 0; i < n; ++i) {
        const T& tile = tiles[mine[(i * stride) % n]];
        for (uint32_t y = tile.y0; y < tile.y1; ++y)
            for (uint32_t x = tile.x0; x < tile.x1; ++x) px.push_back(y * width + x);
    }
    return px;
}

std::vector<Pass> plan_passes(uint32_t n_pixels, uint32_t first_sample, uint32_t sample_count, uint32_t capacity, uint32_t phase_samples) {
    std::vector<Pass> passes;
    if (n_pixels == 0 || sample_count == 0) return passes;
    // A pass must be able to hold one whole phase (10 samples — or all of them for the naive renderer —, or the whole range
    // if shorter) of its pixels, because the per-phase partial sums of tiled.rs:366-391 live in registers of the
    // accumulate kernel.
    const uint32_t period = phase_samples ? phase_samples : 10;
    uint32_t phase = sample_count < period ? sample_count : period;
    uint32_t max_chunk = capacity / phase;
    if (max_chunk == 0) max_chunk = 1;
    uint32_t n_chunks = (n_pixels + max_chunk - 1) / max_chunk;
    uint32_t chunk = (n_pixels + n_chunks - 1) / n_chunks;   // even split
    for (uint32_t p0 = 0; p0 < n_pixels; p0 += chunk) {
        uint32_t pc = (n_pixels - p0 < chunk) ? n_pixels - p0 : chunk;
        uint32_t max_s = capacity / pc;
        if (max_s < phase) max_s = phase;                     // only when capacity < phase (degenerate)
        uint32_t s = first_sample, end = first_sample + sample_count;
        while (s < end) {
            uint32_t take = end - s; if (take > max_s) take = max_s;
            uint32_t stop = s + take;
            if (stop < end) {                                 // end the pass on a phase boundary
                uint32_t aligned = (stop / period) * period;
                if (aligned > s) stop = aligned;
            }
            passes.push_back(Pass{p0, pc, s, stop - s});
            s = stop;
        }
    }
    return passes;
}

ptd::CameraParams camera_params(const pt_camera& c, float aspect_ratio) {
    using namespace ptd;
    CameraParams cam;
    F3 look_from = f3(c.look_from[0], c.look_from[1], c.look_from[2]);
    F3 look_at = f3(c.look_at[0], c.look_at[1], c.look_at[2]);
    F3 v_up = normalize(f3(c.v_up[0], c.v_up[1], c.v_up[2]));          // src/parsing/cameras.rs:139
    F3 direction = normalize(sub(look_at, look_from));
    cam.kind = c.kind; cam.span_x = cam.span_y = 0.0f; cam.w = f3(0.0f, 0.0f, 0.0f);
    if (c.kind == PT_CAMERA_PANORAMA) {  // PanoramaCamera::new (src/camera/panorama_camera.rs:18-62)
        F3 w = direction, u = normalize(cross(v_up, w)), v = normalize(cross(w, u));
        cam.origin = look_from; cam.u = u; cam.v = v; cam.w = w;
        cam.span_x = pt_clamp(c.fov[0] * 0.017453292519943295f, 0.0f, 6.283185307179586f);
        cam.span_y = pt_clamp(c.fov[1] * 0.017453292519943295f, 0.0f, 3.141592653589793f);
        cam.lower_left = cam.horizontal = cam.vertical = f3(0.0f, 0.0f, 0.0f); cam.aperture_diameter = 0.0f;
        return cam;
    }
    float theta = c.vfov * 0.017453292519943295f;                      // f32::to_radians
    float half_height = std::tan(theta / 2.0f);
    float half_width = aspect_ratio * half_height;
    F3 w = neg(direction);
    F3 u = neg(normalize(cross(v_up, w)));
    F3 v = normalize(cross(w, u));
    cam.origin = look_from; cam.u = u; cam.v = v;
    cam.lower_left = sub(sub(sub(look_from, mul(mul(u, half_width), c.focal_distance)), mul(mul(v, half_height), c.focal_distance)),
                         mul(w, c.focal_distance));
    cam.horizontal = mul(mul(mul(u, 2.0f), half_width), c.focal_distance);
    cam.vertical = mul(mul(mul(v, 2.0f), half_height), c.focal_distance);
    cam.aperture_diameter = c.aperture_diameter;
    return cam;
}

bool normalize_render_desc(const pt_render_desc& in, uint32_t camera_count, pt_render_desc* out, std::string* error) {
    pt_render_desc rd = in;
    if (rd.tile_width == 0) rd.tile_width = 32;
    if (rd.tile_height == 0) rd.tile_height = 32;
    if (rd.hero_wavelengths == 0) rd.hero_wavelengths = 1;
    if (rd.phase_samples == 0) rd.phase_samples = 10;
    if (rd.sample_count == 0) { rd.first_sample = 0; rd.sample_count = rd.spp; }
    if (rd.width == 0 || rd.height == 0 || rd.spp == 0) { *error = "width, height and spp must be positive"; return false; }
    if ((uint64_t)rd.width * rd.height > 0xffffffffull) { *error = "film too large"; return false; }
    if (rd.camera_index >= camera_count) { *error = "camera_index out of range"; return false; }
    if (rd.shard_count > 0 && rd.shard_index >= rd.shard_count) { *error = "shard_index >= shard_count"; return false; }
    if (rd.light_samples > PT_MAX_LIGHT_SAMPLES) { *error = "light_samples > 8 is not supported"; return false; }
    if (rd.hero_wavelengths != 1 && rd.hero_wavelengths != 4) { *error = "hero_wavelengths must be 1 or 4"; return false; }
    if (rd.first_sample + rd.sample_count > rd.spp) { *error = "sample range exceeds spp"; return false; }
    if (!(rd.wavelength_hi >= rd.wavelength_lo)) { *error = "bad wavelength bounds"; return false; }
    if (rd.max_bounces > 64) { *error = "max_bounces > 64"; return false; }
    if (rd.medium_aware && rd.hero_wavelengths != 1) { *error = "the medium-aware walk carries one wavelength"; return false; }
    *out = rd;
    return true;
}

}  // namespace pth
