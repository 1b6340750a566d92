// pt_plan.h — host-side planning shared by the HIP engine: film tiles -> pixel list of a shard,
// sample passes aligned to the reference's 10-sample phases, thin-lens camera frame.
#ifndef PT_PLAN_H
#define PT_PLAN_H
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/pt_api.h"
#include "pt_stages.h"

namespace pth {

// Pixels (linear id y * width + x) this call renders, tile by tile in the reference's tile order
// (TiledRenderer::generate_tiles, src/renderer/tiled.rs:190-277), row-major inside a tile; tiles are dealt
// round-robin to shards (tile t belongs to shard t % shard_count).
std::vector<uint32_t> shard_pixels(uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h,
                                   uint32_t shard_index, uint32_t shard_count);

struct Pass { uint32_t pixel_begin, pixel_count, first_sample, sample_count; };
// Passes over (pixel chunk, sample range).  A pass never splits one of the reference's phases of 10 samples
// (tiled.rs:347-361) unless the requested range itself does, so the film sums keep the reference's order.
std::vector<Pass> plan_passes(uint32_t n_pixels, uint32_t first_sample, uint32_t sample_count, uint32_t capacity, uint32_t phase_samples = 10);

// ProjectiveCamera::new + with_aspect_ratio (src/camera/projective_camera.rs:27-95, 121-133)
ptd::CameraParams camera_params(const pt_camera& c, float aspect_ratio);

bool normalize_render_desc(const pt_render_desc& in, uint32_t camera_count, pt_render_desc* out, std::string* error);

}  // namespace pth
#endif
