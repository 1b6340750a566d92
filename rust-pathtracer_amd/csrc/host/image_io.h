// image_io.h — the image readers the reference's texture parser needs (src/parsing/texture.rs:48-153): 8-bit images through
// the `image` crate (PNG, BMP) as greyscale or RGBA in [0,1], Radiance .hdr as float RGB + alpha_fill, OpenEXR RGBA.
#ifndef PT_IMAGE_IO_H
#define PT_IMAGE_IO_H
#include <cstdint>
#include <string>
#include <vector>

namespace pth {

struct Image {
    uint32_t width = 0, height = 0, channels = 0;  // channels 1 (grey) or 4 (RGBA); row-major, top row first
    std::vector<float> data;
};

// parse_bitmap (texture.rs:133-146): into_luma8() / 255
bool read_grey8(const std::string& path, Image* out, std::string* error);
// parse_rgba (texture.rs:48-72): into_rgba8() / 255
bool read_rgba8(const std::string& path, Image* out, std::string* error);
// parse_hdr (texture.rs:102-131): RGBE -> f32 RGB, alpha = alpha_fill
bool read_hdr(const std::string& path, float alpha_fill, Image* out, std::string* error);
// parse_exr (texture.rs:74-100): first RGBA layer (missing channels: 0, alpha 1); uncompressed, RLE, ZIPS and ZIP blocks
bool read_exr(const std::string& path, Image* out, std::string* error);

}  // namespace pth
#endif
