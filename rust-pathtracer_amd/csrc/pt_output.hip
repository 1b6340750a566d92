// pt_output.hip — film output stage: the reference's output_film (src/renderer/mod.rs:24-80):
// Tonemapper::initialize (a luminance / log-average reduction over the film), Tonemapper::map per pixel
// (src/tonemap/clamp.rs:76-102, reinhard0.rs:84-103,179-196, reinhard1.rs:88-110,203-231), XYZ -> linear RGB of the colour
// space's primaries, OETF, 8-bit quantisation (src/tonemap/mod.rs:19-37,147-205,316-333), and the two file writers.
//
// On the GPU the reduction is a two-level f64 tree (one partial per workgroup, summed on the host in workgroup order), the
// per-pixel map is one lane per pixel.  The reference sums sequentially (f64 for the luminance-only tonemappers, f32 lanes
// for the x3 variants); the tree sum differs from those by rounding only, which moves l_w by <= 1e-6 relative (f64) or 1e-4
// (x3) and an 8-bit output by at most one code value — that is the parity bar of tests/test_output.py.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pt_api.h"
#include "../../include/pt_numerics.h"
#include "pt_error.h"

namespace {


constexpr int kBlock = 256;

struct OutParams {
    uint32_t n;
    int32_t tonemap, luminance_only, colorspace;
    float exposure_mult, key_value, inv_white2, factor;
    float lw[3];
};

__host__ __device__ inline void xyz_to_rgb(int colorspace, float x, float y, float z, float* r, float* g, float* b) {
    if (colorspace == PT_COLORSPACE_REC2020) {  // XYZ_TO_REC2020_LINEAR, src/tonemap/mod.rs:34-37
        *r = 1.4628067f * x + -0.1840623f * y + -0.2743606f * z;
        *g = -0.5217933f * x + 1.4472381f * y + 0.0677227f * z;
        *b = 0.0349342f * x + -0.0968930f * y + 1.2884099f * z;
    } else {  // XYZ_TO_REC709_LINEAR, src/tonemap/mod.rs:22-32
        *r = 3.24096994f * x + -1.53738318f * y + -0.49861076f * z;
        *g = -0.96924364f * x + 1.8759675f * y + 0.04155506f * z;
        *b = 0.05563008f * x + -0.20397696f * y + 1.05697151f * z;
    }
}
__host__ __device__ inline float oetf(int colorspace, float v) {  // src/tonemap/mod.rs:147-205
    if (colorspace == PT_COLORSPACE_SRGB) return v < 0.0031308f ? (323.0f / 25.0f) * v : (211.0f / 200.0f) * pt_pow(v, 5.0f / 12.0f) - (11.0f / 200.0f);
    return v < 0.01805397f ? 4.5f * v : 1.0992968f * pt_pow(v, 0.45f) - 0.09929682f;
}
__host__ __device__ inline uint8_t quantize(float v) {  // (v * 255).ceil().clamp(0, 255) as u8, mod.rs:327-331
    float s = v * 255.0f;
    if (!(s == s)) return 0;       // NaN as u8 saturates to 0
    float c = pt_floor(s); if (c < s) c += 1.0f;
    if (c < 0.0f) c = 0.0f; if (c > 255.0f) c = 255.0f;
    return (uint8_t)c;
}

// Tonemapper::initialize: sums of ln(delta + value) over pixels whose luminance is not NaN
__global__ void __launch_bounds__(kBlock) k_log_sums(const float4* __restrict__ film, uint32_t n, int x3, double* __restrict__ partial) {
    __shared__ double sh[3][kBlock];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 c = film[i];
        if (c.y != c.y) continue;
        if (x3) { s0 += pt_log64((double)(0.001f + c.x)); s1 += pt_log64((double)(0.001f + c.y)); s2 += pt_log64((double)(0.001f + c.z)); }
        else s1 += pt_log64(0.001 + (double)c.y);
    }
    sh[0][threadIdx.x] = s0; sh[1][threadIdx.x] = s1; sh[2][threadIdx.x] = s2;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) for (int k = 0; k < 3; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) for (int k = 0; k < 3; ++k) partial[blockIdx.x * 3 + k] = sh[k][0];
}

__host__ __device__ inline void tonemap_pixel(const OutParams& p, float x, float y, float z, float* ox, float* oy, float* oz) {
    const float MX = 0.5199467f, MY = 51.48687f, MZ = 1.0180528f;  // MAUVE, src/lib.rs:46
    bool bad;
    if (p.tonemap == PT_TONEMAP_CLAMP) {  // clamp.rs:76-102
        x *= p.factor; y *= p.factor; z *= p.factor;
        bad = !pt_isfinite(x) || !pt_isfinite(y) || !pt_isfinite(z);
        if (bad) { x = MX; y = MY; z = MZ; }
        if (p.luminance_only) {
            float lum = y;
            float new_lum = pt_clamp(lum * p.exposure_mult, 0.0f, 1.0f);
            float sf = new_lum / lum;
            *ox = sf * x; *oy = sf * y; *oz = sf * z;
        } else {
            *ox = pt_max(pt_min(x * p.exposure_mult, 1.0f), 0.0f); *oy = pt_max(pt_min(y * p.exposure_mult, 1.0f), 0.0f); *oz = pt_max(pt_min(z * p.exposure_mult, 1.0f), 0.0f);
        }
        return;
    }
    bad = !pt_isfinite(x) || !pt_isfinite(y) || !pt_isfinite(z);
    if (p.luminance_only) {  // Reinhard0 reinhard0.rs:84-103 / Reinhard1 reinhard1.rs:88-110
        float l = p.key_value * y / p.lw[1];
        float sf = (p.tonemap == PT_TONEMAP_REINHARD0) ? l / (1.0f + l) : l * (p.inv_white2 * l + 1.0f) / (1.0f + l);
        if (bad) { x = MX; y = MY; z = MZ; }
        *ox = sf * x; *oy = sf * y; *oz = sf * z;
        return;
    }
    // x3 variants: per channel (reinhard0.rs:179-196, reinhard1.rs:203-231)
    float c[3] = {x, y, z}, o[3];
    for (int k = 0; k < 3; ++k) {
        float l = p.key_value * c[k] / p.lw[k];
        float sf = (p.tonemap == PT_TONEMAP_REINHARD0) ? l / (1.0f + l) : l * (p.inv_white2 * l + 1.0f) / (1.0f + l);
        o[k] = sf * c[k];
    }
    if (p.tonemap == PT_TONEMAP_REINHARD0) { if (bad) { float m[3] = {MX, MY, MZ}; for (int k = 0; k < 3; ++k) { float l = p.key_value * c[k] / p.lw[k]; o[k] = (l / (1.0f + l)) * m[k]; } } }
    else if (!pt_isfinite(o[0]) || !pt_isfinite(o[1]) || !pt_isfinite(o[2])) { o[0] = MX; o[1] = MY; o[2] = MZ; }
    *ox = o[0]; *oy = o[1]; *oz = o[2];
}

__global__ void __launch_bounds__(kBlock) k_output(const float4* __restrict__ film, OutParams p, uchar4* __restrict__ rgba8, float* __restrict__ linear_rgb) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += gridDim.x * blockDim.x) {
        float4 c = film[i];
        float tx, ty, tz, r, g, b;
        tonemap_pixel(p, c.x, c.y, c.z, &tx, &ty, &tz);
        xyz_to_rgb(p.colorspace, tx, ty, tz, &r, &g, &b);
        rgba8[i] = make_uchar4(quantize(oetf(p.colorspace, r)), quantize(oetf(p.colorspace, g)), quantize(oetf(p.colorspace, b)), 255);
        if (linear_rgb) {
            xyz_to_rgb(p.colorspace, p.factor * c.x, p.factor * c.y, p.factor * c.z, &r, &g, &b);
            linear_rgb[3 * (size_t)i] = r; linear_rgb[3 * (size_t)i + 1] = g; linear_rgb[3 * (size_t)i + 2] = b;
        }
    }
}

pt_status ofail(pt_status st, const std::string& m) { pt_set_error(m); return st; }

// ---- file writers (host) -----------------------------------------------------------------------------------------
uint32_t crc32_update(uint32_t crc, const uint8_t* d, size_t n) {
    static uint32_t table[256]; static bool init = false;
    if (!init) { for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; } init = true; }
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ d[i]) & 0xff] ^ (crc >> 8);
    return crc;
}
void put_be32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void png_chunk(FILE* f, const char* type, const std::vector<uint8_t>& data) {
    std::vector<uint8_t> c; put_be32(c, (uint32_t)data.size());
    c.insert(c.end(), type, type + 4); c.insert(c.end(), data.begin(), data.end());
    uint32_t crc = crc32_update(0xffffffffu, c.data() + 4, c.size() - 4) ^ 0xffffffffu;
    put_be32(c, crc);
    fwrite(c.data(), 1, c.size(), f);
}
struct Chroma { float wx, wy, rx, ry, gx, gy, bx, by, gamma; };
Chroma chroma_of(int cs) {  // REC709 / REC2020 primaries and effective_gamma, src/tonemap/mod.rs:74-92,160-204
    if (cs == PT_COLORSPACE_REC2020) return Chroma{0.3127f, 0.3290f, 0.708f, 0.292f, 0.292f, 0.170f, 0.131f, 0.046f, 1.0f / 2.4f};
    Chroma c{0.3127f, 0.3290f, 0.64f, 0.33f, 0.30f, 0.60f, 0.15f, 0.06f, cs == PT_COLORSPACE_SRGB ? 1.0f / 2.2f : 1.0f / 1.95f};
    return c;
}

}  // namespace

extern "C" {

pt_status pt_output_film(const pt_output_desc* d, const float* film, uint8_t* rgba8, float* linear_rgb) {
    if (!d || !film || !rgba8) return ofail(PT_ERR_INVALID_ARGUMENT, "null argument");
    if (d->width == 0 || d->height == 0 || !(d->factor > 0.0f)) return ofail(PT_ERR_INVALID_ARGUMENT, "bad size or factor");
    if (d->tonemap < 0 || d->tonemap > 2 || d->colorspace < 0 || d->colorspace > 2) return ofail(PT_ERR_INVALID_ARGUMENT, "bad tonemap / colorspace");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return ofail(PT_ERR_NO_DEVICE, "no HIP device available: the product path has no CPU fallback");
    const uint32_t n = d->width * d->height;
    float4* d_film = nullptr; uchar4* d_rgba = nullptr; float* d_lin = nullptr; double* d_part = nullptr;
    const int grid = 1024;
    hipError_t e = hipMalloc(&d_film, sizeof(float4) * n);
    if (e == hipSuccess) e = hipMalloc(&d_rgba, sizeof(uchar4) * n);
    if (e == hipSuccess && linear_rgb) e = hipMalloc(&d_lin, sizeof(float) * 3 * (size_t)n);
    if (e == hipSuccess) e = hipMalloc(&d_part, sizeof(double) * 3 * grid);
    if (e == hipSuccess) e = hipMemcpy(d_film, film, sizeof(float4) * n, hipMemcpyHostToDevice);
    OutParams p; memset(&p, 0, sizeof(p));
    p.n = n; p.tonemap = d->tonemap; p.luminance_only = d->luminance_only; p.colorspace = d->colorspace;
    p.exposure_mult = pt_pow(2.0f, d->exposure); p.key_value = d->key_value; p.inv_white2 = 1.0f / (d->white_point * d->white_point); p.factor = d->factor;
    if (e == hipSuccess && d->tonemap != PT_TONEMAP_CLAMP) {
        int x3 = d->luminance_only ? (d->tonemap == PT_TONEMAP_REINHARD1 ? 2 : 0) : 1;
        hipLaunchKernelGGL(k_log_sums, dim3(grid), dim3(kBlock), 0, 0, d_film, n, x3, d_part);
        std::vector<double> part(3 * grid);
        e = hipMemcpy(part.data(), d_part, sizeof(double) * 3 * grid, hipMemcpyDeviceToHost);
        double s[3] = {0, 0, 0};
        for (int b = 0; b < grid; ++b) for (int k = 0; k < 3; ++k) s[k] += part[3 * b + k];
        // l_w = exp(sum_of_log / total_pixels) / factor (reinhard0.rs:66, reinhard1.rs:70; x3: per channel in f32)
        for (int k = 0; k < 3; ++k) {
            if (x3 == 1) p.lw[k] = pt_exp((float)s[k] / (float)n) / d->factor;
            else p.lw[k] = (float)pt_exp64(s[1] / (double)n) / d->factor;
        }
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_output, dim3(grid), dim3(kBlock), 0, 0, d_film, p, d_rgba, d_lin);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(rgba8, d_rgba, sizeof(uchar4) * n, hipMemcpyDeviceToHost);
    if (e == hipSuccess && linear_rgb) e = hipMemcpy(linear_rgb, d_lin, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost);
    hipFree(d_film); hipFree(d_rgba); hipFree(d_lin); hipFree(d_part);
    if (e != hipSuccess) return ofail(PT_ERR_DEVICE, hipGetErrorString(e));
    return PT_OK;
}

pt_status pt_write_png(const char* path, uint32_t w, uint32_t h, const uint8_t* rgba8, int32_t colorspace) {
    if (!path || !rgba8 || w == 0 || h == 0) return ofail(PT_ERR_INVALID_ARGUMENT, "bad argument");
    FILE* f = fopen(path, "wb");
    if (!f) return ofail(PT_ERR_INVALID_ARGUMENT, std::string("cannot open ") + path);
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    fwrite(sig, 1, 8, f);
    std::vector<uint8_t> ihdr; put_be32(ihdr, w); put_be32(ihdr, h); ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    png_chunk(f, "IHDR", ihdr);
    Chroma c = chroma_of(colorspace);
    std::vector<uint8_t> gama; put_be32(gama, (uint32_t)(c.gamma * 100000.0f + 0.5f)); png_chunk(f, "gAMA", gama);
    std::vector<uint8_t> chrm; for (float v : {c.wx, c.wy, c.rx, c.ry, c.gx, c.gy, c.bx, c.by}) put_be32(chrm, (uint32_t)(v * 100000.0f + 0.5f));
    png_chunk(f, "cHRM", chrm);
    // zlib stream of stored deflate blocks (no compression library needed)
    std::vector<uint8_t> raw; raw.reserve((size_t)h * (4 * (size_t)w + 1));
    for (uint32_t y = 0; y < h; ++y) { raw.push_back(0); raw.insert(raw.end(), rgba8 + 4 * (size_t)w * y, rgba8 + 4 * (size_t)w * (y + 1)); }
    std::vector<uint8_t> z; z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw.size();) {
        size_t len = raw.size() - off; if (len > 65535) len = 65535;
        z.push_back(off + len == raw.size() ? 1 : 0);
        z.push_back(len & 0xff); z.push_back(len >> 8); z.push_back(~len & 0xff); z.push_back((~len >> 8) & 0xff);
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + len);
        for (size_t i = 0; i < len; ++i) { a = (a + raw[off + i]) % 65521u; b = (b + a) % 65521u; }
        off += len;
    }
    put_be32(z, (b << 16) | a);
    png_chunk(f, "IDAT", z);
    png_chunk(f, "IEND", {});
    fclose(f);
    return PT_OK;
}

pt_status pt_write_exr(const char* path, uint32_t w, uint32_t h, const float* rgb, int32_t colorspace) {
    if (!path || !rgb || w == 0 || h == 0) return ofail(PT_ERR_INVALID_ARGUMENT, "bad argument");
    FILE* f = fopen(path, "wb");
    if (!f) return ofail(PT_ERR_INVALID_ARGUMENT, std::string("cannot open ") + path);
    std::vector<uint8_t> hd;
    auto put32 = [&](std::vector<uint8_t>& v, uint32_t x) { for (int k = 0; k < 4; ++k) v.push_back((x >> (8 * k)) & 0xff); };
    auto putf = [&](std::vector<uint8_t>& v, float x) { uint32_t u; memcpy(&u, &x, 4); put32(v, u); };
    auto puts0 = [&](std::vector<uint8_t>& v, const char* s) { v.insert(v.end(), s, s + strlen(s) + 1); };
    auto attr = [&](const char* name, const char* type, const std::vector<uint8_t>& val) { puts0(hd, name); puts0(hd, type); put32(hd, (uint32_t)val.size()); hd.insert(hd.end(), val.begin(), val.end()); };
    put32(hd, 20000630u); put32(hd, 2u);
    std::vector<uint8_t> ch;
    for (const char* n : {"B", "G", "R"}) { puts0(ch, n); put32(ch, 2u /* FLOAT */); ch.push_back(0); ch.push_back(0); ch.push_back(0); ch.push_back(0); put32(ch, 1); put32(ch, 1); }
    ch.push_back(0);
    attr("channels", "chlist", ch);
    Chroma c = chroma_of(colorspace);
    std::vector<uint8_t> cv; for (float v : {c.rx, c.ry, c.gx, c.gy, c.bx, c.by, c.wx, c.wy}) putf(cv, v);
    attr("chromaticities", "chromaticities", cv);
    attr("compression", "compression", {0});
    std::vector<uint8_t> box; put32(box, 0); put32(box, 0); put32(box, w - 1); put32(box, h - 1);
    attr("dataWindow", "box2i", box); attr("displayWindow", "box2i", box);
    attr("lineOrder", "lineOrder", {0});
    std::vector<uint8_t> one; putf(one, 1.0f); attr("pixelAspectRatio", "float", one);
    std::vector<uint8_t> v2; putf(v2, 0.0f); putf(v2, 0.0f); attr("screenWindowCenter", "v2f", v2);
    attr("screenWindowWidth", "float", one);
    hd.push_back(0);
    fwrite(hd.data(), 1, hd.size(), f);
    uint64_t line_bytes = 8 + 12ull * w, table0 = hd.size() + 8ull * h;
    for (uint32_t y = 0; y < h; ++y) { uint64_t off = table0 + line_bytes * y; fwrite(&off, 8, 1, f); }
    std::vector<float> row(3 * (size_t)w);
    for (uint32_t y = 0; y < h; ++y) {
        int32_t yy = (int32_t)y, sz = (int32_t)(12 * w);
        fwrite(&yy, 4, 1, f); fwrite(&sz, 4, 1, f);
        for (uint32_t x = 0; x < w; ++x) { const float* p = rgb + 3 * ((size_t)y * w + x); row[x] = p[2]; row[w + x] = p[1]; row[2 * (size_t)w + x] = p[0]; }
        fwrite(row.data(), 4, row.size(), f);
    }
    fclose(f);
    return PT_OK;
}

}  // extern "C"
