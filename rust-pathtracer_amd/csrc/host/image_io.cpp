// image_io.cpp — see image_io.h.  Decoders are written against the file format specifications (PNG 1.2 / RFC 1950-1951
// through zlib, BMP BITMAPINFOHEADER, Radiance RGBE with new-style run-length scanlines, OpenEXR 2 single-part files).
#include "image_io.h"

#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>

namespace pth {
namespace {

// Asset files are untrusted input: every size read from a header is checked against the file before it is used as an offset, and
// no image may exceed this many pixels (2^28: a 16384 x 16384 map; 4 GB of f32 RGBA).
constexpr uint64_t kMaxPixels = 1ull << 28;
bool sane_dimensions(uint64_t w, uint64_t h) { return w > 0 && h > 0 && w <= (1u << 20) && h <= (1u << 20) && w * h <= kMaxPixels; }

bool read_file(const std::string& path, std::vector<uint8_t>* out, std::string* error) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { *error = "could not find file at " + path; return false; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out->resize(n > 0 ? (size_t)n : 0);
    size_t got = n > 0 ? fread(out->data(), 1, (size_t)n, f) : 0;
    fclose(f);
    if (got != out->size()) { *error = "short read on " + path; return false; }
    return true;
}
uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
uint32_t le32(const uint8_t* p) { return (uint32_t)p[3] << 24 | (uint32_t)p[2] << 16 | (uint32_t)p[1] << 8 | p[0]; }
uint16_t le16(const uint8_t* p) { return (uint16_t)(p[1] << 8 | p[0]); }
uint64_t le64(const uint8_t* p) { return (uint64_t)le32(p + 4) << 32 | le32(p); }

bool inflate_all(const uint8_t* src, size_t n, std::vector<uint8_t>* dst, size_t expected, std::string* error) {
    if (expected > (1ull << 33)) { *error = "compressed block claims an absurd size"; return false; }
    dst->resize(expected);
    uLongf len = (uLongf)expected;
    int rc = uncompress(dst->data(), &len, src, (uLong)n);
    if (rc != Z_OK) { *error = "zlib inflate failed (" + std::to_string(rc) + ")"; return false; }
    dst->resize(len);
    return true;
}

// ---- 8-bit RGBA raster shared by PNG and BMP -------------------------------------------------------------------------
struct Raster8 { uint32_t w = 0, h = 0; bool grey = false; std::vector<uint8_t> rgba; };

int paeth(int a, int b, int c) { int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }

bool decode_png(const std::vector<uint8_t>& d, Raster8* out, std::string* error) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (d.size() < 33 || memcmp(d.data(), sig, 8) != 0) { *error = "not a PNG file"; return false; }
    uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    size_t p = 8;
    while (p + 12 <= d.size()) {
        uint32_t len = be32(&d[p]);
        const uint8_t* type = &d[p + 4];
        const uint8_t* body = &d[p + 8];
        if ((uint64_t)p + 12 + len > d.size()) { *error = "truncated PNG chunk"; return false; }
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) { *error = "PNG IHDR chunk is too short"; return false; }
            w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
        }
        else if (!memcmp(type, "PLTE", 4)) palette.assign(body, body + len);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!memcmp(type, "IEND", 4)) break;
        p += 12 + len;
    }
    if (w == 0 || h == 0) { *error = "PNG without IHDR"; return false; }
    if (!sane_dimensions(w, h)) { *error = "PNG dimensions out of range"; return false; }
    if (interlace != 0) { *error = "interlaced PNG is not supported"; return false; }
    int samples = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!samples || !(depth == 8 || depth == 16 || (depth < 8 && (ctype == 0 || ctype == 3)))) { *error = "unsupported PNG colour type / depth"; return false; }
    size_t bpp_bits = (size_t)samples * depth, stride = ((size_t)w * bpp_bits + 7) / 8, bpp = bpp_bits < 8 ? 1 : bpp_bits / 8;
    std::vector<uint8_t> raw;
    if (!inflate_all(idat.data(), idat.size(), &raw, (stride + 1) * h, error)) return false;
    if (raw.size() != (stride + 1) * h) { *error = "PNG data has the wrong size"; return false; }
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* src = &raw[(stride + 1) * y];
        uint8_t* cur = &img[stride * y];
        const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
        int ft = src[0];
        for (size_t x = 0; x < stride; ++x) {
            int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0, v = src[1 + x];
            switch (ft) { case 0: break; case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) / 2; break; case 4: v += paeth(a, b, c); break;
                          default: *error = "bad PNG filter"; return false; }
            cur[x] = (uint8_t)v;
        }
    }
    out->w = w; out->h = h; out->grey = (ctype == 0 || ctype == 4);
    out->rgba.resize((size_t)w * h * 4);
    auto sample = [&](const uint8_t* row, size_t index) -> uint32_t {  // sample `index` of the row, scaled to 8 bits
        if (depth == 8) return row[index];
        if (depth == 16) { uint32_t v = (uint32_t)row[2 * index] << 8 | row[2 * index + 1]; return (v * 255u + 32767u) / 65535u; }
        uint32_t per = 8 / depth, byte = row[index / per], shift = (per - 1 - index % per) * depth, v = (byte >> shift) & ((1u << depth) - 1);
        return ctype == 3 ? v : v * 255u / ((1u << depth) - 1);
    };
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* row = &img[stride * y];
        for (uint32_t x = 0; x < w; ++x) {
            uint8_t* o = &out->rgba[((size_t)y * w + x) * 4];
            if (ctype == 0) { uint8_t g = (uint8_t)sample(row, x); o[0] = o[1] = o[2] = g; o[3] = 255; }
            else if (ctype == 2) { for (int k = 0; k < 3; ++k) o[k] = (uint8_t)sample(row, 3 * x + k); o[3] = 255; }
            else if (ctype == 3) {
                uint32_t i = sample(row, x);
                if (3 * i + 2 >= palette.size()) { *error = "PNG palette index out of range"; return false; }
                o[0] = palette[3 * i]; o[1] = palette[3 * i + 1]; o[2] = palette[3 * i + 2]; o[3] = i < trns.size() ? trns[i] : 255;
            }
            else if (ctype == 4) { uint8_t g = (uint8_t)sample(row, 2 * x); o[0] = o[1] = o[2] = g; o[3] = (uint8_t)sample(row, 2 * x + 1); }
            else { for (int k = 0; k < 4; ++k) o[k] = (uint8_t)sample(row, 4 * x + k); }
        }
    }
    return true;
}

bool decode_bmp(const std::vector<uint8_t>& d, Raster8* out, std::string* error) {
    if (d.size() < 54 || d[0] != 'B' || d[1] != 'M') { *error = "not a BMP file"; return false; }
    uint32_t offset = le32(&d[10]), hdr = le32(&d[14]);
    int32_t w = (int32_t)le32(&d[18]), h = (int32_t)le32(&d[22]);
    uint16_t bits = le16(&d[28]);
    uint32_t compression = le32(&d[30]), colors = hdr >= 40 ? le32(&d[46]) : 0;
    if (hdr < 40 || w <= 0 || h == 0 || h == INT32_MIN || !(compression == 0 || (compression == 3 && bits == 32)) || !(bits == 8 || bits == 24 || bits == 32)) {
        *error = "unsupported BMP variant"; return false;
    }
    bool top_down = h < 0;
    uint32_t H = (uint32_t)(h < 0 ? -h : h), W = (uint32_t)w;
    if (!sane_dimensions(W, H)) { *error = "BMP dimensions out of range"; return false; }
    size_t stride = ((size_t)W * bits + 31) / 32 * 4;
    if ((uint64_t)offset + (uint64_t)stride * H > d.size()) { *error = "truncated BMP"; return false; }
    if (bits == 8 && colors == 0) colors = 256;
    if (colors > 256) { *error = "BMP palette too large"; return false; }
    if (bits == 8 && (uint64_t)14 + hdr + 4ull * colors > d.size()) { *error = "BMP palette runs past the end of the file"; return false; }
    const uint8_t* pal = bits == 8 ? &d[14 + hdr] : nullptr;
    out->w = W; out->h = H; out->grey = false;
    out->rgba.resize((size_t)W * H * 4);
    for (uint32_t y = 0; y < H; ++y) {
        const uint8_t* row = &d[offset + stride * (top_down ? y : H - 1 - y)];
        for (uint32_t x = 0; x < W; ++x) {
            uint8_t* o = &out->rgba[((size_t)y * W + x) * 4];
            if (bits == 8) { uint32_t i = row[x]; if (i >= colors) i = 0; o[0] = pal[4 * i + 2]; o[1] = pal[4 * i + 1]; o[2] = pal[4 * i]; o[3] = 255; }
            else if (bits == 24) { o[0] = row[3 * x + 2]; o[1] = row[3 * x + 1]; o[2] = row[3 * x]; o[3] = 255; }
            else { o[0] = row[4 * x + 2]; o[1] = row[4 * x + 1]; o[2] = row[4 * x]; o[3] = compression == 3 ? row[4 * x + 3] : 255; }
        }
    }
    return true;
}

bool read_raster8(const std::string& path, Raster8* r, std::string* error) {
    std::vector<uint8_t> d;
    if (!read_file(path, &d, error)) return false;
    if (d.size() >= 2 && d[0] == 'B' && d[1] == 'M') return decode_bmp(d, r, error);
    return decode_png(d, r, error);
}

float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31, m = h & 1023, bits;
    if (e == 0) {
        if (m == 0) bits = sign;
        else { int shift = 0; while (!(m & 1024)) { m <<= 1; ++shift; } bits = sign | (uint32_t)(127 - 15 - shift + 1) << 23 | (m & 1023) << 13; }
    } else if (e == 31) bits = sign | 0x7f800000u | m << 13;
    else bits = sign | (e + 112) << 23 | m << 13;
    float f; memcpy(&f, &bits, 4); return f;
}

}  // namespace

bool read_grey8(const std::string& path, Image* out, std::string* error) {
    Raster8 r;
    if (!read_raster8(path, &r, error)) return false;
    out->width = r.w; out->height = r.h; out->channels = 1;
    out->data.resize((size_t)r.w * r.h);
    for (size_t i = 0; i < out->data.size(); ++i) {
        const uint8_t* p = &r.rgba[4 * i];
        // image::DynamicImage::into_luma8: integer Rec.709 weights (2126, 7152, 722) / 10000; greyscale sources pass through
        uint32_t l = r.grey ? p[0] : (2126u * p[0] + 7152u * p[1] + 722u * p[2]) / 10000u;
        out->data[i] = (float)l / 255.0f;
    }
    return true;
}

bool read_rgba8(const std::string& path, Image* out, std::string* error) {
    Raster8 r;
    if (!read_raster8(path, &r, error)) return false;
    out->width = r.w; out->height = r.h; out->channels = 4;
    out->data.resize((size_t)r.w * r.h * 4);
    for (size_t i = 0; i < out->data.size(); ++i) out->data[i] = (float)r.rgba[i] / 255.0f;
    return true;
}

bool read_hdr(const std::string& path, float alpha_fill, Image* out, std::string* error) {
    std::vector<uint8_t> d;
    if (!read_file(path, &d, error)) return false;
    size_t p = 0;
    auto line = [&](std::string* s) { s->clear(); while (p < d.size() && d[p] != '\n') s->push_back((char)d[p++]); if (p < d.size()) ++p; return p <= d.size(); };
    std::string l;
    line(&l);
    if (l.compare(0, 2, "#?") != 0) { *error = "not a Radiance HDR file"; return false; }
    for (;;) { if (p >= d.size()) { *error = "truncated HDR header"; return false; } line(&l); if (l.empty()) break; }
    line(&l);
    int h = 0, w = 0;
    if (sscanf(l.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) { *error = "unsupported HDR orientation: " + l; return false; }
    if (!sane_dimensions((uint64_t)w, (uint64_t)h)) { *error = "HDR dimensions out of range"; return false; }
    out->width = (uint32_t)w; out->height = (uint32_t)h; out->channels = 4;
    out->data.resize((size_t)w * h * 4);
    std::vector<uint8_t> scan((size_t)w * 4);
    for (int y = 0; y < h; ++y) {
        if (p + 4 > d.size()) { *error = "truncated HDR data"; return false; }
        if (w >= 8 && w < 32768 && d[p] == 2 && d[p + 1] == 2 && (d[p + 2] << 8 | d[p + 3]) == w) {
            p += 4;
            for (int c = 0; c < 4; ++c) {
                int x = 0;
                while (x < w) {
                    if (p >= d.size()) { *error = "truncated HDR scanline"; return false; }
                    int n = d[p++];
                    if (n > 128) { n -= 128; if (x + n > w || p >= d.size()) { *error = "bad HDR run"; return false; } uint8_t v = d[p++]; while (n--) scan[4 * (x++) + c] = v; }
                    else { if (n == 0 || x + n > w || p + n > d.size()) { *error = "bad HDR run"; return false; } while (n--) scan[4 * (x++) + c] = d[p++]; }
                }
            }
        } else {
            if (p + (size_t)w * 4 > d.size()) { *error = "truncated HDR data"; return false; }
            memcpy(scan.data(), &d[p], (size_t)w * 4); p += (size_t)w * 4;  // flat (old-style runs are not produced by current writers)
        }
        for (int x = 0; x < w; ++x) {
            const uint8_t* q = &scan[4 * x];
            float* o = &out->data[((size_t)y * w + x) * 4];
            // image::codecs::hdr::Rgbe8Pixel::to_hdr: c * 2^(e - 128 - 8), e == 0 -> 0
            float scale = q[3] == 0 ? 0.0f : std::ldexp(1.0f, (int)q[3] - 136);
            o[0] = (float)q[0] * scale; o[1] = (float)q[1] * scale; o[2] = (float)q[2] * scale; o[3] = alpha_fill;
        }
    }
    return true;
}

// ---- PIZ (OpenEXR compression 4): what the reference's `exr` crate decodes for any Poly-Haven-style HDRI (src/parsing/texture.rs:75-89).
// A block holds, per channel and 16-bit half of its samples, a plane of u16 values that were (1) mapped through a table of the values
// that occur (a bitmap of 65536 bits lists them), (2) transformed by a two-dimensional Haar-like wavelet (the 14-bit form when the
// largest table index is below 2^14, the modulo-2^16 form otherwise) and (3) Huffman-coded with canonical codes of up to 58 bits and a
// run-length symbol.  Restated from the published format (OpenEXR's ImfPizCompressor / ImfHuf / ImfWav); every length and index that comes
// from the file is checked before use.
namespace piz {
const int kEncBits = 16, kEncSize = (1 << kEncBits) + 1;   // 65536 symbols + the run-length symbol
struct BitReader {
    const uint8_t* p; const uint8_t* end; uint64_t c = 0; int lc = 0;
    bool get(int n, uint32_t* v) {   // n <= 32, most significant bit first
        while (lc < n) { if (p >= end) return false; c = (c << 8) | *p++; lc += 8; }
        lc -= n; *v = (uint32_t)((c >> lc) & ((1ull << n) - 1ull)); return true;
    }
};
// hufUncompress: 20-byte header (first and last symbol, table length, number of data bits, reserved), packed code lengths, data
bool huf_uncompress(const uint8_t* in, size_t n_in, uint16_t* out, size_t n_out, std::string* error) {
    if (n_in == 0) { if (n_out != 0) { *error = "EXR PIZ block without data"; return false; } return true; }
    if (n_in < 20) { *error = "truncated EXR PIZ Huffman header"; return false; }
    const uint32_t im = le32(in), iM = le32(in + 4), n_bits = le32(in + 12);
    if (im >= (uint32_t)kEncSize || iM >= (uint32_t)kEncSize || im > iM) { *error = "bad EXR PIZ symbol range"; return false; }
    // code lengths, 6 bits each; 59..62 = a run of 2..5 zero lengths, 63 = a run of 6 + (next 8 bits)
    std::vector<uint8_t> len((size_t)kEncSize, 0);
    BitReader br{in + 20, in + n_in};
    for (uint32_t sym = im; sym <= iM; ++sym) {
        uint32_t l;
        if (!br.get(6, &l)) { *error = "truncated EXR PIZ code table"; return false; }
        if (l >= 59) {
            uint32_t run = l - 59 + 2;
            if (l == 63) { uint32_t more; if (!br.get(8, &more)) { *error = "truncated EXR PIZ code table"; return false; } run = more + 6; }
            if (sym + run > iM + 1) { *error = "bad EXR PIZ code table"; return false; }
            sym += run - 1;   // (the lengths stay 0)
        } else len[sym] = (uint8_t)l;
    }
    const uint8_t* data = br.p;   // the table ends on a byte boundary
    if ((uint64_t)n_bits > 8ull * (uint64_t)(in + n_in - data)) { *error = "bad EXR PIZ bit count"; return false; }
    // canonical codes (hufCanonicalCodeTable): within a length, codes rise with the symbol; first[l] = the length's first code
    uint64_t count[59] = {0}, first[59] = {0};
    for (uint32_t sym = im; sym <= iM; ++sym) count[len[sym]]++;
    { uint64_t c = 0; for (int l = 58; l > 0; --l) { const uint64_t nc = (c + count[l]) >> 1; first[l] = c; c = nc; } }
    std::vector<uint32_t> offset(60, 0), sorted;
    for (int l = 1; l <= 58; ++l) offset[l + 1] = offset[l] + (uint32_t)count[l];
    sorted.resize(offset[59]);
    { std::vector<uint32_t> at(offset.begin(), offset.end()); for (uint32_t sym = im; sym <= iM; ++sym) if (len[sym]) sorted[at[len[sym]]++] = sym; }
    // decode bit by bit (hufDecode's result without its 14-bit table): a code of length l is a symbol iff code - first[l] < count[l]
    const uint32_t rlc = iM;
    size_t produced = 0; uint64_t left = n_bits;
    BitReader dr{data, in + n_in};
    while (left > 0) {
        uint64_t code = 0; int l = 0; uint32_t sym = 0; bool found = false;
        while (l < 58 && left > 0) {
            uint32_t bit;
            if (!dr.get(1, &bit)) { *error = "truncated EXR PIZ data"; return false; }
            code = (code << 1) | bit; ++l; --left;
            if (count[l] != 0 && code >= first[l] && code - first[l] < count[l]) { sym = sorted[offset[l] + (uint32_t)(code - first[l])]; found = true; break; }
        }
        if (!found) { *error = "bad EXR PIZ code"; return false; }   // (the header's bit count is exact: no padding inside it)
        if (sym == rlc) {
            uint32_t run;
            if (left < 8 || !dr.get(8, &run)) { *error = "truncated EXR PIZ run"; return false; }
            left -= 8;
            if (produced == 0 || produced + run > n_out) { *error = "bad EXR PIZ run"; return false; }
            for (uint32_t k = 0; k < run; ++k) out[produced + k] = out[produced - 1];
            produced += run;
        } else {
            if (produced >= n_out) { *error = "EXR PIZ block is longer than its tile"; return false; }
            out[produced++] = (uint16_t)sym;
        }
    }
    if (produced != n_out) { *error = "EXR PIZ block has the wrong size"; return false; }
    return true;
}
inline void wdec14(uint16_t l, uint16_t h, uint16_t* a, uint16_t* b) {
    const int ls = (int16_t)l, hs = (int16_t)h;
    const int ai = ls + (hs & 1) + (hs >> 1);
    *a = (uint16_t)(int16_t)ai; *b = (uint16_t)(int16_t)(ai - hs);
}
inline void wdec16(uint16_t l, uint16_t h, uint16_t* a, uint16_t* b) {
    const int m = l, d = h;
    const int bb = (m - (d >> 1)) & 0xffff, aa = (d + bb - 0x8000) & 0xffff;
    *b = (uint16_t)bb; *a = (uint16_t)aa;
}
// wav2Decode on nx x ny values, ox / oy apart (in u16 units), mx = the largest value the encoder's table produced
void wav2_decode(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t mx) {
    const bool w14 = mx < (1 << 14);
    const int n = nx > ny ? ny : nx;
    int p = 1, p2;
    while (p <= n) p <<= 1;
    p >>= 1; p2 = p; p >>= 1;
    while (p >= 1) {
        uint16_t* py = in;
        uint16_t* const ey = in + (ptrdiff_t)oy * (ny - p2);
        const ptrdiff_t oy1 = (ptrdiff_t)oy * p, oy2 = (ptrdiff_t)oy * p2, ox1 = (ptrdiff_t)ox * p, ox2 = (ptrdiff_t)ox * p2;
        uint16_t i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            uint16_t* px = py;
            uint16_t* const ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t* p01 = px + ox1; uint16_t* p10 = px + oy1; uint16_t* p11 = p10 + ox1;
                if (w14) { wdec14(*px, *p10, &i00, &i10); wdec14(*p01, *p11, &i01, &i11); wdec14(i00, i01, px, p01); wdec14(i10, i11, p10, p11); }
                else { wdec16(*px, *p10, &i00, &i10); wdec16(*p01, *p11, &i01, &i11); wdec16(i00, i01, px, p01); wdec16(i10, i11, p10, p11); }
            }
            if (nx & p) {
                uint16_t* p10 = px + oy1;
                if (w14) wdec14(*px, *p10, &i00, p10); else wdec16(*px, *p10, &i00, p10);
                *px = i00;
            }
        }
        if (ny & p) {
            uint16_t* px = py;
            uint16_t* const ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t* p01 = px + ox1;
                if (w14) wdec14(*px, *p01, &i00, p01); else wdec16(*px, *p01, &i00, p01);
                *px = i00;
            }
        }
        p2 = p; p >>= 1;
    }
}
// One block of bw x bh pixels with the given channel types (1 = half: one u16 per sample, else two) -> the block's scanline-ordered bytes
bool decode_block(const uint8_t* src, size_t size, const std::vector<int>& types, int bw, int bh, std::vector<uint8_t>* raw, std::string* error) {
    if (size < 4) { *error = "truncated EXR PIZ block"; return false; }
    const uint32_t min_nz = le16(src), max_nz = le16(src + 2);
    const uint32_t kBitmap = 8192;
    if (max_nz >= kBitmap) { *error = "bad EXR PIZ bitmap range"; return false; }
    std::vector<uint8_t> bitmap(kBitmap, 0);
    size_t q = 4;
    if (min_nz <= max_nz) {
        const size_t n = (size_t)max_nz - min_nz + 1;
        if (q + n > size) { *error = "truncated EXR PIZ bitmap"; return false; }
        memcpy(&bitmap[min_nz], src + q, n); q += n;
    }
    std::vector<uint16_t> lut(65536, 0);
    uint32_t k = 0;
    for (uint32_t i = 0; i < 65536; ++i) if (i == 0 || (bitmap[i >> 3] & (1u << (i & 7)))) lut[k++] = (uint16_t)i;
    const uint16_t max_value = (uint16_t)(k - 1);
    if (q + 4 > size) { *error = "truncated EXR PIZ block"; return false; }
    const uint32_t length = le32(src + q); q += 4;
    if ((uint64_t)q + length > size) { *error = "bad EXR PIZ data length"; return false; }
    size_t total = 0;
    for (int t : types) total += (size_t)bw * bh * (t == 1 ? 1 : 2);
    std::vector<uint16_t> buf(total);
    if (!huf_uncompress(src + q, length, buf.data(), total, error)) return false;
    size_t start = 0;
    std::vector<size_t> starts;
    for (int t : types) {
        const int sz = t == 1 ? 1 : 2;
        starts.push_back(start);
        for (int j = 0; j < sz; ++j) wav2_decode(buf.data() + start + j, bw, sz, bh, bw * sz, max_value);
        start += (size_t)bw * bh * sz;
    }
    for (uint16_t& v : buf) v = lut[v];
    raw->resize(total * 2);
    size_t o = 0;
    for (int y = 0; y < bh; ++y)
        for (size_t c = 0; c < types.size(); ++c) {
            const size_t n = (size_t)bw * (types[c] == 1 ? 1 : 2);
            const uint16_t* from = buf.data() + starts[c] + n * y;
            for (size_t i = 0; i < n; ++i) { (*raw)[o++] = (uint8_t)(from[i] & 0xff); (*raw)[o++] = (uint8_t)(from[i] >> 8); }   // little-endian, as the other codecs leave it
        }
    return true;
}
}  // namespace piz

bool read_exr(const std::string& path, Image* out, std::string* error) {
    std::vector<uint8_t> d;
    if (!read_file(path, &d, error)) return false;
    if (d.size() < 8 || le32(&d[0]) != 20000630u) { *error = "not an OpenEXR file"; return false; }
    uint32_t version = le32(&d[4]);
    bool tiled = (version & 0x200u) != 0;
    if (version & 0x1800u) { *error = "multi-part / deep OpenEXR files are not supported"; return false; }
    size_t p = 8;
    std::map<std::string, std::vector<uint8_t>> attr;
    for (;;) {
        if (p >= d.size()) { *error = "truncated EXR header"; return false; }
        if (d[p] == 0) { ++p; break; }
        std::string name, type;
        while (p < d.size() && d[p]) name.push_back((char)d[p++]);
        ++p;
        while (p < d.size() && d[p]) type.push_back((char)d[p++]);
        ++p;
        if (p + 4 > d.size()) { *error = "truncated EXR header"; return false; }
        uint32_t size = le32(&d[p]); p += 4;
        if ((uint64_t)p + size > d.size()) { *error = "truncated EXR header"; return false; }
        attr[name].assign(&d[p], &d[p] + size); p += size;
    }
    if (!attr.count("channels") || !attr.count("compression") || !attr.count("dataWindow")) { *error = "EXR header lacks required attributes"; return false; }
    struct Channel { std::string name; int type; int slot; };
    std::vector<Channel> channels;
    {
        const std::vector<uint8_t>& c = attr["channels"];
        size_t q = 0;
        while (q < c.size() && c[q]) {
            Channel ch;
            while (q < c.size() && c[q]) ch.name.push_back((char)c[q++]);
            ++q;
            if (q + 16 > c.size()) { *error = "truncated EXR channel list"; return false; }
            ch.type = (int)le32(&c[q]);
            if (ch.type < 0 || ch.type > 2) { *error = "unknown EXR pixel type"; return false; }
            uint32_t xs = le32(&c[q + 8]), ys = le32(&c[q + 12]);
            if (xs != 1 || ys != 1) { *error = "subsampled EXR channels are not supported"; return false; }
            q += 16;
            ch.slot = ch.name == "R" ? 0 : ch.name == "G" ? 1 : ch.name == "B" ? 2 : ch.name == "A" ? 3 : -1;
            channels.push_back(ch);
        }
    }
    if (attr["compression"].size() < 1 || attr["dataWindow"].size() < 16) { *error = "EXR header attribute is too short"; return false; }
    if (channels.empty()) { *error = "EXR file without channels"; return false; }
    int compression = attr["compression"][0];
    if (compression > 5) { *error = "EXR compression " + std::to_string(compression) + " is not supported (only none, RLE, ZIPS, ZIP, PIZ, PXR24)"; return false; }
    const uint8_t* dw = attr["dataWindow"].data();
    int x0 = (int)le32(dw), y0 = (int)le32(dw + 4), x1 = (int)le32(dw + 8), y1 = (int)le32(dw + 12);
    const int64_t W64 = (int64_t)x1 - x0 + 1, H64 = (int64_t)y1 - y0 + 1;
    if (W64 <= 0 || H64 <= 0) { *error = "empty EXR data window"; return false; }
    if (!sane_dimensions((uint64_t)W64, (uint64_t)H64)) { *error = "EXR data window out of range"; return false; }
    const int W = (int)W64, H = (int)H64;
    out->width = (uint32_t)W; out->height = (uint32_t)H; out->channels = 4;
    out->data.assign((size_t)W * H * 4, 0.0f);
    bool has_alpha = false;
    for (auto& c : channels) has_alpha = has_alpha || c.slot == 3;
    if (!has_alpha) for (size_t i = 0; i < (size_t)W * H; ++i) out->data[4 * i + 3] = 1.0f;
    size_t pixel_bytes = 0;
    for (auto& c : channels) pixel_bytes += c.type == 1 ? 2 : 4;
    int tile_w = 0, tile_h = 0;
    if (tiled) {
        if (!attr.count("tiles")) { *error = "tiled EXR without a tiles attribute"; return false; }
        if (attr["tiles"].size() < 9) { *error = "EXR tiles attribute is too short"; return false; }
        const uint8_t* t = attr["tiles"].data();
        tile_w = (int)le32(t); tile_h = (int)le32(t + 4);
        if (tile_w <= 0 || tile_h <= 0) { *error = "bad EXR tile size"; return false; }
        if ((t[8] & 0xf) != 0) { *error = "mip/rip-mapped EXR files are not supported"; return false; }
    }
    int block_lines = compression == 3 || compression == 5 ? 16 : compression == 4 ? 32 : 1;
    std::vector<int> channel_types;
    for (auto& c : channels) channel_types.push_back(c.type);
    size_t blocks = tiled ? (size_t)((W + tile_w - 1) / tile_w) * ((H + tile_h - 1) / tile_h) : (size_t)(H + block_lines - 1) / block_lines;
    if ((uint64_t)p + 8ull * blocks > d.size()) { *error = "truncated EXR offset table"; return false; }
    std::vector<uint8_t> raw, tmp;
    for (size_t b = 0; b < blocks; ++b) {
        const uint64_t off64 = le64(&d[p + 8 * b]);
        if (off64 > d.size()) { *error = "bad EXR block offset"; return false; }
        const size_t off = (size_t)off64;
        int bx, by, bw, bh; uint32_t size; const uint8_t* src;
        if (tiled) {
            if (off + 20 > d.size()) { *error = "bad EXR tile offset"; return false; }
            const int64_t tx = (int32_t)le32(&d[off]), ty = (int32_t)le32(&d[off + 4]);
            size = le32(&d[off + 16]); src = &d[off + 20];
            if (tx < 0 || ty < 0 || tx * tile_w >= W || ty * tile_h >= H) { *error = "EXR tile coordinates outside the data window"; return false; }
            bx = (int)(tx * tile_w); by = (int)(ty * tile_h); bw = std::min(tile_w, W - bx); bh = std::min(tile_h, H - by);
        } else {
            if (off + 8 > d.size()) { *error = "bad EXR scanline offset"; return false; }
            const int64_t line = (int64_t)(int32_t)le32(&d[off]) - y0;
            size = le32(&d[off + 4]); src = &d[off + 8];
            if (line < 0 || line >= H) { *error = "EXR scanline outside the data window"; return false; }
            by = (int)line; bx = 0; bw = W; bh = std::min(block_lines, H - by);
        }
        if ((uint64_t)(src - d.data()) + size > d.size() || bw <= 0 || bh <= 0 || by < 0 || bx < 0 || bx + bw > W || by + bh > H) { *error = "bad EXR block"; return false; }
        size_t expect = pixel_bytes * (size_t)bw * bh;
        const uint8_t* data = src;
        if (compression == 5 && size < expect) {
            // PXR24 (ImfPxr24Compressor): zlib over, per scanline and channel, byte planes of the differences between consecutive samples —
            // two planes for a half, four for an unsigned int, three for a float cut to its upper 24 bits (the one lossy step, the writer's)
            size_t planes = 0;
            for (int t : channel_types) planes += t == 1 ? 2 : t == 0 ? 4 : 3;
            const size_t packed = planes * (size_t)bw * bh;
            if (!inflate_all(src, size, &tmp, packed, error)) return false;
            if (tmp.size() != packed) { *error = "EXR block has the wrong size"; return false; }
            raw.resize(expect);
            size_t in = 0, o = 0;
            for (int y = 0; y < bh; ++y)
                for (int t : channel_types) {
                    const size_t n = (size_t)bw;
                    uint32_t pixel = 0;
                    const uint8_t* q = tmp.data() + in;
                    if (t == 1) {
                        for (size_t x = 0; x < n; ++x) { pixel += (uint32_t)q[x] << 8 | q[n + x]; raw[o++] = (uint8_t)pixel; raw[o++] = (uint8_t)(pixel >> 8); }
                        in += 2 * n;
                    } else if (t == 0) {
                        for (size_t x = 0; x < n; ++x) { pixel += (uint32_t)q[x] << 24 | (uint32_t)q[n + x] << 16 | (uint32_t)q[2 * n + x] << 8 | q[3 * n + x];
                                                         raw[o++] = (uint8_t)pixel; raw[o++] = (uint8_t)(pixel >> 8); raw[o++] = (uint8_t)(pixel >> 16); raw[o++] = (uint8_t)(pixel >> 24); }
                        in += 4 * n;
                    } else {
                        for (size_t x = 0; x < n; ++x) { pixel += (uint32_t)q[x] << 24 | (uint32_t)q[n + x] << 16 | (uint32_t)q[2 * n + x] << 8;
                                                         raw[o++] = (uint8_t)pixel; raw[o++] = (uint8_t)(pixel >> 8); raw[o++] = (uint8_t)(pixel >> 16); raw[o++] = (uint8_t)(pixel >> 24); }
                        in += 3 * n;
                    }
                }
            data = raw.data();
        } else if (compression == 4 && size < expect) {
            if (!piz::decode_block(src, size, channel_types, bw, bh, &raw, error)) return false;
            if (raw.size() != expect) { *error = "EXR block has the wrong size"; return false; }
            data = raw.data();
        } else if (compression != 0 && size < expect) {
            if (compression == 1) {  // RLE
                tmp.clear();
                size_t q = 0;
                while (q < size) {
                    int n = (int8_t)src[q++];
                    if (n < 0) { n = -n; if (q + n > size) { *error = "bad EXR RLE data"; return false; } tmp.insert(tmp.end(), src + q, src + q + n); q += n; }
                    else { if (q >= size) { *error = "bad EXR RLE data"; return false; } tmp.insert(tmp.end(), (size_t)n + 1, src[q++]); }
                    if (tmp.size() > expect) { *error = "EXR RLE block is longer than its tile"; return false; }
                }
            } else if (!inflate_all(src, size, &tmp, expect, error)) return false;
            if (tmp.size() != expect) { *error = "EXR block has the wrong size"; return false; }
            for (size_t k = 1; k < expect; ++k) tmp[k] = (uint8_t)(tmp[k - 1] + tmp[k] - 128);   // predictor
            raw.resize(expect);
            size_t half = (expect + 1) / 2;                                                     // de-interleave
            for (size_t k = 0; k < expect; ++k) raw[k] = (k & 1) ? tmp[half + k / 2] : tmp[k / 2];
            data = raw.data();
        } else if (size != expect) { *error = "EXR block has the wrong size"; return false; }
        for (int y = 0; y < bh; ++y) {
            const uint8_t* row = data + pixel_bytes * (size_t)bw * y;
            for (auto& c : channels) {
                size_t bytes = c.type == 1 ? 2 : 4;
                if (c.slot >= 0)
                    for (int x = 0; x < bw; ++x) {
                        float v;
                        if (c.type == 1) v = half_to_float(le16(row + 2 * x));
                        else if (c.type == 2) { uint32_t u = le32(row + 4 * x); memcpy(&v, &u, 4); }
                        else v = (float)le32(row + 4 * x);
                        out->data[((size_t)(by + y) * W + bx + x) * 4 + c.slot] = v;
                    }
                row += bytes * bw;
            }
        }
    }
    return true;
}

}  // namespace pth
