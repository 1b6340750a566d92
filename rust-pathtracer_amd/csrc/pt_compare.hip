// pt_compare.hip — film comparison: the modes of the reference's compare_exr tool
// (src/bin/compare_exr.rs:39-52,70-170: absolute difference, per-pixel RMSE shown through colorgrad's viridis, relative
// error) on raw float4 images, plus per-channel L-inf / mean-abs / RMSE statistics.  One lane per pixel; statistics are
// reduced per workgroup in LDS (f64) and finished on the host in workgroup order.
//
// colorgrad is not in the reference tree (Cargo.toml dependency): its viridis preset is restated as the uniform B-spline
// ("basis" interpolation, as in d3-interpolate) through the preset's nine sRGB key colours, evaluated in f64.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pt_api.h"
#include "../../include/pt_numerics.h"
#include "pt_error.h"

namespace {

constexpr int kBlock = 256;
constexpr int kGrid = 1024;
constexpr int kPartial = 12;  // linf[4], sum_abs[4], sum_sq, min, max, nonfinite

__device__ const double kViridis[9][3] = {
    {0x44 / 255.0, 0x01 / 255.0, 0x54 / 255.0}, {0x48 / 255.0, 0x27 / 255.0, 0x77 / 255.0}, {0x3f / 255.0, 0x4a / 255.0, 0x8a / 255.0},
    {0x31 / 255.0, 0x67 / 255.0, 0x8e / 255.0}, {0x26 / 255.0, 0x83 / 255.0, 0x8f / 255.0}, {0x1f / 255.0, 0x9d / 255.0, 0x8a / 255.0},
    {0x6c / 255.0, 0xce / 255.0, 0x5a / 255.0}, {0xb6 / 255.0, 0xde / 255.0, 0x2b / 255.0}, {0xfe / 255.0, 0xe8 / 255.0, 0x25 / 255.0}};

__device__ inline double basis(double t1, double v0, double v1, double v2, double v3) {
    double t2 = t1 * t1, t3 = t2 * t1;
    return ((1.0 - 3.0 * t1 + 3.0 * t2 - t3) * v0 + (4.0 - 6.0 * t2 + 3.0 * t3) * v1 + (1.0 + 3.0 * t1 + 3.0 * t2 - 3.0 * t3) * v2 + t3 * v3) / 6.0;
}
__device__ inline void viridis(double t, float* rgb) {
    if (!(t >= 0.0)) t = 0.0;
    if (t > 1.0) t = 1.0;
    const int n = 9;
    int i = t >= 1.0 ? n - 2 : (int)(t * (double)(n - 1));
    double t1 = (t - (double)i / (double)(n - 1)) * (double)(n - 1);
    for (int c = 0; c < 3; ++c) {
        double v1 = kViridis[i][c], v2 = kViridis[i + 1][c];
        double v0 = i > 0 ? kViridis[i - 1][c] : 2.0 * v1 - v2, v3 = i < n - 2 ? kViridis[i + 2][c] : 2.0 * v2 - v1;
        double v = basis(t1, v0, v1, v2, v3);
        rgb[c] = (float)(v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v));
    }
}

__device__ inline double wave_sum(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o); return v; }
__device__ inline double wave_max(double v) { for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o)); return v; }
__device__ inline double wave_min(double v) { for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o)); return v; }

__global__ void __launch_bounds__(kBlock) k_compare(const float4* __restrict__ image, const float4* __restrict__ truth, uint32_t n, int mode,
                                                   float4* __restrict__ out, double* __restrict__ partial) {
    __shared__ double red[kBlock / 64][kPartial];
    double acc[kPartial];
    for (int k = 0; k < kPartial; ++k) acc[k] = 0.0;
    acc[9] = INFINITY; acc[10] = -INFINITY;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 a4 = image[i], b4 = truth[i];
        const float a[4] = {a4.x, a4.y, a4.z, a4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
        float o[4], d[4];
        bool bad = false;
        for (int c = 0; c < 4; ++c) {
            d[c] = a[c] - b[c];
            bad = bad || !(pt_abs(a[c]) < PT_INF) || !(pt_abs(b[c]) < PT_INF);
        }
        float value;
        if (mode == PT_COMPARE_RMSE) {
            float r = pt_sqrt(((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) / 4.0f);
            o[0] = o[1] = o[2] = r; o[3] = 0.0f; value = r;
        } else if (mode == PT_COMPARE_RELATIVE) {
            for (int c = 0; c < 4; ++c) { float r = pt_abs(d[c]) / b[c]; o[c] = (pt_abs(r) < PT_INF) ? r : 0.0f; }
            value = __builtin_fmaxf(__builtin_fmaxf(o[0], o[1]), __builtin_fmaxf(o[2], o[3]));
        } else {
            for (int c = 0; c < 4; ++c) o[c] = pt_abs(d[c]);
            value = __builtin_fmaxf(__builtin_fmaxf(o[0], o[1]), __builtin_fmaxf(o[2], o[3]));
        }
        if (out) out[i] = make_float4(o[0], o[1], o[2], o[3]);
        if (bad) { acc[11] += 1.0; continue; }
        for (int c = 0; c < 4; ++c) {
            double ad = (double)pt_abs(d[c]);
            acc[c] = fmax(acc[c], ad); acc[4 + c] += ad; acc[8] += (double)d[c] * (double)d[c];
        }
        acc[9] = fmin(acc[9], (double)value); acc[10] = fmax(acc[10], (double)value);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = 0; k < kPartial; ++k) {
        double v = (k < 4 || k == 10) ? wave_max(acc[k]) : (k == 9 ? wave_min(acc[k]) : wave_sum(acc[k]));
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < kPartial) {
        const int k = threadIdx.x;
        double v = red[0][k];
        for (int w = 1; w < kBlock / 64; ++w) v = (k < 4 || k == 10) ? fmax(v, red[w][k]) : (k == 9 ? fmin(v, red[w][k]) : v + red[w][k]);
        partial[(size_t)blockIdx.x * kPartial + k] = v;
    }
}

// RMSE mode, second pass: the gradient over [lo, hi] (compare_exr.rs:112-127)
__global__ void __launch_bounds__(kBlock) k_colormap(float4* __restrict__ out, uint32_t n, float lo, float hi) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float r = out[i].x, rgb[3];
        viridis((double)((r - lo) / (hi - lo)), rgb);
        out[i] = make_float4(rgb[0], rgb[1], rgb[2], 1.0f);
    }
}

pt_status cfail(pt_status st, const std::string& m) { pt_set_error(m); return st; }

}  // namespace

extern "C" pt_status pt_compare_films(uint32_t width, uint32_t height, const float* image, const float* truth, int32_t mode, float* out,
                                      pt_compare_stats* stats) {
    if (!image || !truth) return cfail(PT_ERR_INVALID_ARGUMENT, "null argument");
    if (width == 0 || height == 0) return cfail(PT_ERR_INVALID_ARGUMENT, "image dimensions must be positive");
    if (mode < PT_COMPARE_ABSOLUTE || mode > PT_COMPARE_RELATIVE) return cfail(PT_ERR_INVALID_ARGUMENT, "unknown comparison mode");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return cfail(PT_ERR_NO_DEVICE, "no HIP device available: the product path has no CPU fallback");
    const uint32_t n = width * height;
    float4 *d_a = nullptr, *d_b = nullptr, *d_o = nullptr; double* d_p = nullptr;
    hipError_t e = hipMalloc(&d_a, sizeof(float4) * n);
    if (e == hipSuccess) e = hipMalloc(&d_b, sizeof(float4) * n);
    if (e == hipSuccess && out) e = hipMalloc(&d_o, sizeof(float4) * n);
    if (e == hipSuccess) e = hipMalloc(&d_p, sizeof(double) * kPartial * kGrid);
    if (e == hipSuccess) e = hipMemcpy(d_a, image, sizeof(float4) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_b, truth, sizeof(float4) * n, hipMemcpyHostToDevice);
    std::vector<double> part((size_t)kPartial * kGrid);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_compare, dim3(kGrid), dim3(kBlock), 0, 0, d_a, d_b, n, (int)mode, d_o, d_p);
        e = hipMemcpy(part.data(), d_p, sizeof(double) * part.size(), hipMemcpyDeviceToHost);
    }
    pt_compare_stats st; memset(&st, 0, sizeof(st));
    double sum_sq = 0.0, lo = INFINITY, hi = -INFINITY, bad = 0.0;
    for (int b = 0; b < kGrid; ++b) {
        const double* p = &part[(size_t)b * kPartial];
        for (int c = 0; c < 4; ++c) { st.linf[c] = std::fmax(st.linf[c], p[c]); st.mean_abs[c] += p[4 + c]; }
        sum_sq += p[8]; lo = std::fmin(lo, p[9]); hi = std::fmax(hi, p[10]); bad += p[11];
    }
    const double good = (double)n - bad;
    for (int c = 0; c < 4; ++c) st.mean_abs[c] = good > 0 ? st.mean_abs[c] / good : 0.0;
    st.rmse = good > 0 ? std::sqrt(sum_sq / (4.0 * good)) : 0.0;
    st.pixel_min = good > 0 ? (float)lo : 0.0f; st.pixel_max = good > 0 ? (float)hi : 0.0f;
    st.nonfinite = (uint64_t)bad;
    if (e == hipSuccess && out) {
        if (mode == PT_COMPARE_RMSE) hipLaunchKernelGGL(k_colormap, dim3(kGrid), dim3(kBlock), 0, 0, d_o, n, st.pixel_min, st.pixel_max);
        e = hipMemcpy(out, d_o, sizeof(float4) * n, hipMemcpyDeviceToHost);
    }
    hipFree(d_a); hipFree(d_b); hipFree(d_o); hipFree(d_p);
    if (e != hipSuccess) return cfail(PT_ERR_DEVICE, hipGetErrorString(e));
    if (stats) *stats = st;
    return PT_OK;
}
