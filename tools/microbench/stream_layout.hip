// Microbenchmark (MI355X): does it matter to a streaming kernel whether a record of R 32-bit fields lives as R separate arrays (one
// dword per lane and instruction: the engine's SoA queues) or as R/4 arrays of 16-byte groups (one dwordx4 per lane and instruction)?
// Each thread reads RIN words of its item and writes ROUT words (their sum folded in, so that nothing is optimised away), grid-stride
// over N items, like k_shade: 28 words in, 33 out.   build: hipcc --offload-arch=gfx950 -O2 -o stream_layout stream_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int RIN, int ROUT>
__global__ void __launch_bounds__(256) k_words(const float* __restrict__ in, float* __restrict__ out, size_t cap, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v[RIN], s = 0.0f;
#pragma unroll
        for (int k = 0; k < RIN; ++k) { v[k] = in[(size_t)k * cap + i]; s += v[k]; }
#pragma unroll
        for (int k = 0; k < ROUT; ++k) out[(size_t)k * cap + i] = v[k % RIN] + s;
    }
}
template <int GIN, int GOUT>
__global__ void __launch_bounds__(256) k_quads(const float4* __restrict__ in, float4* __restrict__ out, size_t cap, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v[GIN]; float s = 0.0f;
#pragma unroll
        for (int k = 0; k < GIN; ++k) { v[k] = in[(size_t)k * cap + i]; s += v[k].x + v[k].y + v[k].z + v[k].w; }
#pragma unroll
        for (int k = 0; k < GOUT; ++k) { float4 w = v[k % GIN]; w.x += s; out[(size_t)k * cap + i] = w; }
    }
}
template <typename F> static float time_ms(F&& launch, int reps) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) launch();
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}
int main() {
    const size_t cap = 128u << 20, n = 40u << 20;   // queue capacity of the engine's default batch; items of one k_shade launch
    float *in, *out;
    CHECK(hipMalloc(&in, cap * 32 * sizeof(float))); CHECK(hipMalloc(&out, cap * 36 * sizeof(float)));
    CHECK(hipMemset(in, 0, cap * 32 * sizeof(float)));
    for (int grid : {256 * 8, 256 * 64}) {
        float ms = time_ms([&] { hipLaunchKernelGGL((k_words<28, 33>), dim3(grid), dim3(256), 0, 0, in, out, cap, n); }, 5);
        printf("grid %6d  words  28 in / 33 out: %7.3f ms  %7.1f GB/s\n", grid, ms, n * 61.0 * 4 / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL((k_quads<7, 9>), dim3(grid), dim3(256), 0, 0, (const float4*)in, (float4*)out, cap, n); }, 5);
        printf("grid %6d  quads   7 in /  9 out: %7.3f ms  %7.1f GB/s\n", grid, ms, n * 64.0 * 4 / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL((k_words<6, 11>), dim3(grid), dim3(256), 0, 0, in, out, cap, n); }, 5);
        printf("grid %6d  words   6 in / 11 out: %7.3f ms  %7.1f GB/s\n", grid, ms, n * 17.0 * 4 / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL((k_quads<2, 3>), dim3(grid), dim3(256), 0, 0, (const float4*)in, (float4*)out, cap, n); }, 5);
        printf("grid %6d  quads   2 in /  3 out: %7.3f ms  %7.1f GB/s\n", grid, ms, n * 20.0 * 4 / ms / 1e6);
    }
    return 0;
}
