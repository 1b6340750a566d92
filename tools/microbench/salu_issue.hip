// Scalar-unit issue rates on gfx950, alone and mixed into a VALU stream (the light-sample kernel issues 0.78 scalar instructions per vector one:
// is the scalar unit, one per CU, a bound of its own?).  hipcc --offload-arch=gfx950 -O2 -o salu_issue salu_issue.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum Kind { K_FMA, K_SALU64, K_SALU32, K_FMA_SALU_1_1, K_FMA_SALU_2_1, K_FMA_SALU_1_2, K_FMA_SALU_4_1, K_CMP_SDST_AND, K_FMA_BRANCH, K_FMA_WAITCNT, K_FMA_NOP, K_FMA_SMOV, K_COUNT };
static const char* kNames[K_COUNT] = {"v_fma_f32", "s_and_b64 alone (8 independent pairs)", "s_add_u32 alone (8 independent)", "v_fma_f32 : s_and_b64 = 1:1 (fma counted)",
                                      "v_fma_f32 : s_and_b64 = 2:1 (fma counted)", "v_fma_f32 : s_and_b64 = 1:2 (fma counted)", "v_fma_f32 : s_and_b64 = 4:1 (fma counted)",
                                      "v_cmp_lt_f32 -> sgpr pair + s_and_b64 (pairs counted)", "v_fma_f32 + s_cbranch_scc1 not taken (fma counted)",
                                      "v_fma_f32 + s_waitcnt, nothing outstanding (fma counted)", "v_fma_f32 + s_nop 0 (fma counted)", "v_fma_f32 + s_mov_b32 (fma counted)"};

template <int KIND>
__global__ void __launch_bounds__(256) k_issue(int iters, float seed, float* out) {
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = seed + (float)i + (float)threadIdx.x;
    float b = seed * 0.5f + 1.0f, c = seed * 0.25f;
    uint64_t s[8]; uint32_t t[8];
    for (int i = 0; i < 8; ++i) { s[i] = (uint64_t)__builtin_amdgcn_readfirstlane((int)(seed * 7.0f) + i) * 0x100000001ull; t[i] = (uint32_t)s[i]; }
    for (int it = 0; it < iters; ++it) {
        if (KIND == K_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_SALU64) {
#define X(i) asm volatile("s_and_b64 %0, %0, exec" : "+s"(s[i & 7]) : : "scc");
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_SALU32) {
#define X(i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(t[i & 7]) : : "scc");
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_FMA_SALU_1_1) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n\ts_and_b64 %1, %1, exec" : "+v"(a[i]), "+s"(s[i & 7]) : "v"(b), "v"(c) : "scc");
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_FMA_SALU_2_1) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %3, %4\n\tv_fma_f32 %1, %1, %3, %4\n\ts_and_b64 %2, %2, exec" : "+v"(a[i]), "+v"(a[(i + 8) & 15]), "+s"(s[i & 7]) : "v"(b), "v"(c) : "scc");
            REP16(X)
#undef X
        } else if (KIND == K_FMA_SALU_1_2) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %3, %4\n\ts_and_b64 %1, %1, exec\n\ts_and_b64 %2, %2, exec" : "+v"(a[i]), "+s"(s[i & 7]), "+s"(s[(i + 4) & 7]) : "v"(b), "v"(c) : "scc");
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_FMA_SALU_4_1) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %3, %4\n\tv_fma_f32 %1, %1, %3, %4\n\tv_fma_f32 %0, %0, %3, %4\n\tv_fma_f32 %1, %1, %3, %4\n\ts_and_b64 %2, %2, exec" : "+v"(a[i]), "+v"(a[(i + 8) & 15]), "+s"(s[i & 7]) : "v"(b), "v"(c) : "scc");
            REP16(X)
#undef X
        } else if (KIND == K_CMP_SDST_AND) {
#define X(i) asm volatile("v_cmp_lt_f32 %1, %0, %2\n\ts_and_b64 %1, %1, exec" : "+v"(a[i]), "+s"(s[i & 7]) : "v"(b) : "scc");
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_FMA_BRANCH) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_cmp_eq_u32 %3, 0x12345\n\ts_cbranch_scc1 .Lnever%=\n.Lnever%=:" : "+v"(a[i]) : "v"(b), "v"(c), "s"(t[i & 7]) : "scc");
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_FMA_WAITCNT) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(a[i]) : "v"(b), "v"(c));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_FMA_NOP) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_nop 0" : "+v"(a[i]) : "v"(b), "v"(c));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_FMA_SMOV) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n\ts_mov_b32 %1, 0x368637bd" : "+v"(a[i]), "=s"(t[i & 7]) : "v"(b), "v"(c));
            REP16(X) REP16(X)
#undef X
        }
    }
    float r = 0.0f;
    for (int i = 0; i < 16; ++i) r += a[i];
    for (int i = 0; i < 8; ++i) r += (float)s[i] + (float)t[i];
    if (r == 12345.678f) out[0] = r;
}

template <int KIND>
void run(int cus, int waves_per_simd, int iters, float* d_out) {
    const int grid = cus * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(256), 0, 0, iters / 8, 1.0f, d_out);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(256), 0, 0, iters, 1.0f, d_out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double per_iter = (KIND == K_FMA_SALU_2_1) ? 32.0 : (KIND == K_FMA_SALU_4_1 ? 64.0 : 32.0);
    const double wave_instr = (double)grid * 4.0 * iters * per_iter;
    printf("%-62s W=%d  %8.1f G/s   %7.3f ms\n", kNames[KIND], waves_per_simd, wave_instr / (ms * 1e-3) / 1e9, ms);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs, clock %d MHz: one scalar instruction per CU per cycle would be %.0f G/s, one wave-instruction per SIMD per cycle %.0f G/s\n", prop.gcnArchName, cus,
           prop.clockRate / 1000, cus * 1.0 * prop.clockRate * 1e3 / 1e9, cus * 4.0 * prop.clockRate * 1e3 / 1e9);
    float* d_out; hipMalloc(&d_out, 64);
    const int iters = 20000;
    const int ws[] = {1, 4, 8};
    for (int w : ws) {
        run<K_FMA>(cus, w, iters, d_out);
        run<K_SALU64>(cus, w, iters, d_out);
        run<K_SALU32>(cus, w, iters, d_out);
        run<K_FMA_SALU_4_1>(cus, w, iters, d_out);
        run<K_FMA_SALU_2_1>(cus, w, iters, d_out);
        run<K_FMA_SALU_1_1>(cus, w, iters, d_out);
        run<K_FMA_SALU_1_2>(cus, w, iters, d_out);
        run<K_CMP_SDST_AND>(cus, w, iters, d_out);
        run<K_FMA_BRANCH>(cus, w, iters, d_out);
        run<K_FMA_WAITCNT>(cus, w, iters, d_out);
        run<K_FMA_NOP>(cus, w, iters, d_out);
        run<K_FMA_SMOV>(cus, w, iters, d_out);
        printf("\n");
    }
    return 0;
}
