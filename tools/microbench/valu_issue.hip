// valu_issue.hip — what is the VALU issue rate of gfx950 for the instructions the path tracer is made of?
//
// One kernel per instruction kind: every wave runs ITER x 32 independent copies of the instruction (16 accumulators, written as
// inline assembly so that nothing is folded), with W waves resident per SIMD (grid = CUs x W workgroups of 256 threads; 4 waves
// per workgroup -> one per SIMD).  Reported: wave-instructions per second for the whole chip, and cycles per wave-instruction per
// SIMD measured with s_memtime inside the kernel (clock-independent).  DESIGN.md section 6 quotes the result; the raw output is kept
// under profiles/.
//   hipcc --offload-arch=gfx950 -O2 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum Kind { K_FMA, K_PK_FMA, K_MAX3, K_MIN, K_CMP_CNDMASK, K_RCP, K_FMA64, K_MUL_HI_U32, K_AND_OR, K_MAD_U32_U24, K_MUL_F32_SGPR, K_DS_READ_B128, K_MIX_SALU, K_ADD, K_MUL, K_FMAC, K_FMA_SGPR, K_FMA_LIT, K_MAX, K_CNDMASK, K_CMP, K_MOV, K_LSHL, K_XOR, K_CVT, K_MIX_FMA_MIN, K_BPERMUTE, K_DS_READ_B32_SPREAD, K_READFIRSTLANE, K_MAD_U64_U32, K_MUL_LO_U32, K_COUNT };
static const char* kNames[K_COUNT] = {"v_fma_f32", "v_pk_fma_f32 (2 fma per lane)", "v_max3_f32", "v_min_f32", "v_cmp_lt_f32 + v_cndmask_b32 (pair)", "v_rcp_f32",
                                      "v_fma_f64", "v_mul_hi_u32", "v_and_or_b32", "v_mad_u32_u24", "v_mul_f32 with SGPR operand", "ds_read_b128 (same address: broadcast)",
                                      "v_fma_f32 + s_add_u32 (1:1)", "v_add_f32", "v_mul_f32", "v_fmac_f32", "v_fma_f32 with SGPR operand", "v_fma_f32 with literal", "v_max_f32", "v_cndmask_b32 (vcc fixed)", "v_cmp_lt_f32 (to vcc)", "v_mov_b32", "v_lshlrev_b32", "v_xor_b32", "v_cvt_f32_u32", "v_fma_f32 + v_min_f32 (1:1; counted as 2)", "ds_bpermute_b32", "ds_read_b32 (lane-strided addresses)", "v_readfirstlane_b32", "v_mad_u64_u32 (32x32+64 -> 64)", "v_mul_lo_u32"};

template <int KIND>
__global__ void __launch_bounds__(256) k_issue(int iters, float seed, float* out, unsigned long long* cycles) {
    __shared__ float4 lds_data[64];
    if (threadIdx.x < 64) lds_data[threadIdx.x] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    float a[16]; double dd[16]; uint32_t u[16]; float2 p[16]; unsigned long long uu[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed + (float)i + (float)threadIdx.x; dd[i] = a[i]; u[i] = (uint32_t)(a[i] * 1000.0f); uu[i] = u[i]; p[i] = make_float2(a[i], a[i] + 1.0f); }
    float b = seed * 0.5f + 1.0f, c = seed * 0.25f;
    double db = b, dc = c;
    uint32_t sacc = 0;
    const uint32_t lds_addr = (uint32_t)(uintptr_t)lds_data;
    float4 ld[4] = {};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (KIND == K_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_PK_FMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_MAX3) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_MIN) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_CMP_CNDMASK) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
            REP16(X)
#undef X
        } else if (KIND == K_RCP) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_FMA64) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(dd[i]) : "v"(db), "v"(dc));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_MUL_HI_U32) {
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_AND_OR) {
#define X(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(u[(i + 2) & 15]));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_MAD_U32_U24) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(u[(i + 2) & 15]));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_MUL_F32_SGPR) {
#define X(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(seed));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_DS_READ_B128) {
#define X(i) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[i & 3]) : "v"(lds_addr));
            REP16(X) REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (KIND == K_MIX_SALU) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n\ts_add_u32 %1, %1, 1" : "+v"(a[i]), "+s"(sacc) : "v"(b), "v"(c) : "scc");
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_FMAC) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_FMA_SGPR) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(seed), "v"(c));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_FMA_LIT) {
#define X(i) asm volatile("v_fma_f32 %0, %0, 0.5, %1" : "+v"(a[i]) : "v"(c));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_MAX) {
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_CMP) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_MOV) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_LSHL) {
#define X(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_XOR) {
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_CVT) {
#define X(i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_MIX_FMA_MIN) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_min_f32 %1, %1, %2" : "+v"(a[i]), "+v"(p[i].x) : "v"(b), "v"(c));
            REP16(X)
#undef X
        } else if (KIND == K_BPERMUTE) {
#define X(i) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
            REP16(X) REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (KIND == K_DS_READ_B32_SPREAD) {
#define X(i) asm volatile("ds_read_b32 %0, %1" : "=v"(ld[i & 3].x) : "v"(lds_addr + (threadIdx.x & 63u) * 4u));
            REP16(X) REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (KIND == K_READFIRSTLANE) {
#define X(i) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sacc) : "v"(u[i]));
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_MAD_U64_U32) {   // (what the compiler makes of Philox's (uint64_t)M * c: one instruction for both halves)
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(uu[i]) : "v"(u[i]), "v"(u[(i + 1) & 15]) : "vcc");
            REP16(X) REP16(X)
#undef X
        } else if (KIND == K_MUL_LO_U32) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
            REP16(X) REP16(X)
#undef X
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0.0f;
    for (int i = 0; i < 16; ++i) r += a[i] + (float)dd[i] + (float)u[i] + p[i].x + p[i].y + (float)uu[i];
    r += ld[0].x + ld[1].y + ld[2].z + ld[3].w + (float)sacc;
    if (r == 12345.678f) out[0] = r;  // keep everything alive
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(int cus, int waves_per_simd, int iters, float* d_out, unsigned long long* d_cycles) {
    const int grid = cus * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(256), 0, 0, iters / 8, 1.0f, d_out, d_cycles);  // warm
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(256), 0, 0, iters, 1.0f, d_out, d_cycles);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> cyc(grid);
    hipMemcpy(cyc.data(), d_cycles, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
    double mean = 0; for (auto c : cyc) mean += (double)c; mean /= grid;
    const double per_iter = KIND == K_CMP_CNDMASK ? 16.0 : 32.0;  // (K_MIX_FMA_MIN: 16 pairs = 32 instructions)  // instruction groups per loop iteration
    const double wave_instr = (double)grid * 4.0 * iters * per_iter;
    // s_memtime ticks at a constant 100 MHz on gfx9: convert through the measured kernel time instead
    printf("%-42s W=%d  %8.1f G wave-instr/s   %6.3f ms   memtime ticks/instr/wave %.4f\n", kNames[KIND], waves_per_simd, wave_instr / (ms * 1e-3) / 1e9, ms, mean / (iters * per_iter));
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs, clock %d MHz -> 4 SIMDs per CU: one wave-instr per SIMD per cycle would be %.0f G/s at that clock\n", prop.gcnArchName, cus, prop.clockRate / 1000,
           cus * 4.0 * prop.clockRate * 1e3 / 1e9);
    float* d_out; unsigned long long* d_cycles;
    hipMalloc(&d_out, 64); hipMalloc(&d_cycles, sizeof(unsigned long long) * cus * 8);
    const int iters = 20000;
    const int ws[] = {1, 2, 5, 8};
    for (int w : ws) {
        run<K_FMA>(cus, w, iters, d_out, d_cycles);
        run<K_PK_FMA>(cus, w, iters, d_out, d_cycles);
        run<K_MAX3>(cus, w, iters, d_out, d_cycles);
        run<K_MIN>(cus, w, iters, d_out, d_cycles);
        run<K_CMP_CNDMASK>(cus, w, iters, d_out, d_cycles);
        run<K_RCP>(cus, w, iters, d_out, d_cycles);
        run<K_FMA64>(cus, w, iters, d_out, d_cycles);
        run<K_MUL_HI_U32>(cus, w, iters, d_out, d_cycles);
        run<K_AND_OR>(cus, w, iters, d_out, d_cycles);
        run<K_MAD_U32_U24>(cus, w, iters, d_out, d_cycles);
        run<K_MUL_F32_SGPR>(cus, w, iters, d_out, d_cycles);
        run<K_DS_READ_B128>(cus, w, iters, d_out, d_cycles);
        run<K_MIX_SALU>(cus, w, iters, d_out, d_cycles);
        run<K_ADD>(cus, w, iters, d_out, d_cycles);
        run<K_MUL>(cus, w, iters, d_out, d_cycles);
        run<K_FMAC>(cus, w, iters, d_out, d_cycles);
        run<K_FMA_SGPR>(cus, w, iters, d_out, d_cycles);
        run<K_FMA_LIT>(cus, w, iters, d_out, d_cycles);
        run<K_MAX>(cus, w, iters, d_out, d_cycles);
        run<K_CNDMASK>(cus, w, iters, d_out, d_cycles);
        run<K_CMP>(cus, w, iters, d_out, d_cycles);
        run<K_MOV>(cus, w, iters, d_out, d_cycles);
        run<K_LSHL>(cus, w, iters, d_out, d_cycles);
        run<K_XOR>(cus, w, iters, d_out, d_cycles);
        run<K_CVT>(cus, w, iters, d_out, d_cycles);
        run<K_MIX_FMA_MIN>(cus, w, iters, d_out, d_cycles);
        run<K_BPERMUTE>(cus, w, iters, d_out, d_cycles);
        run<K_DS_READ_B32_SPREAD>(cus, w, iters, d_out, d_cycles);
        run<K_READFIRSTLANE>(cus, w, iters, d_out, d_cycles);
        run<K_MAD_U64_U32>(cus, w, iters, d_out, d_cycles);
        run<K_MUL_LO_U32>(cus, w, iters, d_out, d_cycles);
        printf("\n");
    }
    return 0;
}
