// Micro-benchmark (dev tool): issue cost of the instruction mix of the attention inner loop on gfx950, in
// shader cycles (s_memtime) per wave-instruction, at 1 / 2 / 3 / 4 waves per SIMD:
//   v_exp_f32, v_cvt_pk_bf16_f32, v_max3_f32, v_fma_f32, v_mfma_f32_32x32x16_bf16 alone, and the mixes
//   [1 MFMA : 6 VALU] and [1 MFMA : 8 exp : 4 cvt] that a softmax tile would like to sustain.
// Build / run on the box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_rate scripts/micro/issue_rate.hip && /tmp/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int ITER = 256;

// The instruction streams are inline asm (volatile: kept, in order), so the compiler cannot fold or re-schedule them.
// MODE 0: 16 v_exp; 1: 16 v_cvt_pk_bf16_f32; 2: 16 v_max3; 3: 16 v_fma; 4: 8 MFMA over 4 accumulators;
// 5: 8 x [MFMA + 6 fma]; 6: 8 x [MFMA + 4 exp + 2 cvt]  (one softmax tile: 8 MFMA, 32 exp, 16 cvt);
// 7: 32 exp + 16 cvt without MFMAs; 8: mode 6 + 16 max3 + 2 permlane-free extras (a whole tile's VALU);
// 9: 8 x [MFMA + 4 exp + 2 cvt + 2 max3 + 1 salu-ish] with MFMAs on 2 accumulators only (dependent pairs)
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]))
#define CVT(d, a, b) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[d]) : "v"(v[a]), "v"(v[b]))
#define MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(w[i]) : "v"(w[(i + 1) & 7]), "v"(w[(i + 5) & 7]))
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2))
#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb))
template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float* sink, float seed) {
    extern __shared__ float lds_pad[];   // 100 KB requested at launch: exactly one workgroup per CU
    const int lane = threadIdx.x & 63;
    if (seed == 123.f) lds_pad[threadIdx.x] = seed;
    float v[16], w[8];
    unsigned u[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = -seed * (float)(lane + i) * 1e-3f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { w[i] = v[i] * 0.5f; u[i] = 0; }
    float c1 = 0.999f + seed * 1e-9f, c2 = seed * 1e-7f;
    f32x16 a0, a1, a2, a3;
#pragma unroll
    for (int i = 0; i < 16; ++i) { a0[i] = v[i]; a1[i] = -v[i]; a2[i] = v[i] * 2; a3[i] = v[i] * 3; }
    bf16x8 fa, fb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(v[i] * 1e-2f); fb[i] = (__bf16)(v[i + 8] * 1e-2f); }
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(fa), "+v"(fb));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) EXP(i);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) CVT(i & 7, i, (i + 3) & 15);
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) MAX3(i & 7);
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) FMA(i);
        } else if (MODE == 4) {
            MFMA(a0); MFMA(a1); MFMA(a2); MFMA(a3); MFMA(a0); MFMA(a1); MFMA(a2); MFMA(a3);
        } else if (MODE == 5) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if ((g & 3) == 0) MFMA(a0); else if ((g & 3) == 1) MFMA(a1); else if ((g & 3) == 2) MFMA(a2); else MFMA(a3);
#pragma unroll
                for (int i = 0; i < 6; ++i) FMA((g * 6 + i) & 15);
            }
        } else if (MODE == 6 || MODE == 7 || MODE == 8 || MODE == 9) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (MODE == 9) { if (g & 1) MFMA(a1); else MFMA(a0); }
                else if (MODE != 7) { if ((g & 3) == 0) MFMA(a0); else if ((g & 3) == 1) MFMA(a1); else if ((g & 3) == 2) MFMA(a2); else MFMA(a3); }
                EXP((g * 4 + 0) & 15); EXP((g * 4 + 1) & 15); EXP((g * 4 + 2) & 15); EXP((g * 4 + 3) & 15);
                CVT((2 * g) & 7, (g * 4 + 8) & 15, (g * 4 + 9) & 15);
                CVT((2 * g + 1) & 7, (g * 4 + 10) & 15, (g * 4 + 11) & 15);
                if (MODE >= 8) { MAX3(g & 7); MAX3((g + 3) & 7); }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = c1 + c2;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i] + a0[i] + a1[i] + a2[i] + a3[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (float)u[i] + w[i];
    if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
    if (s == 12345.678f) sink[0] = s;
}

template <int MODE>
void run(const char* name, double insts_per_iter, unsigned long long* out, float* sink) {
    printf("%-44s", name);
    // ONE workgroup of 256 * wps threads per CU (100 KB of LDS each: a second one cannot be co-resident), so every
    // SIMD holds exactly wps waves of it (a workgroup's waves are dealt round-robin over the 4 SIMDs)
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int wps = 1; wps <= 4; ++wps) {
        const int grid = 256;
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256 * wps), 100 * 1024, 0, out, sink, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid * 4 * wps);
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double cyc = (double)h[h.size() / 2] / ITER, mx = (double)h.back() / ITER;
        printf("  %dw/SIMD: %6.1f (max %6.1f) cyc/iter = %5.2f /inst/SIMD |", wps, cyc, mx, cyc / insts_per_iter / wps);
    }
    printf("\n");
}

int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 8192 * 8 * 8); hipMalloc(&sink, 4);
    run<0>("16 x v_exp_f32", 16, out, sink);
    run<1>("16 x v_cvt_pk_bf16_f32", 16, out, sink);
    run<2>("16 x v_max3_f32", 16, out, sink);
    run<3>("16 x v_fma_f32", 16, out, sink);
    run<4>("8 x mfma 32x32x16 bf16 (4 accumulators)", 8, out, sink);
    run<5>("8 x [mfma + 6 fma]", 56, out, sink);
    run<6>("8 x [mfma + 4 exp + 2 cvt]", 56, out, sink);
    run<7>("8 x [4 exp + 2 cvt] (no mfma)", 48, out, sink);
    run<8>("8 x [mfma + 4 exp + 2 cvt + 2 max3]", 72, out, sink);
    run<9>("same, mfma on 2 accumulators (dependent)", 72, out, sink);
    return 0;
}
