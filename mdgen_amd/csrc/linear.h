// The fp32 / bf16-operand linear layer's parameter block and epilogue, shared by k_fp32.hip (k32_linear, k16_linear,
// k16_linear_fast) and k_wide16.hip (k16_linear_wide).
#pragma once
#include "common.h"

namespace mdg {

// C[n][col0 + m] (ldc) = sum_k A[n][k] (lda) W[m][k] (ldw) + bias[m].
struct LinearParams {
    const float* a; int lda;
    const float* w; int ldw;
    const float* bias;
    long n; int m, k;
    int mode;             // 0 store, 1 GELU store, 2 gated residual into c, 3 Euler: c += dt * val, 4 store * scale,
                          // 5 accumulate (c += val), 6 store val AND GELU(val) (second copy at c2: training tape),
                          // 7 store val * gelu'(c2[..]) (c2 = the taped pre-activation, read only: backward of the GELU)
    int wtrans;           // 1: the weight operand is stored [k][m] (ldw = row stride): y = x W, used for dX = dY W
    float* c; int ldc; int col0;
    ModMap mm; int gate_chunk; int gated;   // mode 2
    float scalar;                           // mode 3: dt; mode 4: scale
    float* c2;                              // mode 6: GELU output
    // column segments (k16_linear_fast only; 0 = off): output columns [j seg_cols, (j + 1) seg_cols) are the layer
    // (w_seg[j], bias_seg[j]) times scale_seg[j] -- q, k and v projections of one LayerNorm output as ONE pass over it
    int seg_cols;
    const float* w_seg[3];
    const float* bias_seg[3];
    float scale_seg[3];
    // k16_linear_wdma only: the weight(s) as a bf16 stream of 1 KiB MFMA fragments in the kernel's consumption order
    // (launch16_pack_wstream, k_wide16.hip), brought in by LDS-DMA; nullptr: the weight is read as fp32 rows
    const unsigned char* wpack;
    // 1 (bf16-operand training mode): GELU and its derivative (modes 1, 6, 7) through the logistic-polynomial Phi of
    // common.h gelu_erf (abs. error 5e-6, 1/400 of the bf16 rounding their results get as GEMM operands) instead of erff / expf:
    // the exact forms cost ~100 us (mode 6) / ~170 us (mode 7) of VALU time per 64 000 x 1536 launch
    int fast_gelu;
    // storage of two GEMM-only tensors in the bf16-operand mode: a_bf16: the token operand is bf16 rows (k16_linear_wdma<true>
    // only); c2_bf16: the GELU output of mode 6 is written as bf16 (same element indexing as c)
    int a_bf16, c2_bf16;
    // (round 6) the result of modes 7 / 17 (d pre = d hid * gelu'(pre)) is written as bf16 rows (same element indexing as c): it is
    // only ever the token operand of fc1's dX product and dY of fc1's weight gradient
    int c_bf16;
};

// Phi(x) and x of gelu_erf's fit (see common.h): Phi = 1 / (1 + exp2(x P(x^2)))
__device__ __forceinline__ float phi_cdf_fast(float x) {
    const float x2 = x * x;
    float p = -3.936969279e-06f;
    p = p * x2 + 1.012880530e-04f;
    p = p * x2 + 2.890509495e-04f;
    p = p * x2 - 1.051034182e-01f;
    p = p * x2 - 2.302086592e+00f;
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * p));
}

// epilogue shared by k32_linear and k16_linear: bias, then store / GELU / gated residual / Euler / scale / accumulate (LinearParams::mode)
// One instantiation per mode, the mode test outside the element loops: each (column, 32-row tile) is 16 independent
// elements whose read-modify-write loads (modes 2, 3, 5) are all issued before the first one is needed.
template <int MODE, int NT, int NU>
__device__ __forceinline__ void linear_epilogue_mode(const LinearParams& p, const f32x16 (&acc)[NT][NU], long row0, int colt, int wr,
                                                     int wc) {
    const int lane = lane_id();
    const int hh = lane >> 5;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int col = colt + wc * 32 * NU + u * 32 + (lane & 31);
        if (col >= p.m) continue;
        const int sg = p.seg_cols ? col / p.seg_cols : 0;
        const float* bp = p.seg_cols ? p.bias_seg[sg] : p.bias;
        const float bias = bp ? bp[col - sg * p.seg_cols] : 0.f;
        const float sscale = p.seg_cols ? p.scale_seg[sg] : 1.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const long rbase = row0 + wr * 32 * NT + t * 32;
            float old[16], g[16];
            if (MODE == 2 || MODE == 3 || MODE == 5 || MODE == 7 || MODE == 17) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long row = rbase + mfma_row(r, hh);
                    const long rc = row < p.n ? row : p.n - 1;
                    old[r] = (MODE == 7 || MODE == 17 ? p.c2 : p.c)[rc * p.ldc + p.col0 + col];   // mode 7: the taped pre-activation
                    g[r] = (MODE == 2 && p.gated) ? p.mm.mod[p.mm.row_off(rc) + p.gate_chunk * kC + col] : 1.0f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = rbase + mfma_row(r, hh);
                if (row >= p.n) continue;
                const float v = acc[t][u][r] + bias;
                float* dst = p.c + row * p.ldc + p.col0 + col;
                if (MODE == 0) {
                    *dst = p.seg_cols ? v * sscale : v;
                } else if (MODE == 1) {
                    *dst = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
                } else if (MODE == 2) {
                    *dst = old[r] + g[r] * v;
                } else if (MODE == 3) {
                    *dst = old[r] + p.scalar * v;
                } else if (MODE == 4) {
                    *dst = v * p.scalar;
                } else if (MODE == 5) {
                    *dst = old[r] + v;
                } else if (MODE == 7) {   // d pre = d hid * gelu'(pre), gelu'(x) = Phi(x) + x phi(x)   (layers.py:77-84 exact-erf GELU)
                    const float x = old[r];
                    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
                    const float pdf = 0.3989422804014327f * expf(-0.5f * x * x);
                    *dst = v * (cdf + x * pdf);
                } else if (MODE == 11) {
                    *dst = v * phi_cdf_fast(v);
                } else if (MODE == 16) {
                    *dst = v;
                    const float gl = v * phi_cdf_fast(v);
                    if (p.c2_bf16) reinterpret_cast<uint16_t*>(p.c2)[row * p.ldc + p.col0 + col] = (uint16_t)pack_bf16(gl, 0.f);
                    else p.c2[row * p.ldc + p.col0 + col] = gl;
                } else if (MODE == 17) {
                    const float x = old[r];
                    const float pdf = 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);
                    const float dpre = v * (phi_cdf_fast(x) + x * pdf);
                    if (p.c_bf16) reinterpret_cast<uint16_t*>(p.c)[row * p.ldc + p.col0 + col] = (uint16_t)pack_bf16(dpre, 0.f);
                    else *dst = dpre;
                } else {
                    *dst = v;
                    p.c2[row * p.ldc + p.col0 + col] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
                }
            }
        }
    }
}
// wave (wr, wc) of the workgroup holds rows 32 NT wr .., columns 32 NU wc .. of the tile at (row0, colt)
template <int NT, int NU>
__device__ __forceinline__ void linear_epilogue(const LinearParams& p, const f32x16 (&acc)[NT][NU], long row0, int colt, int wr, int wc) {
    switch (p.fast_gelu && (p.mode == 1 || p.mode == 6 || p.mode == 7) ? p.mode + 10 : p.mode) {
        case 11: linear_epilogue_mode<11, NT, NU>(p, acc, row0, colt, wr, wc); break;
        case 16: linear_epilogue_mode<16, NT, NU>(p, acc, row0, colt, wr, wc); break;
        case 17: linear_epilogue_mode<17, NT, NU>(p, acc, row0, colt, wr, wc); break;
        case 0: linear_epilogue_mode<0, NT, NU>(p, acc, row0, colt, wr, wc); break;
        case 1: linear_epilogue_mode<1, NT, NU>(p, acc, row0, colt, wr, wc); break;
        case 2: linear_epilogue_mode<2, NT, NU>(p, acc, row0, colt, wr, wc); break;
        case 3: linear_epilogue_mode<3, NT, NU>(p, acc, row0, colt, wr, wc); break;
        case 4: linear_epilogue_mode<4, NT, NU>(p, acc, row0, colt, wr, wc); break;
        case 5: linear_epilogue_mode<5, NT, NU>(p, acc, row0, colt, wr, wc); break;
        case 7: linear_epilogue_mode<7, NT, NU>(p, acc, row0, colt, wr, wc); break;
        default: linear_epilogue_mode<6, NT, NU>(p, acc, row0, colt, wr, wc); break;
    }
}


}  // namespace mdg
