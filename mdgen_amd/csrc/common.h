// common.h -- shared types/constants for the gfx950 kernels of libmdgen_amd.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "dev.h"

namespace mdg {

// ---- compile-time model geometry (reference defaults, parsing.py:79-90) ------------------
constexpr int kC = 384;        // embed_dim
constexpr int kH = 16;         // mha_heads
constexpr int kDH = 24;        // head_dim
constexpr int kF = 1536;       // ffn dim (4C)
constexpr int kIpaProj = 672;  // linear_q(128) | linear_kv(256) | linear_q_points(96) | linear_kv_points(192)
constexpr int kIpaFeat = 256;  // o(128) | o_pt.x(32) | o_pt.y(32) | o_pt.z(32) | |o_pt|(32)
// Attention operand fragments of one (seq, head, 32-position tile).  Q: k-step 0 (64 lanes x 16 B) + k-step 1 (64 x 8 B).
// K: both k-steps 64 x 16 B -- k-step 1 carries 4 features, the constant 1.0 twice (slots 4, 5) and 2 zeros.  V^T: [k-step 2]
// [key half 2][row d 0..24][16 B], row 24 all ones (it accumulates the softmax denominator).
constexpr int kFragQ = 1536, kFragK = 2048, kFragV = 1600;
constexpr int kFragBytes = 2048;  // allocation size of a fragment tile (the largest of the three)
constexpr int kMaxFlashTiles = 260;   // key tiles per sequence k_flash supports (len <= 8319)
constexpr int kRopeRow = 32;   // floats per position of the rotary table: [2 halves][cos 6 | pad 2 | sin 6 | pad 2]
constexpr int kPanel = 64;     // token rows per GEMM panel (workgroup)
constexpr int kIpaKT = 32;     // keys per LDS tile of the tiled IPA attention kernels
constexpr float kLog2e = 1.4426950408889634f;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// (seq, pos) -> token row of the [B][T][L] token block.  Used by both attention axes:
//   residue axis : seq = b*T + t, pos = l  -> token = seq*L + pos
//   temporal axis: seq = b*L + l, pos = t  -> token = b*T*L + pos*L + l
struct AxisMap {
    int nseq, len;
    int inner;          // seq = outer*inner + in
    int outer_stride;   // token stride of `outer`
    int inner_stride;   // token stride of `in`
    int pos_stride;     // token stride of `pos`
    __host__ __device__ inline long token(int seq, int pos) const {
        return (long)(seq / inner) * outer_stride + (long)(seq % inner) * inner_stride + (long)pos * pos_stride;
    }
    __host__ __device__ inline int ntile() const { return len / 32 + 1; }   // tiles covering len+1 keys
};

// Per-row modulation lookup: group g = token / tokens_per_group;
// vector = mod + (g / groups_per_step) * step_stride + (g % groups_per_step) * group_stride + chunk*kC
struct ModMap {
    const float* mod;
    int tokens_per_group;
    int groups_per_step;
    long step_stride;
    long group_stride;
    __device__ inline long row_off(long token) const {
        long g = token / tokens_per_group;
        return (g / groups_per_step) * step_stride + (g % groups_per_step) * group_stride;
    }
};

// key-padding mask lookup: mask[token % period]
struct MaskMap {
    const float* mask;
    long period;
    // period == 0: mask is indexed by the token itself (trunk); else token % period (IPA stack: (b,l))
    __device__ inline float at(long token) const {
        return mask[period ? (long)((unsigned)token % (unsigned)period) : token];
    }
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ float bf16_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// v + (v moved by a DPP lane pattern inside each row of 16 lanes): one full-rate VALU op, no LDS round trip.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// All-lanes sum of a wave64: 4 DPP steps inside rows of 16 (quad xor 1, quad xor 2, half-row mirror, row mirror),
// one cross-row exchange (lane ^ 16) and one half-wave swap.
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);   // row_half_mirror
    v = dpp_add<0x140>(v);   // row_mirror
    v += __shfl_xor(v, 16, 64);
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// A zero the compiler cannot constant-fold.  hipcc (ROCm 7.2) folds a zero-initialised accumulator
// into the MFMA's inline-constant C operand and then allocates the 16-register destination ON TOP of
// the A/B source registers (no early-clobber on v_mfma_f32_32x32x16_bf16 ... , 0).  With two waves per
// SIMD that corrupts results intermittently.  Accumulators are therefore always zeroed with real
// v_mov instructions, which forces the tied (dst == C) form whose destination never overlaps A/B;
// mdgen_amd/build.py scans the generated ISA and fails the build if such an overlap is emitted.
__device__ __forceinline__ float opaque_zero() {
    float z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}

// gelu(x) = x * Phi(x), Phi = standard normal CDF = 0.5 * (1 + erf(x / sqrt 2))   (layers.py:77-84, exact-erf GELU).
// Phi is evaluated as a logistic function of an odd polynomial, Phi(x) = 1 / (1 + exp(-x * P(x^2))), with the
// degree-4 P fitted to the exact CDF on |x| <= 8 (beyond that Phi is 0 or 1 to fp32 precision, hence the clamp
// of the polynomial's argument).  Maximum absolute error of gelu over all x, evaluated in fp32: 6.2e-6
// (checked by scripts/fit_gelu.py), two orders of magnitude below the bf16 rounding applied to the result right after.
// 11 VALU ops per element (the MLP kernel's GELU phase is VALU-bound): libm erff ~40, A&S 7.1.26 rational ~17.
// Coefficients carry the factor -log2(e) so that the exponential is a bare v_exp_f32.
__device__ __forceinline__ float gelu_erf(float x) {
    // No clamp of x: P(x^2) < 0 everywhere (asserted on a grid by scripts/fit_gelu.py), so for large |x| the exponent
    // x P(x^2) runs off to -inf (x > 0: e = 0, result x) or +inf (x < 0: e = inf, rcp = 0, result -0) on its own.
    const float xc = x;
    const float x2 = xc * xc;
    float p = -3.936969279e-06f;
    p = p * x2 + 1.012880530e-04f;
    p = p * x2 + 2.890509495e-04f;
    p = p * x2 - 1.051034182e-01f;
    p = p * x2 - 2.302086592e+00f;
    const float e = __builtin_amdgcn_exp2f(xc * p);         // exp(-x P(x^2)); +inf for very negative x -> result -0
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// D-layout row of accumulator register r for lane-half h of a 32x32 MFMA tile
__host__ __device__ constexpr int mfma_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

}  // namespace mdg
