// k_fp32.hip -- the fp32-operand ("reference tolerance") form of the denoiser's GEMM family and attention.
//
// The product path feeds the matrix cores bf16 operands (rel-L2 4-6e-3 per network evaluation).  The reference
// itself runs fp32 throughout (mha.py:19-23 softmax in fp32, rigid_utils.py:318-322), and BASELINE.md section 3 gates
// fp32 kernels at rel-L2 <= 1e-5 / 1e-3 A.  This file is the same arithmetic with fp32 operands, selected by the
// context option "precision" = 32: plain row-major fp32 buffers, one kernel per reference op group,
//   k32_ln_mod      LayerNorm (no affine, eps 1e-6) + adaLN modulate, or affine LayerNorm (eps 1e-5)
//                   (layers.py:14-15, latent_model.py:373)
//   k32_linear      y = x W^T + b on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate: an fmaf chain),
//                   with the epilogues the layers need: store, exact-erf GELU (layers.py:77-84), gated residual
//                   h += gate * y (latent_model.py:462,476,481), Euler update / velocity (integrators.py:106)
//   k32_rope        q * dh^-1/2 and rotate-half RoPE of q, k in place (mha.py:260-263, 356-357)
//   k32_attn        softmax(q k^T [+ bias key] with key padding) v, streaming over key tiles (mha.py:265-268, 359-396)
// It is a tolerance mode, not a fast path: ~30x slower than the bf16 kernels at cfg-2 (157 TFLOP/s fp32 MFMA peak vs
// 2.5 PFLOP/s bf16, an unfused structure, and a simple attention kernel).  It shares everything that is fp32 already: time embedding, adaLN table,
// token embedding, IPA point attention, SE(3) kernels.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "kernels.h"
#include "linear.h"

namespace mdg {
bool launch16_linear_wide(const LinearParams& p, hipStream_t s);   // k_wide16.hip; false: shape not eligible, nothing launched
}

namespace mdg {

// one wave per row; rows in natural token order
// keep != nullptr: the input rows are also copied there (the training tape's h_in: the row is in registers anyway)
// y16 != nullptr: y is written THERE, rounded to bf16 (training step, bf16-operand mode: the LayerNorm output of the trunk is only ever
// a GEMM operand -- the token rows of the q | k | v / fc1 products and X of their weight gradients, which round it to bf16 on their
// way into LDS anyway: the same values enter the MFMAs, half the bytes cross HBM three times)
__device__ __forceinline__ unsigned short bf16_bits(float v) { return (unsigned short)(pack_bf16(v, 0.f) & 0xffffu); }
__global__ __launch_bounds__(256) void k32_ln_mod(const float* __restrict__ x, long nrows, ModMap mm, int shift_chunk,
                                                  int scale_chunk, int affine, float eps, float* __restrict__ y,
                                                  float* __restrict__ keep, unsigned short* __restrict__ y16) {
    const long row = (long)blockIdx.x * 4 + wave_id();
    if (row >= nrows) return;
    const int lane = lane_id();
    const float* xr = x + row * kC;
    float v[6];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        v[i] = xr[lane + 64 * i];
        s += v[i];
        if (keep) keep[row * kC + lane + 64 * i] = v[i];
    }
    const float mean = wave_sum(s) * (1.0f / kC);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        v[i] -= mean;
        q += v[i] * v[i];
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / kC) + eps);
    const float* mod = mm.mod + (affine ? 0 : mm.row_off(row));
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c = lane + 64 * i;
        const float sc = mod[scale_chunk * kC + c], sh = mod[shift_chunk * kC + c];
        const float o = v[i] * rstd * (affine ? sc : 1.0f + sc) + sh;
        if (y16) y16[row * kC + c] = bf16_bits(o);
        else y[row * kC + c] = o;
    }
}

// The same with the PREVIOUS sub-layer's gated residual update in front (training step, trunk): x = xp + gate * up is formed in
// registers, written to `keep` (the tape's copy of the residual stream, which is also where the next update reads it) and
// normalised -- the stream is neither updated in place nor read back (one launch and two passes over the stream less per
// sub-layer boundary than k32_gated_add + k32_ln_mod; same arithmetic per element).
__global__ __launch_bounds__(256) void k32_gate_ln_mod(const float* __restrict__ xp, const float* __restrict__ up, long nrows, ModMap gm,
                                                       int gate_chunk, ModMap mm, int shift_chunk, int scale_chunk, float eps,
                                                       float* __restrict__ y, float* __restrict__ keep, unsigned short* __restrict__ y16) {
    const long row = (long)blockIdx.x * 4 + wave_id();
    if (row >= nrows) return;
    const int lane = lane_id();
    const float* gate = gm.mod + gm.row_off(row) + gate_chunk * kC;
    float v[6];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c = lane + 64 * i;
        v[i] = xp[row * kC + c] + gate[c] * up[row * kC + c];
        s += v[i];
        keep[row * kC + c] = v[i];
    }
    const float mean = wave_sum(s) * (1.0f / kC);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        v[i] -= mean;
        q += v[i] * v[i];
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / kC) + eps);
    const float* mod = mm.mod + mm.row_off(row);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c = lane + 64 * i;
        const float o = v[i] * rstd * (1.0f + mod[scale_chunk * kC + c]) + mod[shift_chunk * kC + c];
        if (y16) y16[row * kC + c] = bf16_bits(o);
        else y[row * kC + c] = o;
    }
}

// Operand precision of the training step's linear layers / weight gradients: 0 = fp32 products (k32_linear / k32_dw on
// v_mfma_f32_32x32x2_f32, the exact mode), 1 = k16_linear / k16_dw: operands rounded
// to bf16 on their way into LDS and multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- what the reference
// trains with (train.py:13 `torch.set_float32_matmul_precision('medium')` = bf16-class products, fp32 accumulate, fp32
// master weights).  Set by mdgen_train_forward_backward from the context option "train_precision" for the duration of the
// call; the sampler's fp32 tolerance mode always runs with 0.  Per THREAD: two contexts training on two host threads do not
// see each other's mode.
thread_local int g_k32_bf16_operands = 0;

// A launcher that is handed a shape its caller should have ruled out (bf16-stored operands outside the streamed kernels)
// launches nothing and leaves a message here; the C-ABI entry points turn it into an error return (api.hip LAUNCHCHK).
thread_local const char* g_k32_launch_error = nullptr;
const char* k32_take_launch_error() {
    const char* m = g_k32_launch_error;
    g_k32_launch_error = nullptr;
    return m;
}

// Workgroup tile 128 rows x 128 columns, wave tile 64 x 64 (2 x 2 MFMA tiles: every LDS operand read feeds two MFMAs),
// k in steps of 16; the next k-step's global loads are issued before the current one's MFMAs (register prefetch), so a
// k-step costs max(MFMA, memory) instead of their sum.  Loads are unconditional with clamped indices.
__global__ __launch_bounds__(256) void k32_linear(const LinearParams p) {
    constexpr int BK = 16, LD = BK + 1, TM = 128;
    __shared__ float As[TM * LD];
    __shared__ float Ws[TM * LD];
    const int lane = lane_id(), w = wave_id();
    // grid = (column tiles, row tiles): the workgroups that share a 128-row slice of A are dispatched back to back
    const long row0 = (long)blockIdx.y * TM;
    const int colt = blockIdx.x * TM;
    const int wr = w >> 1, wc = w & 1;   // wave -> 64 x 64 sub-tile
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = opaque_zero();   // a real zero tuple, not the inline constant (common.h)
    // staging: thread -> rows lr, lr + 64 and four consecutive k of the 16 (one 16-byte load each when the operand
    // allows it); a transposed weight operand ([k][m], the dX = dY W products) is staged k-major instead: thread ->
    // k index tid / 16 and eight consecutive columns, so that its loads are contiguous too
    const int lr = threadIdx.x >> 2, lk = (threadIdx.x & 3) * 4;
    const int tk = threadIdx.x >> 4, tc = (threadIdx.x & 15) * 8;
    long ar[2];
    int wrow[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        ar[h] = row0 + lr + 64 * h < p.n ? row0 + lr + 64 * h : p.n - 1;
        wrow[h] = colt + lr + 64 * h < p.m ? colt + lr + 64 * h : p.m - 1;
    }
    const bool veca = ((p.lda | p.k) & 3) == 0 && ((unsigned long long)p.a & 15) == 0;
    const bool vecw = p.wtrans ? ((p.ldw | p.m) & 7) == 0 && ((unsigned long long)p.w & 15) == 0
                               : ((p.ldw | p.k) & 3) == 0 && ((unsigned long long)p.w & 15) == 0;
    float av[2][4], wv[2][4];
    auto fetch = [&](int k0) {
        if (veca) {
            const int kc = k0 + lk < p.k ? k0 + lk : 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(p.a + ar[h] * p.lda + kc);
#pragma unroll
                for (int j = 0; j < 4; ++j) av[h][j] = k0 + lk < p.k ? v[j] : 0.f;
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kk = k0 + lk + j, kc = kk < p.k ? kk : p.k - 1;
                    const float a = p.a[ar[h] * p.lda + kc];
                    av[h][j] = kk < p.k ? a : 0.f;
                }
        }
        if (vecw && p.wtrans) {          // wv[h][j] = W^T[k0 + tk][colt + tc + 4 h + j]
            const int kc = k0 + tk < p.k ? k0 + tk : 0;
            const int cc = colt + tc < p.m ? colt + tc : 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(p.w + (long)kc * p.ldw + cc + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) wv[h][j] = (k0 + tk < p.k && colt + tc < p.m) ? v[j] : 0.f;
            }
        } else if (vecw) {
            const int kc = k0 + lk < p.k ? k0 + lk : 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(p.w + (long)wrow[h] * p.ldw + kc);
#pragma unroll
                for (int j = 0; j < 4; ++j) wv[h][j] = k0 + lk < p.k ? v[j] : 0.f;
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kk = k0 + lk + j, kc = kk < p.k ? kk : p.k - 1;
                    const float b = p.wtrans ? p.w[(long)kc * p.ldw + wrow[h]] : p.w[(long)wrow[h] * p.ldw + kc];
                    wv[h][j] = kk < p.k ? b : 0.f;
                }
        }
    };
    const bool wkmajor = vecw && p.wtrans;
    fetch(0);
    const int i = lane & 31, kh = lane >> 5;
    for (int k0 = 0; k0 < p.k; k0 += BK) {
        __syncthreads();   // previous tile consumed
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                As[(lr + 64 * h) * LD + lk + j] = av[h][j];
                if (wkmajor) Ws[(tc + 4 * h + j) * LD + tk] = wv[h][j];
                else Ws[(lr + 64 * h) * LD + lk + j] = wv[h][j];
            }
        __syncthreads();
        if (k0 + BK < p.k) fetch(k0 + BK);   // in flight under the MFMAs below
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = As[(wr * 64 + t * 32 + i) * LD + kk + kh];
                b[t] = Ws[(wc * 64 + t * 32 + i) * LD + kk + kh];
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[u], acc[t][u], 0, 0, 0);
        }
    }
    linear_epilogue(p, acc, row0, colt, wr, wc);
}

// The bf16-operand linear layer of the training step (option train_precision = 16), as a kernel of its own: same operands,
// modes and epilogue as k32_linear, but built for the bf16 MFMA's 16x arithmetic rate, i.e. around memory and barriers:
// k in steps of 64 (four MFMA k-steps per LDS tile: 16 MFMAs per wave between barriers), two LDS buffers and ONE barrier per
// step (the step's MFMAs read buffer i while the registers that were prefetched during the previous step's MFMAs go to
// buffer i ^ 1), operands rounded to bf16 on their way into LDS, 144-byte LDS rows (16 consecutive rows fall on 16 distinct
// 16-byte slots of the 256-byte bank row), fragment reads that ARE the MFMA operands.
__global__ __launch_bounds__(256, 2) void k16_linear(const LinearParams p) {
    constexpr int BK = 64, TM = 128, ROWB = 144, KPT = 16, NTK = 4;
    __shared__ __attribute__((aligned(16))) unsigned char Ab[2][TM * ROWB];
    __shared__ __attribute__((aligned(16))) unsigned char Wb[2][TM * ROWB];
    const int lane = lane_id(), w = wave_id();
    const long row0 = (long)blockIdx.y * TM;
    const int colt = blockIdx.x * TM;
    const int wr = w >> 1, wc = w & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = opaque_zero();
    const int lr = threadIdx.x >> 2, lk = (threadIdx.x & 3) * KPT;      // row-major staging: rows lr, lr + 64; 16 consecutive k
    const int tk = threadIdx.x >> 4, tc = (threadIdx.x & 15) * 8;       // k-major staging (wtrans): k rows tk + 16 z, 8 columns
    long ar[2];
    int wrow[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        ar[h] = row0 + lr + 64 * h < p.n ? row0 + lr + 64 * h : p.n - 1;
        wrow[h] = colt + lr + 64 * h < p.m ? colt + lr + 64 * h : p.m - 1;
    }
    const bool veca = ((p.lda | p.k) & 3) == 0 && ((unsigned long long)p.a & 15) == 0;
    const bool vecw = p.wtrans ? ((p.ldw | p.m) & 7) == 0 && ((unsigned long long)p.w & 15) == 0
                               : ((p.ldw | p.k) & 3) == 0 && ((unsigned long long)p.w & 15) == 0;
    const bool wkmajor = vecw && p.wtrans;
    float av[2][KPT], wv[2 * NTK][KPT];   // wv: row-major staging uses [2][16] of it, k-major [2 z + h][4]
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < KPT / 4; ++q) {
            const int kb = k0 + lk + 4 * q;
            if (veca) {
                const int kc = kb < p.k ? kb : 0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(p.a + ar[h] * p.lda + kc);
#pragma unroll
                    for (int j = 0; j < 4; ++j) av[h][4 * q + j] = kb < p.k ? v[j] : 0.f;
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int kk = kb + j, kc = kk < p.k ? kk : p.k - 1;
                        const float a = p.a[ar[h] * p.lda + kc];
                        av[h][4 * q + j] = kk < p.k ? a : 0.f;
                    }
            }
        }
        if (wkmajor) {          // wv[2 z + h][j] = W^T[k0 + tk + 16 z][colt + tc + 4 h + j]
#pragma unroll
            for (int z = 0; z < NTK; ++z) {
                const int kr = k0 + tk + 16 * z;
                const int kc = kr < p.k ? kr : 0;
                const int cc = colt + tc < p.m ? colt + tc : 0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(p.w + (long)kc * p.ldw + cc + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) wv[2 * z + h][j] = (kr < p.k && colt + tc < p.m) ? v[j] : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < KPT / 4; ++q) {
                const int kb = k0 + lk + 4 * q;
                if (vecw) {
                    const int kc = kb < p.k ? kb : 0;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(p.w + (long)wrow[h] * p.ldw + kc);
#pragma unroll
                        for (int j = 0; j < 4; ++j) wv[h][4 * q + j] = kb < p.k ? v[j] : 0.f;
                    }
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int kk = kb + j, kc = kk < p.k ? kk : p.k - 1;
                            const float b = p.wtrans ? p.w[(long)kc * p.ldw + wrow[h]] : p.w[(long)wrow[h] * p.ldw + kc];
                            wv[h][4 * q + j] = kk < p.k ? b : 0.f;
                        }
                }
            }
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                *reinterpret_cast<u32x4*>(&Ab[buf][(lr + 64 * h) * ROWB + lk * 2 + 16 * q]) =
                    u32x4{pack_bf16(av[h][8 * q], av[h][8 * q + 1]), pack_bf16(av[h][8 * q + 2], av[h][8 * q + 3]),
                          pack_bf16(av[h][8 * q + 4], av[h][8 * q + 5]), pack_bf16(av[h][8 * q + 6], av[h][8 * q + 7])};
                if (!wkmajor)
                    *reinterpret_cast<u32x4*>(&Wb[buf][(lr + 64 * h) * ROWB + lk * 2 + 16 * q]) =
                        u32x4{pack_bf16(wv[h][8 * q], wv[h][8 * q + 1]), pack_bf16(wv[h][8 * q + 2], wv[h][8 * q + 3]),
                              pack_bf16(wv[h][8 * q + 4], wv[h][8 * q + 5]), pack_bf16(wv[h][8 * q + 6], wv[h][8 * q + 7])};
            }
        if (wkmajor) {
#pragma unroll
            for (int z = 0; z < NTK; ++z)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<uint16_t*>(&Wb[buf][(tc + 4 * h + j) * ROWB + (tk + 16 * z) * 2]) = (uint16_t)pack_bf16(wv[2 * z + h][j], 0.f);
        }
    };
    const int i = lane & 31, kh = lane >> 5;
    fetch(0);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < p.k; k0 += BK, buf ^= 1) {
        const bool more = k0 + BK < p.k;
        if (more) fetch(k0 + BK);   // in flight under the MFMAs below
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = *reinterpret_cast<const bf16x8*>(&Ab[buf][(wr * 64 + t * 32 + i) * ROWB + ks * 32 + kh * 16]);
                b[t] = *reinterpret_cast<const bf16x8*>(&Wb[buf][(wc * 64 + t * 32 + i) * ROWB + ks * 32 + kh * 16]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b[u], acc[t][u], 0, 0, 0);
        }
        if (more) stage(buf ^ 1);   // buffer buf ^ 1 was consumed before the barrier that closed the previous step
        __syncthreads();
    }
    linear_epilogue(p, acc, row0, colt, wr, wc);
}

// k16_linear for the shapes the trunk actually has (k a multiple of 64, 16-byte aligned rows, weight stored [m][k]): no
// bounds checks or alignment branches in the loop, so that a k-step is sixteen 16-byte loads issued together, the sixteen
// MFMAs of the previous step's tile, and the conversion of the loaded registers into the other LDS buffer.  (In the
// general kernel every one of those loads sits in its own basic block behind a run-time `vectorisable?` test and is waited
// for on the spot: measured 185 us for a 64 000 x 384 x 384 layer, 1 TB/s of activation traffic.)  Each load instruction
// of a wave covers FOUR WHOLE 256-byte row segments (lane -> 16-byte piece tid & 15 of row tid >> 4), i.e. eight full
// cache lines, instead of a quarter of each of 32 lines.  Tile order: the column tiles of one 128-row slice run back to
// back on ONE XCD (workgroup i goes to XCD i % 8), whose L2 then serves the slice's re-reads.
__global__ __launch_bounds__(256, 2) void k16_linear_fast(const LinearParams p, int nrt, int nct) {
    constexpr int BK = 64, TM = 128, ROWB = 144;
    __shared__ __attribute__((aligned(16))) unsigned char Ab[2][TM * ROWB];
    __shared__ __attribute__((aligned(16))) unsigned char Wb[2][TM * ROWB];
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int rt = (slot / nct) * 8 + xcd, ct = slot % nct;
    if (rt >= nrt) return;
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const long row0 = (long)rt * TM;
    const int colt = ct * TM;
    const int wr = w >> 1, wc = w & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = opaque_zero();
    const int r0 = tid >> 4, piece = tid & 15;        // rows r0 + 16 q, k0 + 4 piece .. + 3
    const int sg = p.seg_cols ? colt / p.seg_cols : 0;          // segments are whole numbers of column tiles (launcher)
    const float* wbase = p.seg_cols ? p.w_seg[sg] : p.w;
    const int ccol = colt - sg * p.seg_cols, mseg = p.seg_cols ? p.seg_cols : p.m;
    const float* ap[8];
    const float* wp[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const long r = row0 + r0 + 16 * q < p.n ? row0 + r0 + 16 * q : p.n - 1;   // rows / columns past the end: clamped
        ap[q] = p.a + r * p.lda + 4 * piece;                                      // loads, results never stored
        const int c = ccol + r0 + 16 * q < mseg ? ccol + r0 + 16 * q : mseg - 1;
        wp[q] = wbase + (long)c * p.ldw + 4 * piece;
    }
    f32x4 av[8], wv[8];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) av[q] = *reinterpret_cast<const f32x4*>(ap[q] + k0);
#pragma unroll
        for (int q = 0; q < 8; ++q) wv[q] = *reinterpret_cast<const f32x4*>(wp[q] + k0);
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            *reinterpret_cast<u32x2*>(&Ab[buf][(r0 + 16 * q) * ROWB + piece * 8]) = u32x2{pack_bf16(av[q][0], av[q][1]), pack_bf16(av[q][2], av[q][3])};
            *reinterpret_cast<u32x2*>(&Wb[buf][(r0 + 16 * q) * ROWB + piece * 8]) = u32x2{pack_bf16(wv[q][0], wv[q][1]), pack_bf16(wv[q][2], wv[q][3])};
        }
    };
    const int i = lane & 31, kh = lane >> 5;
    fetch(0);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < p.k; k0 += BK, buf ^= 1) {
        const bool more = k0 + BK < p.k;
        if (more) fetch(k0 + BK);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = *reinterpret_cast<const bf16x8*>(&Ab[buf][(wr * 64 + t * 32 + i) * ROWB + ks * 32 + kh * 16]);
                b[t] = *reinterpret_cast<const bf16x8*>(&Wb[buf][(wc * 64 + t * 32 + i) * ROWB + ks * 32 + kh * 16]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b[u], acc[t][u], 0, 0, 0);
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
    }
    linear_epilogue(p, acc, row0, colt, wr, wc);
}

// Launches of a few hundred rows (the IPA stack works on B * L rows: 86 linear layers per training step at cfg-5's size,
// 6 .. 24 workgroups each): with 128 x 128 tiles they are a handful of workgroups walking k in six dependent load -> LDS ->
// barrier steps (~25 us per launch).  Here ONE WAVE owns a 32 x 32 output tile and the whole contraction: both operands come
// straight from global memory in MFMA fragment shape (lane = row / column, eight consecutive k = two 16-byte loads), 128 of k
// at a time (32 loads in flight per lane), rounded to bf16 in registers; no LDS, no barrier, 96 waves for 256 x 384.
__global__ __launch_bounds__(64) void k16_linear_small(const LinearParams p) {
    const int lane = threadIdx.x, i = lane & 31, kh = lane >> 5;
    const long row0 = (long)blockIdx.y * 32;
    const int colt = blockIdx.x * 32;
    const int sg = p.seg_cols ? colt / p.seg_cols : 0;
    const float* wbase = p.seg_cols ? p.w_seg[sg] : p.w;
    const int ccol = colt - sg * p.seg_cols, mseg = p.seg_cols ? p.seg_cols : p.m;
    const long ar = row0 + i < p.n ? row0 + i : p.n - 1;           // past the end: clamped loads, results never stored
    const int wc = ccol + i < mseg ? ccol + i : mseg - 1;
    const float* ap = p.a + ar * p.lda + 8 * kh;
    const float* wp = wbase + (long)wc * p.ldw + 8 * kh;
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = opaque_zero();
    auto chunk = [&](int k0, auto nks) {
        constexpr int NKS = decltype(nks)::value;
        f32x4 a[NKS][2], b[NKS][2];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                a[ks][h] = *reinterpret_cast<const f32x4*>(ap + k0 + 16 * ks + 4 * h);
                b[ks][h] = *reinterpret_cast<const f32x4*>(wp + k0 + 16 * ks + 4 * h);
            }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const bf16x8 af = __builtin_bit_cast(bf16x8, u32x4{pack_bf16(a[ks][0][0], a[ks][0][1]), pack_bf16(a[ks][0][2], a[ks][0][3]),
                                                               pack_bf16(a[ks][1][0], a[ks][1][1]), pack_bf16(a[ks][1][2], a[ks][1][3])});
            const bf16x8 bf = __builtin_bit_cast(bf16x8, u32x4{pack_bf16(b[ks][0][0], b[ks][0][1]), pack_bf16(b[ks][0][2], b[ks][0][3]),
                                                               pack_bf16(b[ks][1][0], b[ks][1][1]), pack_bf16(b[ks][1][2], b[ks][1][3])});
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[0][0], 0, 0, 0);
        }
    };
    int k0 = 0;
    for (; k0 + 128 <= p.k; k0 += 128) chunk(k0, std::integral_constant<int, 8>{});
    if (k0 < p.k) chunk(k0, std::integral_constant<int, 4>{});     // k is a multiple of 64 (launcher)
    linear_epilogue(p, acc, row0, colt, 0, 0);
}

// dst[c][r] (row stride ldd) = src[r][c]: the weight of a dX = dY W product, turned once per use so that the product runs through the same
// [m][k] kernel as the forward layer (64 x 64 tiles through LDS; a 384 x 384 weight is 36 workgroups, ~3 us).
__global__ __launch_bounds__(256) void k32_transpose(const float* __restrict__ src, int rows, int cols, float* __restrict__ dst,
                                                     int ldd) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty + 4 * i, c = c0 + tx;
        tile[ty + 4 * i][tx] = (r < rows && c < cols) ? src[(long)r * cols + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i, r = r0 + tx;
        if (c < cols && r < rows) dst[(long)c * ldd + r] = tile[tx][ty + 4 * i];
    }
}
void launch32_transpose(const float* src, int rows, int cols, float* dst, hipStream_t s, int ldd) {
    hipLaunchKernelGGL(k32_transpose, dim3((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64)), dim3(256), 0, s, src, rows, cols,
                       dst, ldd ? ldd : rows);
}

// q (pre-scaled by the caller's linear, mode 4) and k: rotate-half RoPE in place.  buf[token][ld]: q at col 0, k at col
// 384 (v at 768 untouched).  pos = (token / pos_div) % pos_mod.  One thread per (token, head, pair i < 12).
__global__ void k32_rope(float* __restrict__ buf, long ntok, int ld, long pos_div, int pos_mod,
                         const float* __restrict__ inv_freq) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = ntok * kH * 12 * 2;
    if (idx >= total) return;
    const int i = (int)(idx % 12);
    const int hd = (int)((idx / 12) % kH);
    const int which = (int)((idx / (12 * kH)) & 1);
    const long token = idx / (12 * kH * 2);
    const int pos = (int)((token / pos_div) % pos_mod);
    const float ang = (float)pos * inv_freq[i];
    const float c = cosf(ang), s = sinf(ang);
    float* v = buf + token * ld + which * kC + hd * kDH;
    const float x1 = v[i], x2 = v[i + 12];
    v[i] = x1 * c - x2 * s;          // x * cos + rotate_half(x) * sin, rotate_half(x) = [-x2, x1]
    v[i + 12] = x2 * c + x1 * s;
}

// Attention of one axis, fp32: one workgroup per (sequence, head, 256-query block), one thread per query, keys
// staged through LDS 64 at a time (broadcast reads), online softmax renormalised once per tile.  The learned bias
// key / value (rotated at position len) is the last key; padded keys are excluded (mha.py:367-373 -inf fill).
__global__ __launch_bounds__(256) void k32_attn(const float* __restrict__ qkv, int ld, AxisMap ax, MaskMap mk,
                                                const float* __restrict__ bias_k, const float* __restrict__ bias_v,
                                                const float* __restrict__ inv_freq, float* __restrict__ out,
                                                float* __restrict__ lse_out) {
    constexpr int KT = 64;
    __shared__ __attribute__((aligned(16))) float sk[KT][kDH];
    __shared__ __attribute__((aligned(16))) float sv[KT][kDH];
    __shared__ float sm[KT];
    const int len = ax.len;
    const int nqb = (len + 255) / 256;
    const int qb = blockIdx.x % nqb;
    const int hd = (blockIdx.x / nqb) % kH;
    const int seq = blockIdx.x / (nqb * kH);
    const int tid = threadIdx.x;
    const int qi = qb * 256 + tid;
    const bool qok = qi < len;
    const long qtok = ax.token(seq, qok ? qi : len - 1);
    float q[kDH], o[kDH];
#pragma unroll
    for (int d = 0; d < kDH; ++d) {
        q[d] = qkv[qtok * ld + hd * kDH + d];
        o[d] = 0.f;
    }
    float mrun = -3.0e38f, den = 0.f;
    for (int j0 = 0; j0 < len + 1; j0 += KT) {
        __syncthreads();
        for (int e = tid; e < KT * kDH; e += 256) {   // stage: key row j0 + e / 24, feature e % 24
            const int jj = e / kDH, d = e % kDH;
            const int j = j0 + jj;
            float kvv = 0.f, vvv = 0.f;
            if (j < len) {
                const long t = ax.token(seq, j);
                kvv = qkv[t * ld + kC + hd * kDH + d];
                vvv = qkv[t * ld + 2 * kC + hd * kDH + d];
            } else if (j == len) {   // bias key: rotate-half RoPE at position len
                const int i = d % 12;
                const float ang = (float)len * inv_freq[i];
                const float c = cosf(ang), s = sinf(ang);
                const float x1 = bias_k[hd * kDH + i], x2 = bias_k[hd * kDH + i + 12];
                kvv = d < 12 ? x1 * c - x2 * s : x2 * c + x1 * s;
                vvv = bias_v[hd * kDH + d];
            }
            sk[jj][d] = kvv;
            sv[jj][d] = vvv;
        }
        if (tid < KT) {
            const int j = j0 + tid;
            sm[tid] = j < len ? (mk.at(ax.token(seq, j)) != 0.f ? 1.f : 0.f) : (j == len ? 1.f : 0.f);
        }
        __syncthreads();
        float lg[KT];
        float tmax = -3.0e38f;
#pragma unroll
        for (int jj = 0; jj < KT; ++jj) {
            float dot = 0.f;
#pragma unroll
            for (int d = 0; d < kDH; d += 4) {
                const f32x4 kv = *reinterpret_cast<const f32x4*>(&sk[jj][d]);
                dot += q[d] * kv[0] + q[d + 1] * kv[1] + q[d + 2] * kv[2] + q[d + 3] * kv[3];
            }
            lg[jj] = sm[jj] != 0.f ? dot : -3.0e38f;
            tmax = fmaxf(tmax, lg[jj]);
        }
        const float mnew = fmaxf(mrun, tmax);
        const float alpha = expf(mrun - mnew);
        mrun = mnew;
        den *= alpha;
#pragma unroll
        for (int d = 0; d < kDH; ++d) o[d] *= alpha;
#pragma unroll
        for (int jj = 0; jj < KT; ++jj) {
            const float pw = lg[jj] > -1.0e38f ? expf(lg[jj] - mnew) : 0.f;
            den += pw;
#pragma unroll
            for (int d = 0; d < kDH; d += 4) {
                const f32x4 vv = *reinterpret_cast<const f32x4*>(&sv[jj][d]);
                o[d] += pw * vv[0]; o[d + 1] += pw * vv[1]; o[d + 2] += pw * vv[2]; o[d + 3] += pw * vv[3];
            }
        }
    }
    if (!qok) return;
    const float inv = 1.0f / den;
#pragma unroll
    for (int d = 0; d < kDH; ++d) out[qtok * kC + hd * kDH + d] = o[d] * inv;
    if (lse_out) lse_out[qtok * kH + hd] = mrun + logf(den);   // training tape: the backward pass starts from it
}

// x += dt * v (Euler) is mode 3 of k32_linear; bf16 IPA features -> fp32 is avoided by the fp32 feature output of the
// IPA attention kernels (IpaAttnParams::feat32).

void launch32_ln_mod(const float* x, long nrows, const ModMap& mm, int shift_chunk, int scale_chunk, int affine, float eps,
                     float* y, hipStream_t s, float* keep, bool y_bf16) {
    hipLaunchKernelGGL(k32_ln_mod, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, s, x, nrows, mm, shift_chunk, scale_chunk,
                       affine, eps, y, keep, y_bf16 ? reinterpret_cast<unsigned short*>(y) : nullptr);
}
void launch32_gate_ln_mod(const float* xp, const float* up, long nrows, const ModMap& gm, int gate_chunk, const ModMap& mm,
                          int shift_chunk, int scale_chunk, float eps, float* y, float* keep, hipStream_t s, bool y_bf16) {
    hipLaunchKernelGGL(k32_gate_ln_mod, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, s, xp, up, nrows, gm, gate_chunk, mm, shift_chunk,
                       scale_chunk, eps, y, keep, y_bf16 ? reinterpret_cast<unsigned short*>(y) : nullptr);
}
void launch32_linear(const float* a, int lda, const float* w, int ldw, const float* bias, long n, int m, int k, int mode,
                     float* c, int ldc, int col0, const ModMap& mm, int gate_chunk, int gated, float scalar, hipStream_t s,
                     int wtrans, float* c2, const void* wpack, int flags) {
    LinearParams p{a, lda, w, ldw, bias, n, m, k, mode, wtrans, c, ldc, col0, mm, gate_chunk, gated, scalar, c2, 0, {}, {}, {},
                   static_cast<const unsigned char*>(wpack), g_k32_bf16_operands, flags & 1, (flags >> 1) & 1, (flags >> 2) & 1};
    if ((flags & 4) && !(g_k32_bf16_operands && mode == 7)) {   // (fast_gelu turns mode 7 into 17, the only one that stores bf16)
        g_k32_launch_error = "launch32_linear: a bf16 result is the GELU-derivative epilogue's in the bf16-operand mode only";
        return;
    }
    if ((flags & 3) && !(g_k32_bf16_operands && (!(flags & 1) || wpack))) {
        g_k32_launch_error = "launch32_linear: bf16 operand storage outside the streamed bf16-operand kernel";
        return;
    }
    const dim3 grid((unsigned)((m + 127) / 128), (unsigned)((n + 127) / 128));
    if (!g_k32_bf16_operands) {
        hipLaunchKernelGGL(k32_linear, grid, dim3(256), 0, s, p);
        return;
    }
    const auto al = [](const void* q) { return ((unsigned long long)q & 15) == 0; };
    if (launch16_linear_wide(p, s)) return;    // 128 x 384 tiles: every trunk-sized layer (k_wide16.hip)
    if (flags & 1) {
        g_k32_launch_error = "launch32_linear: bf16 token rows need the streamed kernel (n >= 1024, m % 384 == 0)";
        return;
    }
    if (n <= 2048 && !wtrans && k % 64 == 0 && (lda & 3) == 0 && (ldw & 3) == 0 && ((unsigned long long)a & 15) == 0 &&
        ((unsigned long long)w & 15) == 0) {   // a few hundred rows: one wave per 32 x 32 tile
        hipLaunchKernelGGL(k16_linear_small, dim3((unsigned)((m + 31) / 32), (unsigned)((n + 31) / 32)), dim3(64), 0, s, p);
        return;
    }
    const bool fast = !wtrans && k % 64 == 0 && (lda & 3) == 0 && (ldw & 3) == 0 && al(a) && al(w);
    if (!fast) {
        hipLaunchKernelGGL(k16_linear, grid, dim3(256), 0, s, p);
        return;
    }
    const int nrt = (int)((n + 127) / 128), nct = (m + 127) / 128;
    hipLaunchKernelGGL(k16_linear_fast, dim3((unsigned)(8 * ((nrt + 7) / 8) * nct)), dim3(256), 0, s, p, nrt, nct);
}
// q | k | v (three [mseg][k] layers of the same input) in one pass of k16_linear_fast: c[n][col0 + j mseg + i] =
// (a . w[j][i] + bias[j][i]) * scale[j].  false: shape not eligible (or exact-fp32 mode), nothing launched.
bool launch16_linear_seg3(const float* a, int lda, const float* const* w, int ldw, const float* const* bias, const float* scale,
                          long n, int mseg, int k, float* c, int ldc, int col0, hipStream_t s, const void* wpack, bool a_bf16) {
    const auto al = [](const void* q) { return ((unsigned long long)q & 15) == 0; };
    if (!g_k32_bf16_operands || mseg % 128 || k % 64 || (lda & 3) || (ldw & 3) || !al(a) || !al(w[0]) || !al(w[1]) || !al(w[2]))
        return false;
    LinearParams p{a, lda, w[0], ldw, nullptr, n, 3 * mseg, k, 0, 0, c, ldc, col0, ModMap{nullptr, 1, 1, 0, 0}, 0, 0, 0.f, nullptr,
                   mseg, {w[0], w[1], w[2]}, {bias[0], bias[1], bias[2]}, {scale[0], scale[1], scale[2]},
                   static_cast<const unsigned char*>(wpack), 1, a_bf16 ? 1 : 0, 0};
    if (launch16_linear_wide(p, s)) return true;
    if (a_bf16) {
        g_k32_launch_error = "launch16_linear_seg3: bf16 token rows need the streamed kernel";
        return true;
    }
    if (n <= 2048) {
        hipLaunchKernelGGL(k16_linear_small, dim3((unsigned)(3 * mseg / 32), (unsigned)((n + 31) / 32)), dim3(64), 0, s, p);
        return true;
    }
    const int nrt = (int)((n + 127) / 128), nct = 3 * mseg / 128;
    hipLaunchKernelGGL(k16_linear_fast, dim3((unsigned)(8 * ((nrt + 7) / 8) * nct)), dim3(256), 0, s, p, nrt, nct);
    return true;
}
void launch32_rope(float* buf, long ntok, int ld, long pos_div, int pos_mod, const float* inv_freq, hipStream_t s) {
    const long total = ntok * kH * 12 * 2;
    hipLaunchKernelGGL(k32_rope, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, buf, ntok, ld, pos_div, pos_mod,
                       inv_freq);
}
void launch32_attn(const float* qkv, int ld, const AxisMap& ax, const MaskMap& mk, const float* bias_k, const float* bias_v,
                   const float* inv_freq, float* out, hipStream_t s, float* lse_out) {
    const int nqb = (ax.len + 255) / 256;
    hipLaunchKernelGGL(k32_attn, dim3((unsigned)((long)ax.nseq * kH * nqb)), dim3(256), 0, s, qkv, ld, ax, mk, bias_k, bias_v,
                       inv_freq, out, lse_out);
}

}  // namespace mdg
