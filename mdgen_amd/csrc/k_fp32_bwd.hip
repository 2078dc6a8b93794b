// k_fp32_bwd.hip -- backward kernels of the training step, fp32 operands.
//
// The training step differentiates the fp32 form of the network (k_fp32.hip): every reference op group is one kernel
// there, so every derivative is one kernel here, and gradients can be checked against the reference's own autograd at
// fp32 tolerance.  All reductions over tokens are two-stage with fixed slices (bit-reproducible run to run).
//   k32_dw / k32_reduce_add     dW[m][k] += sum_n dY[n][m] X[n][k]   (v_mfma_f32_32x32x2_f32, split over n); k16_dw: bf16 operands
//   k32_colsum / _final         per-group column sums  sum_t a[t][c] * b[t][c]  (biases, adaLN shift / scale / gate)
//   k32_ln_bwd (_sums)          LayerNorm (+ modulate / affine) backward (with the two adaLN column sums of the same rows)
//   k32_gate_mul, k32_gate_bwd_sums   gated residual backward (with its adaLN column sum); the GELU derivative is an epilogue
//                               of the dX product (linear.h mode 7)
//   k32_attn_bwd_q / _kv        softmax attention backward (query pass: dq; key pass: dk, dv incl. the bias key)
//   k32_rope_bwd                inverse rotation of dq, dk and the q scale
//   k32_ipa_bwd_q / _kv         invariant point attention backward (sliced over workgroups when the groups are few)
//   k32_skinny_wt               [B rows] x [long contraction] product behind the time embedder
//   k32_loss_grad               d(mean_b masked-MSE_b) / d pred
//   k32_sum_frames              d ipa_out[b,l] = sum_t dh0[b,t,l]
#include <cstdio>
#include <cstdlib>
#include "kernels.h"

namespace mdg {

// ---- dW ---------------------------------------------------------------------------------------------------------------
// part[z][m][k] = sum_{n in slice z} dY[n][m] * X[n][k];  grid (ceil(M/128), ceil(K/128), nsplit), 256 threads.
// Workgroup tile 128 x 128, wave tile 64 x 64 (2 x 2 MFMA tiles), sixteen token rows per step, the next step's rows
// requested before the current step's MFMAs.  Both operands are read along their contiguous dimension (thread -> token
// row tid / 16, eight consecutive columns: two 16-byte loads each) and transposed on their way into LDS.
__global__ __launch_bounds__(256) void k32_dw(const float* __restrict__ dy, int ldy, const float* __restrict__ x, int ldx,
                                              long n, int m, int k, float* __restrict__ part) {
    constexpr int BN = 16, LD = BN + 1, TM = 128;
    __shared__ float As[TM * LD];   // dY^T tile: [m][n]
    __shared__ float Bs[TM * LD];   // X^T  tile: [k][n]
    const int lane = lane_id(), w = wave_id();
    const int m0 = blockIdx.x * TM, k0 = blockIdx.y * TM;
    const long per = ((n + gridDim.z - 1) / gridDim.z + BN - 1) / BN * BN;
    const long nlo = (long)blockIdx.z * per, nhi = nlo + per < n ? nlo + per : n;
    const int wr = w >> 1, wc = w & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = opaque_zero();
    const int ln = threadIdx.x >> 4, lc = (threadIdx.x & 15) * 8;   // staging: 16 token rows x 128 columns (8 per thread)
    const bool veca = ((ldy | m) & 7) == 0 && ((unsigned long long)dy & 15) == 0;
    const bool vecb = ((ldx | k) & 7) == 0 && ((unsigned long long)x & 15) == 0;
    float av[8], bv[8];
    auto fetch = [&](long n0) {
        const long row = n0 + ln < nhi ? n0 + ln : (nhi > 0 ? nhi - 1 : 0);
        const bool rok = n0 + ln < nhi;
        if (veca) {
            const int mc = m0 + lc < m ? m0 + lc : 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(dy + row * ldy + mc + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) av[4 * h + j] = (rok && m0 + lc < m) ? v[j] : 0.f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int mc = m0 + lc + j;
                const float v = dy[row * ldy + (mc < m ? mc : m - 1)];
                av[j] = (rok && mc < m) ? v : 0.f;
            }
        }
        if (vecb) {
            const int kc = k0 + lc < k ? k0 + lc : 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + row * ldx + kc + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[4 * h + j] = (rok && k0 + lc < k) ? v[j] : 0.f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kc = k0 + lc + j;
                const float v = x[row * ldx + (kc < k ? kc : k - 1)];
                bv[j] = (rok && kc < k) ? v : 0.f;
            }
        }
    };
    if (nlo < nhi) fetch(nlo);
    const int i = lane & 31, kh = lane >> 5;
    for (long n0 = nlo; n0 < nhi; n0 += BN) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            As[(lc + j) * LD + ln] = av[j];
            Bs[(lc + j) * LD + ln] = bv[j];
        }
        __syncthreads();
        if (n0 + BN < nhi) fetch(n0 + BN);
#pragma unroll
        for (int kk = 0; kk < BN; kk += 2) {
            float a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = As[(wr * 64 + t * 32 + i) * LD + kk + kh];
                b[t] = Bs[(wc * 64 + t * 32 + i) * LD + kk + kh];
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[u], acc[t][u], 0, 0, 0);
        }
    }
    const int hh = lane >> 5;
    float* dst = part + (long)blockIdx.z * m * k;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int col = k0 + wc * 64 + u * 32 + (lane & 31);
        if (col >= k) continue;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * 64 + t * 32 + mfma_row(r, hh);
                if (row < m) dst[(long)row * k + col] = acc[t][u][r];
            }
    }
}
// The bf16-operand weight gradient (option train_precision = 16) as a kernel of its own, built like k16_linear (k_fp32.hip):
// 64 token rows per step (four MFMA k-steps), two LDS buffers and one barrier per step, operands rounded to bf16 on their
// way into LDS.  Both operands are read along their contiguous dimension and transposed on the way in: a thread carries FOUR
// ADJACENT token rows of a few columns, so that the four token values of a column are one 8-byte LDS store.
// part[z][m][k] = sum_{n in slice z} dY[n][m] * X[n][k].
// FAST: both operands 16-byte aligned with row strides and widths that are multiples of 8 (every trunk layer), decided by
// the launcher: no run-time-tested blocks around the loads, and each load instruction of a wave covers two whole 512-byte
// row segments (lane -> 16-byte piece tid & 31 of token row 4 (tid >> 5) + z).
template <bool FAST>
__global__ __launch_bounds__(256, 2) void k16_dw(const float* __restrict__ dy, int ldy, const float* __restrict__ x, int ldx,
                                                 long n, int m, int k, float* __restrict__ part, float* __restrict__ bpart) {
    constexpr int BN = 64, TM = 128, ROWB = 144;
    __shared__ __attribute__((aligned(16))) unsigned char Ab[2][TM * ROWB];   // dY^T tile: [m][n]
    __shared__ __attribute__((aligned(16))) unsigned char Bb[2][TM * ROWB];   // X^T  tile: [k][n]
    const int lane = lane_id(), w = wave_id();
    const int m0 = blockIdx.x * TM, k0 = blockIdx.y * TM;
    const long per = ((n + gridDim.z - 1) / gridDim.z + BN - 1) / BN * BN;
    const long nlo = (long)blockIdx.z * per, nhi = nlo + per < n ? nlo + per : n;
    const int wr = w >> 1, wc = w & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = opaque_zero();
    // general: token rows 4 rg .. 4 rg + 3, columns lc .. lc + 7;  FAST: token rows 4 (rq + 8 zz) .. + 3, columns lp .. lp + 3
    const int rg = threadIdx.x >> 4, lc = (threadIdx.x & 15) * 8;
    const int rq = threadIdx.x >> 5, lp = (threadIdx.x & 31) * 4;
    const bool veca = !FAST && ((ldy | m) & 7) == 0 && ((unsigned long long)dy & 15) == 0;
    const bool vecb = !FAST && ((ldx | k) & 7) == 0 && ((unsigned long long)x & 15) == 0;
    float av[4][8], bv[4][8];    // FAST: [z][4 zz + j]
    auto fetch = [&](long n0) {
        if (FAST) {
            const int mc = m0 + lp < m ? m0 + lp : 0, kc = k0 + lp < k ? k0 + lp : 0;
#pragma unroll
            for (int zz = 0; zz < 2; ++zz)
#pragma unroll
                for (int z = 0; z < 4; ++z) {
                    const long rw = n0 + 4 * (rq + 8 * zz) + z;
                    const long row = rw < nhi ? rw : nhi - 1;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(dy + row * ldy + mc);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(x + row * ldx + kc);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        av[z][4 * zz + j] = rw < nhi ? a[j] : 0.f;     // columns past m / k: clamped loads, results never stored
                        bv[z][4 * zz + j] = rw < nhi ? b[j] : 0.f;
                    }
                }
            return;
        }
#pragma unroll
        for (int z = 0; z < 4; ++z) {
            const long rw = n0 + 4 * rg + z;
            const long row = rw < nhi ? rw : (nhi > 0 ? nhi - 1 : 0);
            const bool rok = rw < nhi;
            if (veca) {
                const int mc = m0 + lc < m ? m0 + lc : 0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(dy + row * ldy + mc + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) av[z][4 * h + j] = (rok && m0 + lc < m) ? v[j] : 0.f;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int mc = m0 + lc + j;
                    const float v = dy[row * ldy + (mc < m ? mc : m - 1)];
                    av[z][j] = (rok && mc < m) ? v : 0.f;
                }
            }
            if (vecb) {
                const int kc = k0 + lc < k ? k0 + lc : 0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(x + row * ldx + kc + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) bv[z][4 * h + j] = (rok && k0 + lc < k) ? v[j] : 0.f;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int kc = k0 + lc + j;
                    const float v = x[row * ldx + (kc < k ? kc : k - 1)];
                    bv[z][j] = (rok && kc < k) ? v : 0.f;
                }
            }
        }
    };
    // FAST, bpart != nullptr: the workgroups of the first k tile also sum their dY columns (the bias gradient, in fp32
    // from the registers the operand passes through): bpart[z][m]
    const bool colsum = FAST && bpart && blockIdx.y == 0;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    auto stage = [&](int buf) {
        if (FAST) {
            if (colsum)
#pragma unroll
                for (int c = 0; c < 8; ++c) cs[c & 3] += (av[0][c] + av[1][c]) + (av[2][c] + av[3][c]);
#pragma unroll
            for (int zz = 0; zz < 2; ++zz)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = 4 * zz + j;
                    *reinterpret_cast<u32x2*>(&Ab[buf][(lp + j) * ROWB + 8 * (rq + 8 * zz)]) = u32x2{pack_bf16(av[0][c], av[1][c]), pack_bf16(av[2][c], av[3][c])};
                    *reinterpret_cast<u32x2*>(&Bb[buf][(lp + j) * ROWB + 8 * (rq + 8 * zz)]) = u32x2{pack_bf16(bv[0][c], bv[1][c]), pack_bf16(bv[2][c], bv[3][c])};
                }
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            *reinterpret_cast<u32x2*>(&Ab[buf][(lc + j) * ROWB + 8 * rg]) = u32x2{pack_bf16(av[0][j], av[1][j]), pack_bf16(av[2][j], av[3][j])};
            *reinterpret_cast<u32x2*>(&Bb[buf][(lc + j) * ROWB + 8 * rg]) = u32x2{pack_bf16(bv[0][j], bv[1][j]), pack_bf16(bv[2][j], bv[3][j])};
        }
    };
    const int i = lane & 31, kh = lane >> 5;
    if (nlo < nhi) {
        fetch(nlo);
        stage(0);
    }
    __syncthreads();
    int buf = 0;
    for (long n0 = nlo; n0 < nhi; n0 += BN, buf ^= 1) {
        const bool more = n0 + BN < nhi;
        if (more) fetch(n0 + BN);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = *reinterpret_cast<const bf16x8*>(&Ab[buf][(wr * 64 + t * 32 + i) * ROWB + ks * 32 + kh * 16]);
                b[t] = *reinterpret_cast<const bf16x8*>(&Bb[buf][(wc * 64 + t * 32 + i) * ROWB + ks * 32 + kh * 16]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b[u], acc[t][u], 0, 0, 0);
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
    }
    const int hh = lane >> 5;
    float* dst = part + (long)blockIdx.z * m * k;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int col = k0 + wc * 64 + u * 32 + (lane & 31);
        if (col >= k) continue;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * 64 + t * 32 + mfma_row(r, hh);
                if (row < m) dst[(long)row * k + col] = acc[t][u][r];
            }
    }
    if (colsum) {   // eight row groups per column -> one value (the operand tiles are dead: the loop ended on a barrier)
        float* red = reinterpret_cast<float*>(&Ab[0][0]);
#pragma unroll
        for (int j = 0; j < 4; ++j) red[rq * TM + lp + j] = cs[j];
        __syncthreads();
        if (threadIdx.x < TM && m0 + threadIdx.x < m) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) v += red[g * TM + threadIdx.x];
            bpart[(long)blockIdx.z * m + m0 + threadIdx.x] = v;
        }
    }
}

__global__ void k32_reduce_add(const float* __restrict__ part, int nsplit, long stride, long count, float* __restrict__ dst) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float s = 0.f;
    for (int z = 0; z < nsplit; ++z) s += part[(long)z * stride + i];
    dst[i] += s;
}

// out[b][c] = sum_r x[b][r] W[r][c] for a handful of rows b and a LONG contraction (the time embedder behind all adaLN
// heads: B rows, K = every modulation output of the network).  As a GEMM tile that is three workgroups walking the
// whole weight one after the other (1.7 ms at cfg-5); here it is a matrix-vector product split over slices of 64
// weight rows, thread = output column, partials summed in a fixed order by the second kernel.
__global__ __launch_bounds__(128) void k32_skinny_wt(const float* __restrict__ x, int ldx, const float* __restrict__ W, int ldw,
                                                     int nb, int m, long K, float* __restrict__ part) {
    const int c = blockIdx.y * 128 + threadIdx.x;
    const long r0 = (long)blockIdx.x * 64, r1 = r0 + 64 < K ? r0 + 64 : K;
    const int b0 = blockIdx.z * 8;
    float acc[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[b] = 0.f;
    if (c < m)
        for (long r = r0; r < r1; ++r) {
            const float w = W[r * ldw + c];
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[b] += (b0 + b < nb ? x[(long)(b0 + b) * ldx + r] : 0.f) * w;
        }
    if (c < m)
#pragma unroll
        for (int b = 0; b < 8; ++b)
            if (b0 + b < nb) part[((long)blockIdx.x * nb + b0 + b) * m + c] = acc[b];
}
// 16 outputs per workgroup, 16 threads per output walking the slices z = sl, sl + 16, ..., their sums added in a fixed order
__global__ __launch_bounds__(256) void k32_skinny_final(const float* __restrict__ part, int nslice, long count, float* __restrict__ out) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long i = (long)blockIdx.x * 16 + cl;
    float s = 0.f;
    if (i < count)
        for (int z = sl; z < nslice; z += 16) s += part[(long)z * count + i];
    red[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && i < count) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        out[i] = t;
    }
}
bool launch32_skinny_wt(const float* x, int ldx, const float* W, int ldw, int nb, int m, long K, float* out, float* part,
                        size_t part_floats, hipStream_t s) {
    const long nslice = (K + 63) / 64;
    if ((size_t)nslice * nb * m > part_floats) return false;
    hipLaunchKernelGGL(k32_skinny_wt, dim3((unsigned)nslice, (unsigned)((m + 127) / 128), (unsigned)((nb + 7) / 8)), dim3(128), 0, s, x,
                       ldx, W, ldw, nb, m, K, part);
    const long count = (long)nb * m;
    hipLaunchKernelGGL(k32_skinny_final, dim3((unsigned)((count + 15) / 16)), dim3(256), 0, s, part, (int)nslice, count, out);
    return true;
}

// ---- grouped column sums ----------------------------------------------------------------------------------------------
// partial[slice][c] = sum over the slice's rows of a[t][c] * B(t, c), B by `mode`:
//   0: 1     1: b[t][c]     2: xhat(t, c) = (x[t][c] - mean_t) * rstd_t with x = b (LayerNorm statistics recomputed)
//   3: roww[t] (a per-row weight, e.g. an indicator)
// Groups are contiguous runs of tokens_per_group rows; slice = (group, i) covers rows_per_slice rows of it.
struct ColsumParams {
    const float* a; int lda;
    const float* b; int ldb;
    const float* roww;
    int mode;
    long nrows; int ncols;
    long tokens_per_group; int rows_per_slice; int slices_per_group;
    float eps;
    float* partial;
};
__global__ __launch_bounds__(384) void k32_colsum(const ColsumParams p) {
    const long g = blockIdx.x / p.slices_per_group;
    const int si = blockIdx.x % p.slices_per_group;
    const long g0 = g * p.tokens_per_group;
    long lo = g0 + (long)si * p.rows_per_slice, hi = lo + p.rows_per_slice;
    const long gend = g0 + p.tokens_per_group < p.nrows ? g0 + p.tokens_per_group : p.nrows;
    hi = hi < gend ? hi : gend;
    if (p.mode == 2) {
        // b = LayerNorm-normalised row (384 columns): one WAVE per row (lane holds columns lane + 64 i), so the row
        // statistics are wave reductions and six rows are in flight per workgroup; the six waves' column sums are
        // combined in a fixed order at the end.
        __shared__ float red[6 * kC];
        const int w = wave_id(), lane = lane_id();
        float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (long t = lo + w; t < hi; t += 6) {
            float xv[6], av[6];
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                xv[i] = p.b[t * p.ldb + lane + 64 * i];
                av[i] = p.a[t * p.lda + lane + 64 * i];
                sm += xv[i];
            }
            const float mean = wave_sum(sm) * (1.0f / kC);
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                xv[i] -= mean;
                sq += xv[i] * xv[i];
            }
            const float rstd = 1.0f / sqrtf(wave_sum(sq) * (1.0f / kC) + p.eps);
#pragma unroll
            for (int i = 0; i < 6; ++i) s[i] += av[i] * (xv[i] * rstd);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) red[w * kC + lane + 64 * i] = s[i];
        __syncthreads();
        const int c = threadIdx.x;
        p.partial[(long)blockIdx.x * p.ncols + c] =
            ((red[c] + red[kC + c]) + (red[2 * kC + c] + red[3 * kC + c])) + (red[4 * kC + c] + red[5 * kC + c]);
        return;
    }
    const int c = blockIdx.y * 384 + threadIdx.x;   // column chunks of 384
    const int cc = c < p.ncols ? c : p.ncols - 1;   // loads are unconditional (clamped column), eight rows in flight
    float s = 0.f;
    long t = lo;
    for (; t + 8 <= hi; t += 8) {
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            av[u] = p.a[(t + u) * p.lda + cc];
            bv[u] = p.mode == 1 ? p.b[(t + u) * p.ldb + cc] : p.mode == 3 ? p.roww[t + u] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += av[u] * bv[u];
    }
    for (; t < hi; ++t) {
        const float bb = p.mode == 1 ? p.b[t * p.ldb + cc] : p.mode == 3 ? p.roww[t] : 1.f;
        s += p.a[t * p.lda + cc] * bb;
    }
    if (c < p.ncols) p.partial[(long)blockIdx.x * p.ncols + c] = s;
}
// out[g * ldo + c] += sum_i partial[(g * spg + i)][c].  Workgroup = 16 columns x 16 slice lanes: lane s sums slices
// s, s + 16, ... (four loads in flight), the sixteen lane sums are added in a fixed order -- bit-reproducible, and
// 16 x the parallelism of one thread per column (at B = 1 a bias gradient is ONE group of ~1000 slices).
__global__ __launch_bounds__(256) void k32_colsum_final(const float* __restrict__ partial, int ngroups, int spg, int ncols,
                                                        float* __restrict__ out, long ldo) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int ncb = (ncols + 15) / 16;
    const int g = blockIdx.x / ncb, c = (blockIdx.x % ncb) * 16 + cl;
    const int cc = c < ncols ? c : ncols - 1;
    const float* base = partial + (long)g * spg * ncols + cc;
    float s = 0.f;
    int k = sl;
    for (; k + 48 < spg; k += 64) {
        const float v0 = base[(long)k * ncols], v1 = base[(long)(k + 16) * ncols], v2 = base[(long)(k + 32) * ncols],
                    v3 = base[(long)(k + 48) * ncols];
        s += v0;
        s += v1;
        s += v2;
        s += v3;
    }
    for (; k < spg; k += 16) s += base[(long)k * ncols];
    red[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && c < ncols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][cl];
        out[(long)g * ldo + c] += t;
    }
}

// ---- LayerNorm (+ modulate | affine) backward -------------------------------------------------------------------------
// y = xhat * mult + shift, mult = 1 + scale[g] (modulate) or gamma (affine).  Given dy: dx (+)= rstd * (dxh - mean(dxh)
// - xhat * mean(dxh * xhat)), dxh = dy * mult.  One wave per row.
__global__ __launch_bounds__(256) void k32_ln_bwd(const float* __restrict__ x, const float* __restrict__ dy, long nrows,
                                                  ModMap mm, int scale_chunk, int affine, float eps, float* __restrict__ dx,
                                                  int accumulate) {
    const long row = (long)blockIdx.x * 4 + wave_id();
    if (row >= nrows) return;
    const int lane = lane_id();
    float v[6], g[6];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        v[i] = x[row * kC + lane + 64 * i];
        s += v[i];
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
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c = lane + 64 * i;
        const float sc = mod[scale_chunk * kC + c];
        v[i] *= rstd;                                            // xhat
        g[i] = dy[row * kC + c] * (affine ? sc : 1.0f + sc);     // dxhat
        s1 += g[i];
        s2 += g[i] * v[i];
    }
    s1 = wave_sum(s1) * (1.0f / kC);
    s2 = wave_sum(s2) * (1.0f / kC);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c = lane + 64 * i;
        const float d = rstd * (g[i] - s1 - v[i] * s2);
        dx[row * kC + c] = accumulate ? dx[row * kC + c] + d : d;
    }
}

// LayerNorm + modulate backward WITH the two adaLN column sums of the same rows (d shift[g] += sum dy, d scale[g] += sum dy
// xhat): the separate k32_colsum passes re-read dy twice and x once.  A workgroup owns one slice of rows_per_slice rows of one
// group (four waves, wave w the rows w, w + 4, ...), keeps the column sums of its rows in registers, and writes one partial
// row [shift(384) | scale(384)] that k32_colsum_final reduces in a fixed order (same two-stage determinism as k32_colsum).
__global__ __launch_bounds__(256) void k32_ln_bwd_sums(const float* __restrict__ x, const float* __restrict__ dy, long nrows, ModMap mm,
                                                       int scale_chunk, float eps, float* __restrict__ dx, int accumulate,
                                                       long tokens_per_group, int rps, int spg, float* __restrict__ partial) {
    __shared__ float red[4][2 * kC];
    const long grp = blockIdx.x / spg;
    const int si = blockIdx.x % spg;
    const long lo = grp * tokens_per_group + (long)si * rps;
    long hi = lo + rps;
    if (hi > (grp + 1) * tokens_per_group) hi = (grp + 1) * tokens_per_group;
    if (hi > nrows) hi = nrows;
    const int lane = lane_id(), w = wave_id();
    float c1[6], c2[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) c1[i] = c2[i] = 0.f;
    for (long row = lo + w; row < hi; row += 4) {
        float v[6], g[6], d[6];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            v[i] = x[row * kC + lane + 64 * i];
            d[i] = dy[row * kC + lane + 64 * i];
            s += v[i];
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
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int c = lane + 64 * i;
            v[i] *= rstd;                                          // xhat
            g[i] = d[i] * (1.0f + mod[scale_chunk * kC + c]);      // dxhat
            s1 += g[i];
            s2 += g[i] * v[i];
            c1[i] += d[i];
            c2[i] += d[i] * v[i];
        }
        s1 = wave_sum(s1) * (1.0f / kC);
        s2 = wave_sum(s2) * (1.0f / kC);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int c = lane + 64 * i;
            const float r = rstd * (g[i] - s1 - v[i] * s2);
            dx[row * kC + c] = accumulate ? dx[row * kC + c] + r : r;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        red[w][lane + 64 * i] = c1[i];
        red[w][kC + lane + 64 * i] = c2[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * kC; c += 256)
        partial[(long)blockIdx.x * 2 * kC + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// du = gate * dh AND d gate[g] += sum dh * u over the same rows (one pass over dh instead of two); slices as above.
__global__ __launch_bounds__(256) void k32_gate_bwd_sums(const float* __restrict__ dh, const float* __restrict__ u, long nrows,
                                                         ModMap mm, int gate_chunk, float* __restrict__ du, long tokens_per_group,
                                                         int rps, int spg, float* __restrict__ partial, int du_bf16) {
    __shared__ float red[4][kC];
    const long grp = blockIdx.x / spg;
    const int si = blockIdx.x % spg;
    const long lo = grp * tokens_per_group + (long)si * rps;
    long hi = lo + rps;
    if (hi > (grp + 1) * tokens_per_group) hi = (grp + 1) * tokens_per_group;
    if (hi > nrows) hi = nrows;
    const int lane = lane_id(), w = wave_id();
    float c1[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) c1[i] = 0.f;
    for (long row = lo + w; row < hi; row += 4) {
        const float* mod = mm.mod + mm.row_off(row) + gate_chunk * kC;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int c = lane + 64 * i;
            const float d = dh[row * kC + c];
            c1[i] += d * u[row * kC + c];
            // du_bf16: du is stored as bf16 rows (only ever the token operand of a dX product and dY of a weight gradient)
            if (du_bf16) reinterpret_cast<unsigned short*>(du)[row * kC + c] = (unsigned short)(pack_bf16(d * mod[c], 0.f) & 0xffffu);
            else du[row * kC + c] = d * mod[c];
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) red[w][lane + 64 * i] = c1[i];
    __syncthreads();
    for (int c = threadIdx.x; c < kC; c += 256) partial[(long)blockIdx.x * kC + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// out[t][c] = a[t][c] * gate[g(t)][c]   (du = gate * dh; gate == null -> copy)
__global__ void k32_gate_mul(const float* __restrict__ a, long nrows, ModMap mm, int gate_chunk, int gated,
                             float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * kC) return;
    const long t = i / kC;
    const int c = (int)(i % kC);
    const float g = gated ? mm.mod[mm.row_off(t) + gate_chunk * kC + c] : 1.0f;
    out[i] = a[i] * g;
}

// ---- attention backward ------------------------------------------------------------------------------------------------
// Shared helpers: staging of one (sequence, head) key tile as in k32_attn (bias key at position len, rotated there).
__device__ __forceinline__ void bias_kv(const float* bias_k, const float* bias_v, const float* inv_freq, int hd, int len, int d,
                                        float& kvv, float& vvv) {
    const int i = d % 12;
    const float ang = (float)len * inv_freq[i];
    const float c = cosf(ang), s = sinf(ang);
    const float x1 = bias_k[hd * kDH + i], x2 = bias_k[hd * kDH + i + 12];
    kvv = d < 12 ? x1 * c - x2 * s : x2 * c + x1 * s;
    vvv = bias_v[hd * kDH + d];
}

// Query pass: thread = query.  lse = m + log(l) is recomputed here (a first sweep over the keys), then
// delta = dO . O, ds = p (dO . v - delta), dq = sum_j ds k_j.  Writes dq into dqkv[:, 0:384], and (lse, delta) into
// stats[token][head][2] for the key pass.
__global__ __launch_bounds__(256) void k32_attn_bwd_q(const float* __restrict__ qkv, int ld, AxisMap ax, MaskMap mk,
                                                      const float* __restrict__ bias_k, const float* __restrict__ bias_v,
                                                      const float* __restrict__ inv_freq, const float* __restrict__ o,
                                                      const float* __restrict__ dout, float* __restrict__ dqkv,
                                                      float* __restrict__ stats, const float* __restrict__ lse_in) {
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
    float q[kDH], dO[kDH], dq[kDH];
    float delta = 0.f;
#pragma unroll
    for (int d = 0; d < kDH; ++d) {
        q[d] = qkv[qtok * ld + hd * kDH + d];
        dO[d] = dout[qtok * kC + hd * kDH + d];
        delta += dO[d] * o[qtok * kC + hd * kDH + d];
        dq[d] = 0.f;
    }
    float mrun = -3.0e38f, den = 0.f;
    // pass 0 recomputes the row's log-sum-exp; with the forward pass's own value on the tape (lse_in, the same
    // arithmetic in the same order) it is skipped
    const float lse_tape = lse_in ? lse_in[qtok * kH + hd] : 0.f;
    for (int pass = lse_in ? 1 : 0; pass < 2; ++pass) {
        const float lse = lse_in ? lse_tape : (pass ? mrun + logf(den) : 0.f);
        for (int j0 = 0; j0 < len + 1; j0 += KT) {
            __syncthreads();
            for (int e = tid; e < KT * kDH; e += 256) {
                const int jj = e / kDH, d = e % kDH;
                const int j = j0 + jj;
                float kvv = 0.f, vvv = 0.f;
                if (j < len) {
                    const long t = ax.token(seq, j);
                    kvv = qkv[t * ld + kC + hd * kDH + d];
                    vvv = qkv[t * ld + 2 * kC + hd * kDH + d];
                } else if (j == len) {
                    bias_kv(bias_k, bias_v, inv_freq, hd, len, d, kvv, vvv);
                }
                sk[jj][d] = kvv;
                sv[jj][d] = vvv;
            }
            if (tid < KT) {
                const int j = j0 + tid;
                sm[tid] = j < len ? (mk.at(ax.token(seq, j)) != 0.f ? 1.f : 0.f) : (j == len ? 1.f : 0.f);
            }
            __syncthreads();
            if (pass == 0) {
                float tmax = -3.0e38f;
                float lg[KT];
#pragma unroll
                for (int jj = 0; jj < KT; ++jj) {
                    float dot = 0.f;
#pragma unroll
                    for (int d = 0; d < kDH; ++d) dot += q[d] * sk[jj][d];
                    lg[jj] = sm[jj] != 0.f ? dot : -3.0e38f;
                    tmax = fmaxf(tmax, lg[jj]);
                }
                const float mnew = fmaxf(mrun, tmax);
                den *= expf(mrun - mnew);
                mrun = mnew;
#pragma unroll
                for (int jj = 0; jj < KT; ++jj) den += lg[jj] > -1.0e38f ? expf(lg[jj] - mnew) : 0.f;
            } else {
#pragma unroll 4
                for (int jj = 0; jj < KT; ++jj) {
                    if (sm[jj] == 0.f) continue;
                    float dot = 0.f, dp = 0.f;
#pragma unroll
                    for (int d = 0; d < kDH; ++d) {
                        dot += q[d] * sk[jj][d];
                        dp += dO[d] * sv[jj][d];
                    }
                    const float ds = expf(dot - lse) * (dp - delta);
#pragma unroll
                    for (int d = 0; d < kDH; ++d) dq[d] += ds * sk[jj][d];
                }
            }
        }
    }
    if (!qok) return;
#pragma unroll
    for (int d = 0; d < kDH; ++d) dqkv[qtok * ld + hd * kDH + d] = dq[d];
    stats[(qtok * kH + hd) * 2] = lse_in ? lse_tape : mrun + logf(den);
    stats[(qtok * kH + hd) * 2 + 1] = delta;
}

// Key pass: thread = key (the bias key is key `len`), queries staged through LDS.  dk = sum_i ds_ij q_i, dv = sum_i p_ij dO_i.
// Real keys write dqkv[:, 384:1152]; the bias key writes dbias[seq][dk rotated back: head x 24 | dv: head x 24],
// reduced over sequences later (two column sums: the rows ARE the (1, 1, C) bias tensors' layout).
__global__ __launch_bounds__(256) void k32_attn_bwd_kv(const float* __restrict__ qkv, int ld, AxisMap ax, MaskMap mk,
                                                       const float* __restrict__ bias_k, const float* __restrict__ bias_v,
                                                       const float* __restrict__ inv_freq, const float* __restrict__ dout,
                                                       const float* __restrict__ stats, float* __restrict__ dqkv,
                                                       float* __restrict__ dbias) {
    constexpr int QT = 64;
    __shared__ __attribute__((aligned(16))) float sq[QT][kDH];
    __shared__ __attribute__((aligned(16))) float sdo[QT][kDH];
    __shared__ float slse[QT], sdel[QT];
    const int len = ax.len;
    const int nkb = (len + 1 + 255) / 256;
    const int kb = blockIdx.x % nkb;
    const int hd = (blockIdx.x / nkb) % kH;
    const int seq = blockIdx.x / (nkb * kH);
    const int tid = threadIdx.x;
    const int j = kb * 256 + tid;
    const bool kok = j <= len;
    float k[kDH], v[kDH], dk[kDH], dv[kDH];
    bool valid = false;
    long ktok = 0;
    if (j < len) {
        ktok = ax.token(seq, j);
        valid = mk.at(ktok) != 0.f;
#pragma unroll
        for (int d = 0; d < kDH; ++d) {
            k[d] = qkv[ktok * ld + kC + hd * kDH + d];
            v[d] = qkv[ktok * ld + 2 * kC + hd * kDH + d];
        }
    } else {
        valid = j == len;
#pragma unroll
        for (int d = 0; d < kDH; ++d) bias_kv(bias_k, bias_v, inv_freq, hd, len, d, k[d], v[d]);
    }
#pragma unroll
    for (int d = 0; d < kDH; ++d) dk[d] = dv[d] = 0.f;
    for (int i0 = 0; i0 < len; i0 += QT) {
        __syncthreads();
        for (int e = tid; e < QT * kDH; e += 256) {
            const int ii = e / kDH, d = e % kDH;
            const int i = i0 + ii;
            float qv = 0.f, dov = 0.f;
            if (i < len) {
                const long t = ax.token(seq, i);
                qv = qkv[t * ld + hd * kDH + d];
                dov = dout[t * kC + hd * kDH + d];
            }
            sq[ii][d] = qv;
            sdo[ii][d] = dov;
        }
        if (tid < QT) {
            const int i = i0 + tid;
            const long t = ax.token(seq, i < len ? i : len - 1);
            slse[tid] = i < len ? stats[(t * kH + hd) * 2] : 3.0e38f;    // beyond len: p = exp(-inf) = 0
            sdel[tid] = stats[(t * kH + hd) * 2 + 1];
        }
        __syncthreads();
        if (valid) {
#pragma unroll 4
            for (int ii = 0; ii < QT; ++ii) {
                float dot = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < kDH; ++d) {
                    dot += sq[ii][d] * k[d];
                    dp += sdo[ii][d] * v[d];
                }
                const float p = expf(dot - slse[ii]);
                const float ds = p * (dp - sdel[ii]);
#pragma unroll
                for (int d = 0; d < kDH; ++d) {
                    dk[d] += ds * sq[ii][d];
                    dv[d] += p * sdo[ii][d];
                }
            }
        }
    }
    if (!kok) return;
    if (j < len) {
#pragma unroll
        for (int d = 0; d < kDH; ++d) {
            dqkv[ktok * ld + kC + hd * kDH + d] = dk[d];
            dqkv[ktok * ld + 2 * kC + hd * kDH + d] = dv[d];
        }
    } else {   // bias key: undo its rotation (position len) here, so that the per-sequence rows just add up
        float* dst = dbias + (long)seq * 2 * kC + hd * kDH;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const float ang = (float)len * inv_freq[i];
            const float c = cosf(ang), s = sinf(ang);
            dst[i] = dk[i] * c + dk[i + 12] * s;            // R(-theta): d x1 = d y1 c + d y2 s
            dst[i + 12] = dk[i + 12] * c - dk[i] * s;       //            d x2 = d y2 c - d y1 s
        }
#pragma unroll
        for (int d = 0; d < kDH; ++d) dst[kC + d] = dv[d];
    }
}

// dq, dk (post-RoPE) -> pre-RoPE in place: inverse rotation; dq additionally times the q scale.
__global__ void k32_rope_bwd(float* __restrict__ buf, long ntok, int ld, long pos_div, int pos_mod,
                             const float* __restrict__ inv_freq, float qscale) {
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
    const float y1 = v[i], y2 = v[i + 12];
    const float sc = which == 0 ? qscale : 1.0f;
    v[i] = (y1 * c + y2 * s) * sc;
    v[i + 12] = (y2 * c - y1 * s) * sc;
}

// ---- loss ----------------------------------------------------------------------------------------------------------
// total = mean_b ( sum((pred - ut)^2 m) / sum(m) )   (transport.py:13-17, 184; wrapper.py:403 loss.mean())
// den[b] = sum(m) from k32_colsum-free tiny kernel; dpred = 2 (pred - ut) m / den[b] / B
__global__ __launch_bounds__(256) void k32_mask_sum(const float* __restrict__ mask, long per_sample, float* __restrict__ den) {
    __shared__ float red[4];
    const long base = (long)blockIdx.x * per_sample;
    float s = 0.f;
    long i = threadIdx.x;
    for (; i + 7 * 256 < per_sample; i += 8 * 256) {
        float mv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) mv[u] = mask[base + i + 256 * u];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += mv[u];
    }
    for (; i < per_sample; i += 256) s += mask[base + i];
    s = wave_sum(s);
    if (lane_id() == 0) red[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) den[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void k32_loss_grad(const float* __restrict__ pred, const float* __restrict__ target, const float* __restrict__ mask,
                              const float* __restrict__ den, long per_sample, long total, float inv_b,
                              float* __restrict__ dpred) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long b = i / per_sample;
    dpred[i] = 2.0f * (pred[i] - target[i]) * mask[i] / den[b] * inv_b;
}

// out[(b, l)][c] (+)= sum_t a[(b, t, l)][c]   (h = h + ipa_out[:, None]: latent_model.py:245-246)
__global__ void k32_sum_frames(const float* __restrict__ a, int B, int T, int L, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * L * kC) return;
    const int c = (int)(i % kC);
    const int l = (int)((i / kC) % L);
    const int b = (int)(i / ((long)kC * L));
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += a[(((long)b * T + t) * L + l) * kC + c];
    out[i] = s;
}

// ---- launchers ------------------------------------------------------------------------------------------------------
// dW of nseg layers that share the input x and whose dY sit side by side (dy[n][j mseg + i]): one pass over x and dY,
// m = nseg * mseg.  dw[j] / db[j] may be null.  Returns true if the bias gradients were computed by the same pass.
int launch16_dw_wide(const float* dy, int ldy, const float* x, int ldx, long n, int m, int k, float* part, size_t part_floats,
                     bool want_db, float** bpart_out, hipStream_t s, bool x_bf16, bool dy_bf16);   // k_wide16.hip
bool launch32_dw_seg(const float* dy, int ldy, const float* x, int ldx, long n, int mseg, int nseg, int k, float* const* dw,
                     float* const* db, float* part, size_t part_floats, hipStream_t s, bool x_bf16, bool dy_bf16) {
    const int m = mseg * nseg;
    bool want_db = false;
    for (int j = 0; j < nseg; ++j) want_db = want_db || (db && db[j]);
    const long count = (long)mseg * k;
    auto reduce = [&](int nsplit, float* bpart) {
        for (int j = 0; j < nseg; ++j) {
            if (dw[j])
                hipLaunchKernelGGL(k32_reduce_add, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, part + (size_t)j * count, nsplit,
                                   (long)m * k, count, dw[j]);
            if (bpart && db[j])
                hipLaunchKernelGGL(k32_reduce_add, dim3((unsigned)((mseg + 255) / 256)), dim3(256), 0, s, bpart + (size_t)j * mseg, nsplit,
                                   (long)m, (long)mseg, db[j]);
        }
    };
    if (g_k32_bf16_operands) {   // 128 x 384 tiles (k_wide16.hip): each dY tile read once
        float* bpart = nullptr;
        if (const int ns = launch16_dw_wide(dy, ldy, x, ldx, n, m, k, part, part_floats, want_db, &bpart, s, x_bf16, dy_bf16)) {
            reduce(ns, bpart);
            return bpart != nullptr;
        }
    }
    if (x_bf16 || dy_bf16) {
        g_k32_launch_error = "launch32_dw: bf16 X / dY rows need the wide kernel (bf16-operand mode, n >= 4096)";
        return false;
    }
    // enough slices to fill the chip ONCE with 128 x 128 tiles at two workgroups per CU (a 384 x 384 weight is only 9 of
    // them); more slices only add partial-sum traffic (113 slices of a 384 x 384 weight: 66 MB written and read back)
    const int tiles = ((m + 127) / 128) * ((k + 127) / 128);
    int nsplit = (int)((n + 511) / 512);
    const int want = (512 + tiles - 1) / tiles;
    if (nsplit > want) nsplit = want;
    if (nsplit > 128) nsplit = 128;
    if (nsplit < 1) nsplit = 1;
    while (nsplit > 1 && (size_t)nsplit * m * (k + 1) > part_floats) --nsplit;
    const dim3 grid((m + 127) / 128, (k + 127) / 128, nsplit);
    const bool fast = ((ldy | m | ldx | k) & 7) == 0 && (((unsigned long long)dy | (unsigned long long)x) & 15) == 0;
    // bf16-operand fast path: the bias gradient (column sums of dY) rides along, partials behind the dW partials
    float* bpart = (g_k32_bf16_operands && fast && want_db && (size_t)nsplit * m * (k + 1) <= part_floats) ? part + (size_t)nsplit * m * k : nullptr;
    if (g_k32_bf16_operands && fast) hipLaunchKernelGGL(k16_dw<true>, grid, dim3(256), 0, s, dy, ldy, x, ldx, n, m, k, part, bpart);
    else if (g_k32_bf16_operands) hipLaunchKernelGGL(k16_dw<false>, grid, dim3(256), 0, s, dy, ldy, x, ldx, n, m, k, part, bpart);
    else hipLaunchKernelGGL(k32_dw, grid, dim3(256), 0, s, dy, ldy, x, ldx, n, m, k, part);
    reduce(nsplit, bpart);
    return bpart != nullptr;
}
bool launch32_dw(const float* dy, int ldy, const float* x, int ldx, long n, int m, int k, float* dw, float* part,
                 size_t part_floats, hipStream_t s, float* db, bool x_bf16, bool dy_bf16) {
    return launch32_dw_seg(dy, ldy, x, ldx, n, m, 1, k, &dw, &db, part, part_floats, s, x_bf16, dy_bf16);
}
// out[g][c] (ldo) += sum_{t in group g} a[t][c] * B(t, c); groups of tokens_per_group rows
void launch32_colsum(const float* a, int lda, const float* b, int ldb, const float* roww, int mode, long nrows, int ncols,
                     long tokens_per_group, float eps, float* out, long ldo, float* part, size_t part_floats, hipStream_t s) {
    const long ng = (nrows + tokens_per_group - 1) / tokens_per_group;
    int rps = 64;   // rows per slice: ~1000 workgroups per call at cfg-5's per-GPU size, each with 8 (mode 2: 6) rows in flight
    int spg = (int)((tokens_per_group + rps - 1) / rps);
    while ((size_t)ng * spg * ncols > part_floats && rps < (1 << 24)) {
        rps *= 2;
        spg = (int)((tokens_per_group + rps - 1) / rps);
    }
    ColsumParams p{a, lda, b, ldb, roww, mode, nrows, ncols, tokens_per_group, rps, spg, eps, part};
    hipLaunchKernelGGL(k32_colsum, dim3((unsigned)(ng * spg), (unsigned)((ncols + 383) / 384)), dim3(384), 0, s, p);
    hipLaunchKernelGGL(k32_colsum_final, dim3((unsigned)(ng * ((ncols + 15) / 16))), dim3(256), 0, s, part, (int)ng, spg, ncols, out, ldo);
}
void launch32_ln_bwd(const float* x, const float* dy, long nrows, const ModMap& mm, int scale_chunk, int affine, float eps,
                     float* dx, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(k32_ln_bwd, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, s, x, dy, nrows, mm, scale_chunk, affine,
                       eps, dx, accumulate);
}
static bool slice_plan(long nrows, long tokens_per_group, int ncols, size_t part_floats, long* ng, int* rps, int* spg) {
    *ng = (nrows + tokens_per_group - 1) / tokens_per_group;
    *rps = 64;
    while (*rps > 4 && *ng * ((tokens_per_group + *rps - 1) / *rps) < 256) *rps /= 2;   // a few hundred rows: still fill the chip
    *spg = (int)((tokens_per_group + *rps - 1) / *rps);
    while ((size_t)*ng * *spg * ncols > part_floats && *rps < (1 << 24)) {
        *rps *= 2;
        *spg = (int)((tokens_per_group + *rps - 1) / *rps);
    }
    return (size_t)*ng * *spg * ncols <= part_floats;
}
// dx (+)= LN'(dy (1 + scale)); out[g][0:384] += sum dy, out[g][384:768] += sum dy xhat  (out = the shift chunk of the group's
// modulation-gradient row, the scale chunk right behind it).  false: partial buffer too small, nothing launched.
bool launch32_ln_bwd_sums(const float* x, const float* dy, long nrows, const ModMap& mm, int scale_chunk, float eps, float* dx,
                          int accumulate, long tokens_per_group, float* out, long ldo, float* part, size_t part_floats, hipStream_t s) {
    long ng; int rps, spg;
    if (!slice_plan(nrows, tokens_per_group, 2 * kC, part_floats, &ng, &rps, &spg)) return false;
    hipLaunchKernelGGL(k32_ln_bwd_sums, dim3((unsigned)(ng * spg)), dim3(256), 0, s, x, dy, nrows, mm, scale_chunk, eps, dx, accumulate,
                       tokens_per_group, rps, spg, part);
    hipLaunchKernelGGL(k32_colsum_final, dim3((unsigned)(ng * ((2 * kC + 15) / 16))), dim3(256), 0, s, part, (int)ng, spg, 2 * kC, out, ldo);
    return true;
}
// du = gate * dh; out[g][0:384] += sum dh u
bool launch32_gate_bwd_sums(const float* dh, const float* u, long nrows, const ModMap& mm, int gate_chunk, float* du,
                            long tokens_per_group, float* out, long ldo, float* part, size_t part_floats, hipStream_t s, bool du_bf16) {
    long ng; int rps, spg;
    if (!slice_plan(nrows, tokens_per_group, kC, part_floats, &ng, &rps, &spg)) return false;
    hipLaunchKernelGGL(k32_gate_bwd_sums, dim3((unsigned)(ng * spg)), dim3(256), 0, s, dh, u, nrows, mm, gate_chunk, du, tokens_per_group,
                       rps, spg, part, du_bf16 ? 1 : 0);
    hipLaunchKernelGGL(k32_colsum_final, dim3((unsigned)(ng * ((kC + 15) / 16))), dim3(256), 0, s, part, (int)ng, spg, kC, out, ldo);
    return true;
}
void launch32_gate_mul(const float* a, long nrows, const ModMap& mm, int gate_chunk, int gated, float* out, hipStream_t s) {
    const long n = nrows * kC;
    hipLaunchKernelGGL(k32_gate_mul, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, nrows, mm, gate_chunk, gated, out);
}
void launch32_attn_bwd(const float* qkv, int ld, const AxisMap& ax, const MaskMap& mk, const float* bias_k,
                       const float* bias_v, const float* inv_freq, const float* o, const float* dout, float* dqkv,
                       float* stats, float* dbias, hipStream_t s, const float* lse_in) {
    const int nqb = (ax.len + 255) / 256, nkb = (ax.len + 1 + 255) / 256;
    hipLaunchKernelGGL(k32_attn_bwd_q, dim3((unsigned)((long)ax.nseq * kH * nqb)), dim3(256), 0, s, qkv, ld, ax, mk, bias_k,
                       bias_v, inv_freq, o, dout, dqkv, stats, lse_in);
    hipLaunchKernelGGL(k32_attn_bwd_kv, dim3((unsigned)((long)ax.nseq * kH * nkb)), dim3(256), 0, s, qkv, ld, ax, mk, bias_k,
                       bias_v, inv_freq, dout, stats, dqkv, dbias);
}
void launch32_rope_bwd(float* buf, long ntok, int ld, long pos_div, int pos_mod, const float* inv_freq, float qscale,
                       hipStream_t s) {
    const long total = ntok * kH * 12 * 2;
    hipLaunchKernelGGL(k32_rope_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, buf, ntok, ld, pos_div, pos_mod,
                       inv_freq, qscale);
}
void launch32_loss_grad(const float* pred, const float* target, const float* mask, long per_sample, long B, float* den,
                        float* dpred, hipStream_t s, float* scratch, size_t scratch_floats) {
    if (scratch && (size_t)(2 * B * ((per_sample + 8191) / 8192)) <= scratch_floats && per_sample > 8192)
        launch_masked_mse(nullptr, nullptr, mask, nullptr, per_sample, B, s, scratch, scratch_floats, den);   // two-stage mask sums
    else
        hipLaunchKernelGGL(k32_mask_sum, dim3((unsigned)B), dim3(256), 0, s, mask, per_sample, den);
    const long total = per_sample * B;
    hipLaunchKernelGGL(k32_loss_grad, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pred, target, mask, den,
                       per_sample, total, 1.0f / (float)B, dpred);
}
void launch32_sum_frames(const float* a, int B, int T, int L, float* out, hipStream_t s) {
    const long n = (long)B * L * kC;
    hipLaunchKernelGGL(k32_sum_frames, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, B, T, L, out);
}

}  // namespace mdg

namespace mdg {

// h[t][c] += gate[g(t)][c] * u[t][c]   (the residual update kept apart from the projection so that u can be taped)
__global__ void k32_gated_add(float* __restrict__ h, const float* __restrict__ u, long nrows, ModMap mm, int gate_chunk,
                              int gated) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * kC) return;
    const long t = i / kC;
    const int c = (int)(i % kC);
    const float g = gated ? mm.mod[mm.row_off(t) + gate_chunk * kC + c] : 1.0f;
    h[i] += g * u[i];
}
// h[t][c] = x[t][c] + gate[g(t)][c] * u[t][c]   (the deferred update of the training step's trunk, materialised: train.inc Pending)
__global__ void k32_gated_sum(float* __restrict__ h, const float* __restrict__ x, const float* __restrict__ u, long nrows, ModMap mm,
                              int gate_chunk) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * kC) return;
    const long t = i / kC;
    const int c = (int)(i % kC);
    h[i] = x[i] + mm.mod[mm.row_off(t) + gate_chunk * kC + c] * u[i];
}
void launch32_gated_sum(float* h, const float* x, const float* u, long nrows, const ModMap& mm, int gate_chunk, hipStream_t s) {
    const long n = nrows * kC;
    hipLaunchKernelGGL(k32_gated_sum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h, x, u, nrows, mm, gate_chunk);
}
void launch32_gated_add(float* h, const float* u, long nrows, const ModMap& mm, int gate_chunk, int gated, hipStream_t s) {
    const long n = nrows * kC;
    hipLaunchKernelGGL(k32_gated_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h, u, nrows, mm, gate_chunk, gated);
}

// ind0[t] = (cm[t] == 0), ind1[t] = (cm[t] != 0) as floats (mask_to_emb rows; latent_model.py:240-241)
__global__ void k32_indicator(const int64_t* __restrict__ cm, long n, float* __restrict__ ind0, float* __restrict__ ind1) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool one = cm[i] != 0;
    ind0[i] = one ? 0.f : 1.f;
    ind1[i] = one ? 1.f : 0.f;
}
void launch32_indicator(const int64_t* cm, long n, float* ind0, float* ind1, hipStream_t s) {
    hipLaunchKernelGGL(k32_indicator, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, cm, n, ind0, ind1);
}

// d aatype_to_emb[v][c] += sum over rows (b, l) with aatype == v of dx0[(g, l)][c]   (rows = ngroups * L, fixed order)
__global__ void k32_embed_rows_bwd(const float* __restrict__ dx0, const int64_t* __restrict__ aatype, int ngroups, int B, int L,
                                   float* __restrict__ dw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 21 * kC) return;
    const int v = i / kC, c = i % kC;
    float s = 0.f;
    for (long r = 0; r < (long)ngroups * L; ++r) {
        const int l = (int)(r % L), b = (int)((r / L) % B);
        if ((int)aatype[(long)b * L + l] == v) s += dx0[r * kC + c];
    }
    dw[i] += s;
}
void launch32_embed_rows_bwd(const float* dx0, const int64_t* aatype, int ngroups, int B, int L, float* dw, hipStream_t s) {
    hipLaunchKernelGGL(k32_embed_rows_bwd, dim3((21 * kC + 255) / 256), dim3(256), 0, s, dx0, aatype, ngroups, B, L, dw);
}

// TimestepEmbedder backward (layers.py:17-55 + the SiLU of the adaLN heads).  One block per time row r: recompute
// emb, pre1, h1, pre2; from dst = d silu(pre2): dpre2 -> scratch, h1 -> scratch, dpre1 -> scratch, emb -> scratch
// (the weight gradients are then plain dW / column-sum calls over the R rows).
__device__ __forceinline__ float silu_grad(float x) {
    const float s = 1.0f / (1.0f + expf(-x));
    return s * (1.0f + x * (1.0f - s));
}
__global__ __launch_bounds__(384) void k32_temb_bwd(const float* __restrict__ t_rows, float tmul, const float* __restrict__ w0,
                                                    const float* __restrict__ b0, const float* __restrict__ w2,
                                                    const float* __restrict__ b2, const float* __restrict__ dst,
                                                    float* __restrict__ emb_out, float* __restrict__ h1_out,
                                                    float* __restrict__ dpre1_out, float* __restrict__ dpre2_out) {
    __shared__ float emb[256];
    __shared__ float h1[kC];
    __shared__ float dp2[kC];
    const int r = blockIdx.x, c = threadIdx.x;
    const float t = t_rows[r] * tmul;
    if (c < 256) {
        const int i = c & 127;
        const float f = expf(-9.210340371976184f * (float)i / 128.0f);
        const float a = t * f;
        emb[c] = (c < 128) ? cosf(a) : sinf(a);
        emb_out[(long)r * 256 + c] = emb[c];
    }
    __syncthreads();
    float pre1 = b0[c];
    const float* wr = w0 + (long)c * 256;
    for (int i = 0; i < 256; ++i) pre1 += wr[i] * emb[i];
    h1[c] = pre1 / (1.0f + expf(-pre1));
    h1_out[(long)r * kC + c] = h1[c];
    __syncthreads();
    float pre2 = b2[c];
    const float* wr2 = w2 + (long)c * kC;
    for (int i = 0; i < kC; ++i) pre2 += wr2[i] * h1[i];
    dp2[c] = dst[(long)r * kC + c] * silu_grad(pre2);
    dpre2_out[(long)r * kC + c] = dp2[c];
    __syncthreads();
    float dh1 = 0.f;
    for (int o = 0; o < kC; ++o) dh1 += w2[(long)o * kC + c] * dp2[o];
    dpre1_out[(long)r * kC + c] = dh1 * silu_grad(pre1);
}
void launch32_temb_bwd(const float* t_rows, int nrows, float tmul, const float* w0, const float* b0, const float* w2,
                       const float* b2, const float* dst, float* emb, float* h1, float* dpre1, float* dpre2, hipStream_t s) {
    hipLaunchKernelGGL(k32_temb_bwd, dim3(nrows), dim3(384), 0, s, t_rows, tmul, w0, b0, w2, b2, dst, emb, h1, dpre1, dpre2);
}

}  // namespace mdg

namespace mdg {

// ---- invariant point attention backward (ipa.py:92-255 with c_z = 0; forward: k_ipa_attn / k_ipa_attn_tiled) ------------
// logit_ij = c_qk q_i.k_j + hwh_h sum_p |Qp_ip - Kp_jp|^2 + 1e5 (m_i m_j - 1),  a = softmax_j,
// o_i = sum_j a_ij v_j,  Op_i = sum_j a_ij Vp_j (global frame),  op_i = R_i^T (Op_i - t_i),  n_i = sqrt(|op_i|^2 + 1e-8).
// Query pass (thread = query): from d feat -> do, d Op (global), delta_i = do.o + dOp.Op; then over key tiles
//   dl_ij = a_ij (do.v_j + dOp.Vp_j - delta_i);  dq_i += c_qk dl_ij k_j;  dQp_i += 2 hwh dl_ij (Qp_i - Kp_j);  dhw_i += dl_ij d2_ij
// writes dproj (q columns, q-point columns rotated back to the local frame), dhw[token][head], and the per-query record
// qrec[token][head][49] = Qp (24) | dOp (24) | delta for the key pass.
constexpr int kIpaRec = 49;
__device__ __forceinline__ void stage_ipa_keys(const IpaAttnParams& p, long g, int b, int hd, int j0, int skey, int ssub,
                                               float (*sk)[32], float (*sv)[32], float (*skp)[24], float (*svp)[24], float* sm) {
    const int j = j0 + skey;
    const int jc = j < p.L ? j : p.L - 1;
    const float* pj = p.proj + (g * p.L + jc) * kIpaProj;
    const f32x4 kk = *reinterpret_cast<const f32x4*>(pj + 128 + hd * 64 + 4 * ssub);
    const f32x4 vv = *reinterpret_cast<const f32x4*>(pj + 128 + hd * 64 + 32 + 4 * ssub);
    float Rj[9], tj[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rj[k] = p.rot[((long)b * p.L + jc) * 9 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) tj[k] = p.trans[((long)b * p.L + jc) * 3 + k];
    float kx, ky, kz, vx, vy, vz;
    {
        const float x = pj[480 + hd * 16 + ssub], y = pj[544 + hd * 16 + ssub], z = pj[608 + hd * 16 + ssub];
        kx = Rj[0] * x + Rj[1] * y + Rj[2] * z + tj[0];
        ky = Rj[3] * x + Rj[4] * y + Rj[5] * z + tj[1];
        kz = Rj[6] * x + Rj[7] * y + Rj[8] * z + tj[2];
    }
    {
        const float x = pj[480 + hd * 16 + 8 + ssub], y = pj[544 + hd * 16 + 8 + ssub], z = pj[608 + hd * 16 + 8 + ssub];
        vx = Rj[0] * x + Rj[1] * y + Rj[2] * z + tj[0];
        vy = Rj[3] * x + Rj[4] * y + Rj[5] * z + tj[1];
        vz = Rj[6] * x + Rj[7] * y + Rj[8] * z + tj[2];
    }
    *reinterpret_cast<f32x4*>(&sk[skey][4 * ssub]) = kk;
    *reinterpret_cast<f32x4*>(&sv[skey][4 * ssub]) = vv;
    skp[skey][3 * ssub] = kx; skp[skey][3 * ssub + 1] = ky; skp[skey][3 * ssub + 2] = kz;
    svp[skey][3 * ssub] = vx; svp[skey][3 * ssub + 1] = vy; svp[skey][3 * ssub + 2] = vz;
    if (ssub == 0) sm[skey] = j < p.L ? p.mask_bl[(long)b * p.L + jc] : -1.0f;
}

struct IpaBwdParams {
    IpaAttnParams f;          // forward inputs (proj, rot, trans, mask_bl, head_w, stats = lse, feat32 = forward features)
    const float* dfeat;       // [M][256]
    float* dproj;             // [M][672]
    float* dhw;               // [M][4]
    float* qrec;              // [M][4][49]
    // nsplit > 1: the key loop of the query pass / the query loop of the key pass is cut into nsplit slices (blockIdx.y) whose
    // partial sums land in part[slice][M][672 + 4] (dproj row | dhw) and are added in slice order by k32_ipa_bwd_reduce: with
    // one thread per query / key and B * 4 heads that is 4 workgroups at B = 1 -- 303 us of one wave per SIMD
    int nsplit;
    float* part;
};
constexpr int kIpaPartRow = kIpaProj + 4;

__global__ __launch_bounds__(256) void k32_ipa_bwd_q(const IpaBwdParams bp) {
    const IpaAttnParams& p = bp.f;
    __shared__ __attribute__((aligned(16))) float sk[kIpaKT][32];
    __shared__ __attribute__((aligned(16))) float sv[kIpaKT][32];
    __shared__ __attribute__((aligned(16))) float skp[kIpaKT][24];
    __shared__ __attribute__((aligned(16))) float svp[kIpaKT][24];
    __shared__ float sm[kIpaKT];
    const int nqt = (p.L + 255) / 256;
    const int qt = blockIdx.x % nqt;
    const int hd = (blockIdx.x / nqt) & 3;
    const long g = blockIdx.x / (nqt * 4);
    const int b = (int)(g % p.B);
    const int tid = threadIdx.x;
    const int i = qt * 256 + tid;
    const bool qok = i < p.L;
    const int ic = qok ? i : p.L - 1;
    const long gi = g * p.L + ic;
    const float* pi = p.proj + gi * kIpaProj;
    float Ri[9], ti[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) Ri[k] = p.rot[((long)b * p.L + ic) * 9 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) ti[k] = p.trans[((long)b * p.L + ic) * 3 + k];
    const float mi = p.mask_bl[(long)b * p.L + ic];
    const float qk_scale = 0.10206207261596575f;
    const float hwraw = p.head_w[hd];
    const float sp = (hwraw > 20.f) ? hwraw : log1pf(expf(hwraw));
    const float hwh = -0.5f * sp * 0.09622504486493763f;
    const float lse = p.stats[gi * 4 + hd];
    float q[32], qp[8][3], dO[32], dOp[8][3], dq[32], dqp[8][3];
    const float* fi = p.feat32 + gi * kIpaFeat;
    const float* dfi = bp.dfeat + gi * kIpaFeat;
    float delta = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        q[c] = pi[hd * 32 + c] * qk_scale;
        dO[c] = dfi[hd * 32 + c];
        delta += dO[c] * fi[hd * 32 + c];
        dq[c] = 0.f;
    }
#pragma unroll
    for (int pt = 0; pt < 8; ++pt) {
        const float x = pi[384 + hd * 8 + pt], y = pi[416 + hd * 8 + pt], z = pi[448 + hd * 8 + pt];
        qp[pt][0] = Ri[0] * x + Ri[1] * y + Ri[2] * z + ti[0];
        qp[pt][1] = Ri[3] * x + Ri[4] * y + Ri[5] * z + ti[1];
        qp[pt][2] = Ri[6] * x + Ri[7] * y + Ri[8] * z + ti[2];
        // local output point and its norm (forward features), d local = d feat + d norm * op / norm
        const float lx = fi[128 + hd * 8 + pt], ly = fi[160 + hd * 8 + pt], lz = fi[192 + hd * 8 + pt];
        const float nr = fi[224 + hd * 8 + pt];
        const float dn = dfi[224 + hd * 8 + pt] / nr;
        const float dlx = dfi[128 + hd * 8 + pt] + dn * lx, dly = dfi[160 + hd * 8 + pt] + dn * ly,
                    dlz = dfi[192 + hd * 8 + pt] + dn * lz;
        // global: Op = R op + t,  dOp = R d op
        const float Ox = Ri[0] * lx + Ri[1] * ly + Ri[2] * lz + ti[0], Oy = Ri[3] * lx + Ri[4] * ly + Ri[5] * lz + ti[1],
                    Oz = Ri[6] * lx + Ri[7] * ly + Ri[8] * lz + ti[2];
        dOp[pt][0] = Ri[0] * dlx + Ri[1] * dly + Ri[2] * dlz;
        dOp[pt][1] = Ri[3] * dlx + Ri[4] * dly + Ri[5] * dlz;
        dOp[pt][2] = Ri[6] * dlx + Ri[7] * dly + Ri[8] * dlz;
        delta += dOp[pt][0] * Ox + dOp[pt][1] * Oy + dOp[pt][2] * Oz;
        dqp[pt][0] = dqp[pt][1] = dqp[pt][2] = 0.f;
    }
    float dhw = 0.f;
    const int skey = tid >> 3, ssub = tid & 7;
    const int sl = blockIdx.y;
    const int per = ((p.L + kIpaKT - 1) / kIpaKT + bp.nsplit - 1) / bp.nsplit * kIpaKT;    // whole tiles per slice
    const int jlo = sl * per, jhi = jlo + per < p.L ? jlo + per : p.L;
    for (int j0 = jlo; j0 < jhi; j0 += kIpaKT) {
        __syncthreads();
        stage_ipa_keys(p, g, b, hd, j0, skey, ssub, sk, sv, skp, svp, sm);
        __syncthreads();
#pragma unroll 2
        for (int jj = 0; jj < kIpaKT; ++jj) {
            const float mj = sm[jj];
            if (mj < 0.f) continue;
            float dot = 0.f, da = 0.f, d2 = 0.f;
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                dot += q[c] * sk[jj][c];
                da += dO[c] * sv[jj][c];
            }
            float dif[8][3];
#pragma unroll
            for (int pt = 0; pt < 8; ++pt) {
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    dif[pt][x] = qp[pt][x] - skp[jj][3 * pt + x];
                    d2 += dif[pt][x] * dif[pt][x];
                    da += dOp[pt][x] * svp[jj][3 * pt + x];
                }
            }
            const float logit = dot + hwh * d2 + 1e5f * (mi * mj - 1.0f);
            const float dl = expf(logit - lse) * (da - delta);
#pragma unroll
            for (int c = 0; c < 32; ++c) dq[c] += dl * sk[jj][c];
            const float w2 = 2.0f * hwh * dl;
#pragma unroll
            for (int pt = 0; pt < 8; ++pt)
#pragma unroll
                for (int x = 0; x < 3; ++x) dqp[pt][x] += w2 * dif[pt][x];
            dhw += dl * d2;
        }
    }
    if (!qok) return;
    const long mtot = (long)p.ngroups * p.L;
    float* dp = bp.nsplit > 1 ? bp.part + ((long)sl * mtot + gi) * kIpaPartRow : bp.dproj + gi * kIpaProj;
#pragma unroll
    for (int c = 0; c < 32; ++c) dp[hd * 32 + c] = dq[c] * qk_scale;
    float* rec = bp.qrec + (gi * 4 + hd) * kIpaRec;    // the same values from every slice
#pragma unroll
    for (int pt = 0; pt < 8; ++pt) {
        // local q-point gradient: R^T d Qp
        dp[384 + hd * 8 + pt] = Ri[0] * dqp[pt][0] + Ri[3] * dqp[pt][1] + Ri[6] * dqp[pt][2];
        dp[416 + hd * 8 + pt] = Ri[1] * dqp[pt][0] + Ri[4] * dqp[pt][1] + Ri[7] * dqp[pt][2];
        dp[448 + hd * 8 + pt] = Ri[2] * dqp[pt][0] + Ri[5] * dqp[pt][1] + Ri[8] * dqp[pt][2];
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            rec[3 * pt + x] = qp[pt][x];
            rec[24 + 3 * pt + x] = dOp[pt][x];
        }
    }
    rec[48] = delta;
    if (bp.nsplit > 1) dp[kIpaProj + hd] = dhw;
    else bp.dhw[gi * 4 + hd] = dhw;
}

// Key pass (thread = key j): over query tiles  dk_j += c_qk dl_ij q_i;  dv_j += a_ij do_i;  dVp_j += a_ij dOp_i;
// dKp_j -= 2 hwh dl_ij (Qp_i - Kp_j);  then points back to the local frame of j.
__global__ __launch_bounds__(256) void k32_ipa_bwd_kv(const IpaBwdParams bp) {
    const IpaAttnParams& p = bp.f;
    constexpr int QT = 32;
    __shared__ __attribute__((aligned(16))) float sq[QT][32];
    __shared__ __attribute__((aligned(16))) float sdo[QT][32];
    __shared__ float srec[QT][kIpaRec + 3];   // Qp | dOp | delta | lse | m_i | valid
    const int nkt = (p.L + 255) / 256;
    const int kt = blockIdx.x % nkt;
    const int hd = (blockIdx.x / nkt) & 3;
    const long g = blockIdx.x / (nkt * 4);
    const int b = (int)(g % p.B);
    const int tid = threadIdx.x;
    const int j = kt * 256 + tid;
    const bool kok = j < p.L;
    const int jc = kok ? j : p.L - 1;
    const long gj = g * p.L + jc;
    const float* pj = p.proj + gj * kIpaProj;
    float Rj[9], tj[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rj[k] = p.rot[((long)b * p.L + jc) * 9 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) tj[k] = p.trans[((long)b * p.L + jc) * 3 + k];
    const float mj = p.mask_bl[(long)b * p.L + jc];
    const float qk_scale = 0.10206207261596575f;
    const float hwraw = p.head_w[hd];
    const float sp = (hwraw > 20.f) ? hwraw : log1pf(expf(hwraw));
    const float hwh = -0.5f * sp * 0.09622504486493763f;
    float k[32], v[32], kp[8][3], vp[8][3], dk[32], dv[32], dkp[8][3], dvp[8][3];
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        k[c] = pj[128 + hd * 64 + c];
        v[c] = pj[128 + hd * 64 + 32 + c];
        dk[c] = dv[c] = 0.f;
    }
#pragma unroll
    for (int pt = 0; pt < 8; ++pt) {
        {
            const float x = pj[480 + hd * 16 + pt], y = pj[544 + hd * 16 + pt], z = pj[608 + hd * 16 + pt];
            kp[pt][0] = Rj[0] * x + Rj[1] * y + Rj[2] * z + tj[0];
            kp[pt][1] = Rj[3] * x + Rj[4] * y + Rj[5] * z + tj[1];
            kp[pt][2] = Rj[6] * x + Rj[7] * y + Rj[8] * z + tj[2];
        }
        {
            const float x = pj[480 + hd * 16 + 8 + pt], y = pj[544 + hd * 16 + 8 + pt], z = pj[608 + hd * 16 + 8 + pt];
            vp[pt][0] = Rj[0] * x + Rj[1] * y + Rj[2] * z + tj[0];
            vp[pt][1] = Rj[3] * x + Rj[4] * y + Rj[5] * z + tj[1];
            vp[pt][2] = Rj[6] * x + Rj[7] * y + Rj[8] * z + tj[2];
        }
#pragma unroll
        for (int x = 0; x < 3; ++x) dkp[pt][x] = dvp[pt][x] = 0.f;
    }
    const int sl = blockIdx.y;
    const int per = ((p.L + QT - 1) / QT + bp.nsplit - 1) / bp.nsplit * QT;
    const int ilo = sl * per, ihi = ilo + per < p.L ? ilo + per : p.L;
    for (int i0 = ilo; i0 < ihi; i0 += QT) {
        __syncthreads();
        for (int e = tid; e < QT * 32; e += 256) {
            const int ii = e >> 5, c = e & 31;
            const int i = i0 + ii;
            const long gi = g * p.L + (i < p.L ? i : p.L - 1);
            sq[ii][c] = p.proj[gi * kIpaProj + hd * 32 + c] * qk_scale;
            sdo[ii][c] = bp.dfeat[gi * kIpaFeat + hd * 32 + c];
        }
        for (int e = tid; e < QT * (kIpaRec + 3); e += 256) {
            const int ii = e / (kIpaRec + 3), c = e % (kIpaRec + 3);
            const int i = i0 + ii;
            const int icl = i < p.L ? i : p.L - 1;
            const long gi = g * p.L + icl;
            float val;
            if (c < kIpaRec) val = bp.qrec[(gi * 4 + hd) * kIpaRec + c];
            else if (c == kIpaRec) val = p.stats[gi * 4 + hd];
            else if (c == kIpaRec + 1) val = p.mask_bl[(long)b * p.L + icl];
            else val = i < p.L ? 1.f : 0.f;
            srec[ii][c] = val;
        }
        __syncthreads();
#pragma unroll 2
        for (int ii = 0; ii < QT; ++ii) {
            if (srec[ii][kIpaRec + 2] == 0.f) continue;
            float dot = 0.f, da = 0.f, d2 = 0.f;
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                dot += sq[ii][c] * k[c];
                da += sdo[ii][c] * v[c];
            }
            float dif[8][3];
#pragma unroll
            for (int pt = 0; pt < 8; ++pt)
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    dif[pt][x] = srec[ii][3 * pt + x] - kp[pt][x];
                    d2 += dif[pt][x] * dif[pt][x];
                    da += srec[ii][24 + 3 * pt + x] * vp[pt][x];
                }
            const float logit = dot + hwh * d2 + 1e5f * (srec[ii][kIpaRec + 1] * mj - 1.0f);
            const float a = expf(logit - srec[ii][kIpaRec]);
            const float dl = a * (da - srec[ii][48]);
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                dk[c] += dl * sq[ii][c];
                dv[c] += a * sdo[ii][c];
            }
            const float w2 = -2.0f * hwh * dl;
#pragma unroll
            for (int pt = 0; pt < 8; ++pt)
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    dkp[pt][x] += w2 * dif[pt][x];
                    dvp[pt][x] += a * srec[ii][24 + 3 * pt + x];
                }
        }
    }
    if (!kok) return;
    float* dp = bp.nsplit > 1 ? bp.part + ((long)sl * p.ngroups * p.L + gj) * kIpaPartRow : bp.dproj + gj * kIpaProj;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        dp[128 + hd * 64 + c] = dk[c];          // sq already carries c_qk
        dp[128 + hd * 64 + 32 + c] = dv[c];
    }
#pragma unroll
    for (int pt = 0; pt < 8; ++pt) {
        dp[480 + hd * 16 + pt] = Rj[0] * dkp[pt][0] + Rj[3] * dkp[pt][1] + Rj[6] * dkp[pt][2];
        dp[544 + hd * 16 + pt] = Rj[1] * dkp[pt][0] + Rj[4] * dkp[pt][1] + Rj[7] * dkp[pt][2];
        dp[608 + hd * 16 + pt] = Rj[2] * dkp[pt][0] + Rj[5] * dkp[pt][1] + Rj[8] * dkp[pt][2];
        dp[480 + hd * 16 + 8 + pt] = Rj[0] * dvp[pt][0] + Rj[3] * dvp[pt][1] + Rj[6] * dvp[pt][2];
        dp[544 + hd * 16 + 8 + pt] = Rj[1] * dvp[pt][0] + Rj[4] * dvp[pt][1] + Rj[7] * dvp[pt][2];
        dp[608 + hd * 16 + 8 + pt] = Rj[2] * dvp[pt][0] + Rj[5] * dvp[pt][1] + Rj[8] * dvp[pt][2];
    }
}

// d head_weights[h] += (-1/2 sqrt(1/108)) sigmoid(w_h) * sum_tokens dhw[token][h]    (softplus' = sigmoid)
__global__ void k32_ipa_headw_bwd(const float* __restrict__ dhw, long ntok, const float* __restrict__ head_w,
                                  float* __restrict__ g) {
    const int hd = threadIdx.x;
    if (hd >= 4) return;
    float s = 0.f;
    for (long t = 0; t < ntok; ++t) s += dhw[t * 4 + hd];
    const float w = head_w[hd];
    const float sig = w > 20.f ? 1.0f : 1.0f / (1.0f + expf(-w));
    g[hd] += s * (-0.5f * 0.09622504486493763f) * sig;
}

// dproj[row][0:672], dhw[row][0:4] = sum over slices (fixed order) of part[slice][row][676]
__global__ void k32_ipa_bwd_reduce(const float* __restrict__ part, int nsplit, long mtot, float* __restrict__ dproj,
                                   float* __restrict__ dhw) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= mtot * kIpaPartRow) return;
    const long row = i / kIpaPartRow;
    const int c = (int)(i % kIpaPartRow);
    float v = 0.f;
    for (int z = 0; z < nsplit; ++z) v += part[(long)z * mtot * kIpaPartRow + i];
    if (c < kIpaProj) dproj[row * kIpaProj + c] = v;
    else dhw[row * 4 + c - kIpaProj] = v;
}
void launch32_ipa_bwd(const IpaAttnParams& f, const float* dfeat, float* dproj, float* dhw, float* qrec, float* dheadw,
                      hipStream_t s, float* part, size_t part_floats) {
    const int nqt = (f.L + 255) / 256;
    const long nblk = (long)f.ngroups * 4 * nqt;
    const long mtot = (long)f.ngroups * f.L;
    // enough slices for ~64 workgroups, at most one 32-row tile per slice, within the partial buffer
    int nsplit = (int)((64 + nblk - 1) / nblk);
    const int ntile = (f.L + 31) / 32;
    if (nsplit > ntile) nsplit = ntile;
    if (nsplit > 16) nsplit = 16;
    while (nsplit > 1 && (!part || (size_t)nsplit * mtot * kIpaPartRow > part_floats)) --nsplit;
    IpaBwdParams bp{f, dfeat, dproj, dhw, qrec, nsplit, part};
    hipLaunchKernelGGL(k32_ipa_bwd_q, dim3((unsigned)nblk, (unsigned)nsplit), dim3(256), 0, s, bp);
    hipLaunchKernelGGL(k32_ipa_bwd_kv, dim3((unsigned)nblk, (unsigned)nsplit), dim3(256), 0, s, bp);
    if (nsplit > 1) {
        const long n = mtot * kIpaPartRow;
        hipLaunchKernelGGL(k32_ipa_bwd_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, nsplit, mtot, dproj, dhw);
    }
    if (dheadw) hipLaunchKernelGGL(k32_ipa_headw_bwd, dim3(1), dim3(64), 0, s, dhw, (long)f.ngroups * f.L, f.head_w, dheadw);
}

}  // namespace mdg
