// panel.h -- device building blocks of the "resident A-panel" GEMM family (gfx950).
//
// Design (DESIGN.md "GEMM family"): a workgroup owns a panel of 64 token rows.  The panel's
// activations are normalised/modulated once and kept in LDS as bf16 [64][K] with an XOR swizzle
// that makes the MFMA fragment reads (ds_read_b128) bank-conflict free.  Weights are pre-packed
// at load time into MFMA fragment order, so each wave streams its own weight slab straight from
// L2 into VGPRs with fully coalesced 1 KiB loads -- no LDS staging and no barriers in the K loop.
#pragma once
#include "common.h"

namespace mdg {

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also carries a workgroup-scope fence over
// GLOBAL memory, i.e. s_waitcnt vmcnt(0): a wave that has read-modify-write stores or prefetches in flight would
// sit at the barrier until HBM has acknowledged them, and every other wave with it.  Use where the waves
// exchange data through LDS only.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- LDS panel addressing -----------------------------------------------------------------
// Row r holds K bf16 (ROWB = 2K bytes, a multiple of 256).  Byte offset b within the row is
// stored at b ^ ((r & 15) << 4): the 16 rows a ds_read_b128 lane-group touches land on 16
// distinct 16-byte slots of the 256-byte bank row (MI355X_MICROARCH LDS table).
__device__ __forceinline__ int panel_off(int row, int byte_in_row, int rowb) {
    return row * rowb + (byte_in_row ^ ((row & 15) << 4));
}

// MFMA operand fragment (32 rows x 16 k) for k-step ks from the panel: lane -> row (lane&31),
// k = ks*16 + (lane>>5)*8 .. +7  (the same k mapping the packed weights use).
__device__ __forceinline__ bf16x8 panel_frag(const unsigned char* panel, const int rowb, int tile, int ks) {
    const int lane = lane_id();
    const int row = tile * 32 + (lane & 31);
    return *reinterpret_cast<const bf16x8*>(panel + panel_off(row, ks * 32 + (lane >> 5) * 16, rowb));
}

// ---- wave-level GEMM ------------------------------------------------------------------------
// TT token tiles (32 rows of the panel each) x FT feature tiles (32 weight rows each), K = 16*KSTEPS.
// wfrag points at this lane's 16 bytes of the first feature tile's first k-step; feature tile f,
// k-step k lives (f*wtile_stride + k*64) bf16x8 further (packed layout [ftile][kstep][lane][8]).
// TRANS = false: acc[tt*FT+ft] = D[token][feature]  (A = activations, B = weights)
// TRANS = true : acc[ft*TT+tt] = D[feature][token]  (A = weights,      B = activations)
// ---- wave-level GEMM, software-pipelined --------------------------------------------------------------
// hipcc's scheduler sinks loads down to their first use (prefetch distance 0: every k-step then eats a full
// L2 round trip).  The pipeline below is therefore pinned with sched_barrier(0) fences: the weight fragments of
// k-step ks+PF are requested (L2 -> VGPR) and the activation fragments of k-step ks+1 are read from LDS in a
// scheduling region that precedes the MFMAs of k-step ks.  The loads stay ordinary loads, so hipcc itself
// inserts exact counted waits (vmcnt(PF*FT)) -- no hand-counted asm waits, no stale-register hazards.
template <int TT, int FT, int KSTEPS, bool TRANS, int PF = 4>
__device__ __forceinline__ void wave_gemm(const unsigned char* panel, const int rowb, const int tile0, const int ks0,
                                          const bf16x8* __restrict__ wfrag, const int wtile_stride,
                                          f32x16* acc) {
    static_assert(PF < KSTEPS, "prefetch depth");
    bf16x8 wring[PF + 1][FT];
    bf16x8 aring[2][TT];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int f = 0; f < FT; ++f) wring[p][f] = wfrag[(size_t)f * wtile_stride + p * 64];
#pragma unroll
    for (int t = 0; t < TT; ++t) aring[0][t] = panel_frag(panel, rowb, tile0 + t, ks0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        if (ks + PF < KSTEPS) {
#pragma unroll
            for (int f = 0; f < FT; ++f)
                wring[(ks + PF) % (PF + 1)][f] = wfrag[(size_t)f * wtile_stride + (ks + PF) * 64];
        }
        if (ks + 1 < KSTEPS) {
#pragma unroll
            for (int t = 0; t < TT; ++t) aring[(ks + 1) & 1][t] = panel_frag(panel, rowb, tile0 + t, ks0 + ks + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int f = 0; f < FT; ++f) {
                if (TRANS)
                    acc[f * TT + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wring[ks % (PF + 1)][f], aring[ks & 1][t],
                                                                             acc[f * TT + t], 0, 0, 0);
                else
                    acc[t * FT + f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aring[ks & 1][t], wring[ks % (PF + 1)][f],
                                                                             acc[t * FT + f], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int N>
__device__ __forceinline__ void zero_acc(f32x16* acc) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = opaque_zero();
}

// ---- panel row table (LDS) ----------------------------------------------------------------------
// tok[r]  = global token row of panel row r, or -1 for padding rows
// moff[r] = offset (floats) of that token's adaLN table row inside ModMap::mod
// uniform = that offset when every valid row of the panel shares it (the usual case: a panel rarely straddles two
//           samples), else -1: prologues / epilogues then read the modulation vectors once instead of per row
struct PanelRows {
    int tok[kPanel];
    int moff[kPanel];
    int uniform;
    int pad_[3];   // keeps the panel that follows 16-byte aligned
};

// called by the ONE full wave that fills the table (tid 0..63), with its (token, offset) pair
__device__ __forceinline__ void set_uniform(PanelRows* pr, int tid, long t, long mo) {
    const int mo0 = __builtin_amdgcn_readfirstlane((int)mo);
    const int t0 = __builtin_amdgcn_readfirstlane((int)t);
    const bool same = t < 0 || (int)mo == mo0;
    const bool all = __builtin_amdgcn_ballot_w64(same) == ~0ull;
    if (tid == 0) pr->uniform = (t0 >= 0 && all) ? mo0 : -1;
}

// rows = positions pos0 .. pos0+63 of sequence `seq` along an attention axis
// (nrows < 64: the panel's rows from nrows on are padding rows -- the 32-row workgroups of the small launches)
__device__ __forceinline__ void setup_rows_axis(PanelRows* pr, const AxisMap ax, int seq, int pos0, const ModMap mm, int nrows = kPanel) {
    if (threadIdx.x < kPanel) {
        const int pos = pos0 + threadIdx.x;
        long t = -1, mo = 0;
        if (pos < ax.len && (int)threadIdx.x < nrows) {
            t = ax.token(seq, pos);
            mo = mm.row_off(t);
        }
        pr->tok[threadIdx.x] = (int)t;
        pr->moff[threadIdx.x] = (int)mo;
        set_uniform(pr, threadIdx.x, t, mo);
    }
}

// rows = natural token order row0 .. row0+63 (clipped at nrows)
// `tid`: index of the calling thread within the group of threads that fills the table (default: the workgroup)
__device__ __forceinline__ void setup_rows_linear(PanelRows* pr, long row0, long nrows, const ModMap mm,
                                                  int tid = threadIdx.x) {
    if (tid >= 0 && tid < kPanel) {
        long t = row0 + tid, mo = 0;
        if (t < nrows) mo = mm.row_off(t); else t = -1;
        pr->tok[tid] = (int)t;
        pr->moff[tid] = (int)mo;
        set_uniform(pr, tid, t, mo);
    }
}

// ---- panel prologues --------------------------------------------------------------------------
// LayerNorm (no affine, eps) + adaLN modulate  y = LN(x)*(1+scale)+shift   (layers.py:14-15), or
// affine LayerNorm y = LN(x)*gamma+beta (AFFINE; nn.LayerNorm of IPALayer.ipa_norm, eps 1e-5).
// 256 threads; wave w normalises rows 16 b + 4 w + j (b, j < 4); a row is 384 fp32 = 6 per lane as 3 float2.
// `tok[r]` (in LDS) = global token row of panel row r, or -1 for padding rows (written as zeros).
// Memory-level parallelism: each wave normalises FOUR rows per iteration and every global load is
// UNCONDITIONAL (padding rows read token 0 and are written as zeros).  A load under `if (t >= 0)` makes hipcc
// merge old and new register values right behind it, i.e. wait for the data it has just requested -- with one
// row per iteration that exposed a full HBM round trip per row, 16 times per wave.
// B0..B1: the batches (of 4 rows per wave) handled by this call, so a caller can split the prologue in parts;
// w: index (0..3) of the calling wave among the four that share the panel.
template <bool AFFINE, int B0, int B1, bool UNI>
__device__ __forceinline__ void prologue_ln_impl(unsigned char* panel, const PanelRows* pr, const float* __restrict__ x,
                                            const ModMap mm, int shift_chunk, int scale_chunk, float eps,
                                            const int w, const int lane, const int um) {
    constexpr int ROWB = kC * 2;
    constexpr int RB = 4, NB = kPanel / (4 * RB);   // a wave owns NB batches of RB rows: rows 16 b + 4 w + j
    static_assert(0 <= B0 && B0 < B1 && B1 <= NB, "batch range");
    // One batch of RB rows in flight at a time.  (Requesting all 16 rows of a wave up front was measured SLOWER
    // -- prologue 28k -> 34k cycles: every workgroup of the launch is in its prologue at the same time and the
    // bigger burst only deepens the HBM queue.)
    int t[NB][RB];
    f32x2 v[NB][RB][3];
    f32x2 scu[3], shu[3];   // UNI: the one modulation row all rows of the panel share
    if (UNI) {
        const unsigned char* mb = reinterpret_cast<const unsigned char*>(mm.mod);
        const unsigned osh = ((unsigned)um + (unsigned)(shift_chunk * kC)) * 4u + (unsigned)lane * 8u;
        const unsigned osc = ((unsigned)um + (unsigned)(scale_chunk * kC)) * 4u + (unsigned)lane * 8u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            scu[i] = *reinterpret_cast<const f32x2*>(mb + osc + 512u * i);
            shu[i] = *reinterpret_cast<const f32x2*>(mb + osh + 512u * i);
        }
    }
#pragma unroll
    for (int b = B0; b < B1; ++b) {
        const int r0 = 4 * RB * b + RB * w;
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            t[b][j] = pr->tok[r0 + j];
            // uniform base (SGPR pair) + 32-bit byte offset: no 64-bit address VGPRs to keep alive or spill
            const unsigned off = (unsigned)(t[b][j] < 0 ? 0 : t[b][j]) * (unsigned)(kC * 4) + (unsigned)lane * 8u;
            const unsigned char* xb = reinterpret_cast<const unsigned char*>(x);
#pragma unroll
            for (int i = 0; i < 3; ++i) v[b][j][i] = *reinterpret_cast<const f32x2*>(xb + off + 512u * i);
        }
        f32x2 sc[RB][3], sh[RB][3];
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            if (UNI) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    sc[j][i] = scu[i];
                    sh[j][i] = shu[i];
                }
            } else {
                const unsigned mo = AFFINE ? 0u : (unsigned)pr->moff[r0 + j];   // moff == 0 for padding rows
                const unsigned char* mb = reinterpret_cast<const unsigned char*>(mm.mod);
                const unsigned osh = (mo + (unsigned)(shift_chunk * kC)) * 4u + (unsigned)lane * 8u;
                const unsigned osc = (mo + (unsigned)(scale_chunk * kC)) * 4u + (unsigned)lane * 8u;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    sc[j][i] = *reinterpret_cast<const f32x2*>(mb + osc + 512u * i);
                    sh[j][i] = *reinterpret_cast<const f32x2*>(mb + osh + 512u * i);
                }
            }
        }
        float mean[RB], rstd[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const float s = (v[b][j][0][0] + v[b][j][0][1]) + (v[b][j][1][0] + v[b][j][1][1]) + (v[b][j][2][0] + v[b][j][2][1]);
            mean[j] = wave_sum(s) * (1.0f / kC);
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                v[b][j][i][0] -= mean[j];
                v[b][j][i][1] -= mean[j];
                q += v[b][j][i][0] * v[b][j][i][0] + v[b][j][i][1] * v[b][j][i][1];
            }
            rstd[j] = 1.0f / sqrtf(wave_sum(q) * (1.0f / kC) + eps);
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float m0 = AFFINE ? sc[j][i][0] : 1.0f + sc[j][i][0];
                const float m1 = AFFINE ? sc[j][i][1] : 1.0f + sc[j][i][1];
                const float y0 = v[b][j][i][0] * rstd[j] * m0 + sh[j][i][0];
                const float y1 = v[b][j][i][1] * rstd[j] * m1 + sh[j][i][1];
                *reinterpret_cast<uint32_t*>(panel + panel_off(r0 + j, 4 * lane + 256 * i, ROWB)) =
                    t[b][j] >= 0 ? pack_bf16(y0, y1) : 0u;
            }
        }
    }
}

template <bool AFFINE, int B0 = 0, int B1 = 4>
__device__ __forceinline__ void prologue_ln(unsigned char* panel, const PanelRows* pr, const float* __restrict__ x,
                                            const ModMap mm, int shift_chunk, int scale_chunk, float eps,
                                            const int w = wave_id(), const int lane = lane_id()) {
    const int um = AFFINE ? 0 : pr->uniform;   // wave-uniform (LDS broadcast); AFFINE: gamma/beta are one row
    if (um >= 0) prologue_ln_impl<AFFINE, B0, B1, true>(panel, pr, x, mm, shift_chunk, scale_chunk, eps, w, lane, um);
    else prologue_ln_impl<AFFINE, B0, B1, false>(panel, pr, x, mm, shift_chunk, scale_chunk, eps, w, lane, 0);
}

// Plain bf16 rows [token][K] -> panel (K = 384 or 256).  16-byte chunks, 256 threads.
template <int K>
__device__ __forceinline__ void prologue_bf16(unsigned char* panel, const PanelRows* pr, const __bf16* __restrict__ src) {
    constexpr int ROWB = K * 2;
    constexpr int CPR = ROWB / 16;  // 16-byte chunks per row
    constexpr int NIT = kPanel * CPR / 256;
    static_assert(kPanel * CPR % 256 == 0, "whole iterations");
    // all of a thread's loads are issued back to back and unconditionally (padding rows read token 0)
    u32x4 v[NIT];
    int tk[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + 256 * it;
        const int r = i / CPR, c = i % CPR;
        tk[it] = pr->tok[r];
        v[it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(src) +
                                                (long)(tk[it] < 0 ? 0 : tk[it]) * ROWB + c * 16);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + 256 * it;
        const int r = i / CPR, c = i % CPR;
        *reinterpret_cast<u32x4*>(panel + panel_off(r, c * 16, ROWB)) = tk[it] >= 0 ? v[it] : u32x4{0u, 0u, 0u, 0u};
    }
}

// Epilogue: residual update  h[token][col] += gate[col] * (acc + bias[col])   (latent_model.py:462,476,481)
// acc tile is D[token][feature] (non-transposed).  gate == nullptr -> 1 (IPA linear_out residual).
template <int TT, int FT>
__device__ __forceinline__ void epilogue_gate_residual(const f32x16* acc, const PanelRows* pr, int col0,
                                                       const float* __restrict__ bias, const ModMap mm,
                                                       int gate_chunk, bool gated, float* __restrict__ h) {
    const int lane = lane_id();
    const int hh = lane >> 5, n = lane & 31;
    float b[FT];
#pragma unroll
    for (int f = 0; f < FT; ++f) b[f] = bias[col0 + f * 32 + n];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = t * 32 + mfma_row(r, hh);
            const int tk = pr->tok[row];
            if (tk >= 0) {
                float* hp = h + (long)tk * kC + col0 + n;
                const float* gp = mm.mod + pr->moff[row] + gate_chunk * kC + col0 + n;
#pragma unroll
                for (int f = 0; f < FT; ++f) {
                    const float g = gated ? gp[f * 32] : 1.0f;
                    hp[f * 32] = hp[f * 32] + g * (acc[t * FT + f][r] + b[f]);
                }
            }
        }
    }
}

// Same update, staged through a wave-private 12 KiB LDS slab so that the read-modify-write of h uses 16-byte
// accesses on whole 384-byte row segments: the direct form above issues 288 dword memory instructions per
// lane and measured 100 of the 128 us of the out-projection kernel; this one issues 32 loads + 32 stores.
// `stage` = this wave's slab ([32 rows][96 cols] fp32); the caller guarantees nothing else uses it meanwhile.
// Two steps per 32-row half t of the panel:
//   epi_stage: accumulator tiles acc[t*3 .. t*3+2] (MFMA layout) -> slab (row-major)
//   epi_rmw  : slab -> h[token][col0 .. col0+95] += gate * (slab + bias), 48 lanes x float4 = two rows per step
__device__ __forceinline__ void epi_stage(const f32x16* acc3, float* stage) {
    const int lane = lane_id();
    const int hh = lane >> 5, n = lane & 31;
#pragma unroll
    for (int f = 0; f < 3; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[mfma_row(r, hh) * 96 + f * 32 + n] = acc3[f][r];
}

// The loads of BR row pairs at a time are UNCONDITIONAL (padding rows and idle lanes read token 0) and issued
// back to back; only the stores are predicated.  With a load under `if (tk >= 0)` every row paid its own HBM
// round trip, 32 in sequence per wave.
// prd / hd (optional): the updated rows go to hd[prd->tok[row]] instead of back to h[pr->tok[row]] (k_mlp8's split form: rows read
// from / written to a workgroup-private copy of the panel's residual rows); prd marks the same rows valid as pr.
template <int BR>
__device__ __forceinline__ void epi_rmw(const int t, const PanelRows* pr, const float* stage, int col0,
                                        const float* __restrict__ bias, const ModMap mm, int gate_chunk, bool gated,
                                        float* __restrict__ h, const PanelRows* prd = nullptr, float* hd = nullptr) {
    static_assert(16 % BR == 0, "batches of row pairs");
    const int lane = lane_id();
    const int q = lane % 24, r2 = lane / 24;
    const bool active = lane < 48;
    const int slot = active ? lane : 0;   // lanes 48..63 idle along: keep their LDS/global addresses in range
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + col0 + 4 * q);
    const f32x4* stage4 = reinterpret_cast<const f32x4*>(stage);
    // gate vector: one read when every row of the panel shares its modulation row (PanelRows::uniform), else per row
    const int um = gated ? pr->uniform : -1;
    const bool per_row = gated && um < 0;
    f32x4 gu = f32x4{1.f, 1.f, 1.f, 1.f};
    if (gated && um >= 0) gu = *reinterpret_cast<const f32x4*>(mm.mod + um + gate_chunk * kC + col0 + 4 * q);
#pragma unroll
    for (int b = 0; b < 16 / BR; ++b) {
        f32x4 hv[BR], g[BR];
        int tk[BR], td[BR];
#pragma unroll
        for (int k = 0; k < BR; ++k) {
            const int row = active ? t * 32 + 2 * (BR * b + k) + r2 : 0;
            tk[k] = active ? pr->tok[row] : -1;
            td[k] = prd ? prd->tok[row] : tk[k];
            const long tc = tk[k] < 0 ? 0 : tk[k];
            hv[k] = *reinterpret_cast<const f32x4*>(h + tc * kC + col0 + 4 * q);
            g[k] = gu;
            if (per_row) g[k] = *reinterpret_cast<const f32x4*>(mm.mod + pr->moff[row] + gate_chunk * kC + col0 + 4 * q);
        }
#pragma unroll
        for (int k = 0; k < BR; ++k) {
            const f32x4 v = stage4[(BR * b + k) * 48 + slot];
            f32x4 o = hv[k];
            o[0] += g[k][0] * (v[0] + b4[0]);
            o[1] += g[k][1] * (v[1] + b4[1]);
            o[2] += g[k][2] * (v[2] + b4[2]);
            o[3] += g[k][3] * (v[3] + b4[3]);
            if (tk[k] >= 0) *reinterpret_cast<f32x4*>((hd ? hd : h) + (long)td[k] * kC + col0 + 4 * q) = o;
        }
    }
}

// The same read-modify-write with the residual rows of a batch requested AHEAD of their use (epi_rmw_request before a GEMM or
// before the previous batch's stores, epi_rmw_finish afterwards): the rows do not depend on the GEMM.
template <int BR>
struct EpiPre {
    f32x4 hv[BR];
    int tk[BR];
};
template <int BR>
__device__ __forceinline__ void epi_rmw_request(const int t, const int b, const PanelRows* pr, int col0, const float* __restrict__ h,
                                                EpiPre<BR>& e) {
    const int lane = lane_id();
    const int q = lane % 24, r2 = lane / 24;
    const bool active = lane < 48;
#pragma unroll
    for (int k = 0; k < BR; ++k) {
        const int row = active ? t * 32 + 2 * (BR * b + k) + r2 : 0;
        e.tk[k] = active ? pr->tok[row] : -1;
        const long tc = e.tk[k] < 0 ? 0 : e.tk[k];
        e.hv[k] = *reinterpret_cast<const f32x4*>(h + tc * kC + col0 + 4 * q);
    }
}
template <int BR>
__device__ __forceinline__ void epi_rmw_finish(const int b, const float* stage, int col0, const f32x4 b4, const f32x4 gu,
                                               float* __restrict__ h, const EpiPre<BR>& e) {
    const int lane = lane_id();
    const int q = lane % 24;
    const int slot = lane < 48 ? lane : 0;
    const f32x4* stage4 = reinterpret_cast<const f32x4*>(stage);
#pragma unroll
    for (int k = 0; k < BR; ++k) {
        const f32x4 v = stage4[(BR * b + k) * 48 + slot];
        f32x4 o = e.hv[k];
        o[0] += gu[0] * (v[0] + b4[0]);
        o[1] += gu[1] * (v[1] + b4[1]);
        o[2] += gu[2] * (v[2] + b4[2]);
        o[3] += gu[3] * (v[3] + b4[3]);
        if (e.tk[k] >= 0) *reinterpret_cast<f32x4*>(h + (long)e.tk[k] * kC + col0 + 4 * q) = o;
    }
}
// (uniform-gate panels only: the caller's launch shares one modulation row -- else falls back to the plain form)
template <int FT>
__device__ __forceinline__ void epilogue_gate_residual_lds_pre(const f32x16* acc, const PanelRows* pr, float* stage, int col0,
                                                               const float* __restrict__ bias, const ModMap mm, int gate_chunk,
                                                               bool gated, float* __restrict__ h, EpiPre<8>& e0) {
    static_assert(FT == 3, "slab is [32][96]");
    const int um = gated ? pr->uniform : -1;
    if (um < 0) {   // per-row gates: the plain path (the early request is dropped)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            epi_stage(acc + t * FT, stage);
            epi_rmw<8>(t, pr, stage, col0, bias, mm, gate_chunk, gated, h);
        }
        return;
    }
    const int q = lane_id() % 24;
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + col0 + 4 * q);
    const f32x4 gu = *reinterpret_cast<const f32x4*>(mm.mod + um + gate_chunk * kC + col0 + 4 * q);
    EpiPre<8> e1;
    // batches (t, b): (0,0) requested by the caller; each batch's successor is requested before its own stores
    epi_stage(acc, stage);
    epi_rmw_request<8>(0, 1, pr, col0, h, e1);
    __builtin_amdgcn_sched_barrier(0);
    epi_rmw_finish<8>(0, stage, col0, b4, gu, h, e0);
    __builtin_amdgcn_sched_barrier(0);
    epi_rmw_request<8>(1, 0, pr, col0, h, e0);
    __builtin_amdgcn_sched_barrier(0);
    epi_rmw_finish<8>(1, stage, col0, b4, gu, h, e1);
    __builtin_amdgcn_sched_barrier(0);
    epi_stage(acc + FT, stage);
    epi_rmw_request<8>(1, 1, pr, col0, h, e1);
    __builtin_amdgcn_sched_barrier(0);
    epi_rmw_finish<8>(0, stage, col0, b4, gu, h, e0);
    __builtin_amdgcn_sched_barrier(0);
    epi_rmw_finish<8>(1, stage, col0, b4, gu, h, e1);
}

template <int FT>
__device__ __forceinline__ void epilogue_gate_residual_lds(const f32x16* acc, const PanelRows* pr, float* stage, int col0,
                                                           const float* __restrict__ bias, const ModMap mm,
                                                           int gate_chunk, bool gated, float* __restrict__ h) {
    static_assert(FT == 3, "slab is [32][96]");
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        epi_stage(acc + t * FT, stage);
        epi_rmw<8>(t, pr, stage, col0, bias, mm, gate_chunk, gated, h);
    }
}

}  // namespace mdg
