// k_gemm.hip -- the resident-panel GEMM family: LN+QKV(+RoPE), out-proj(+micro-attention),
// fused MLP, IPA projections, final layer.  gfx950 only (MFMA 32x32x16 bf16, wave64).
#include "kernels.h"
#include "panel.h"

namespace mdg {

constexpr int kRowB = kC * 2;            // panel row bytes at K = 384
// (Round 6, measured and removed: small-launch forms that request ALL rows of their LayerNorm prologue / residual epilogue at once --
// k_mlp8, k_ln_qkv8, a k_ln_qkv_attn4 instantiation for <= one workgroup per CU.  B = 1: attn_L_fused 41.0 against 41.7 us, proj_mlp
// 54.4 against 53.2, 30 627 against 30 495 frames/s; the TPS shard 83 869 against 84 367: nothing -- hipcc already hoists the batched
// requests of an unrolled prologue as far as the registers allow.  profiles/r06_experiments.txt #8.)
constexpr int kPanelBytes = kPanel * kRowB;

// =================================================================================================
// LN + modulate + QKV projection + bias + (q scale folded) + RoPE -> attention operand fragments.
// Replaces latent_model.py:457-461 / 465-475 (modulate(LN(x))) + mha.py:258-263 (q/k/v proj, q scaling)
// + mha.py:356-357 (rotary).  One workgroup = 64 positions of one sequence (FLASH layout) or 64
// consecutive tokens (SMALL layout, residue axis with L <= 8).
// Wave w owns heads 4w..4w+3.  Q and K are computed TRANSPOSED (D[feature][token]) so that a lane
// holds, for its token, 12 features of each head = six rotary pairs (i, i+12): RoPE is lane-local
// and the accumulators ARE the attention MFMA fragments (DESIGN.md "fragment layout").
// =================================================================================================
template <bool ROPE, int TT = 2>
__device__ __forceinline__ void epilogue_heads_T(const f32x16* acc /*[3 ft][TT]*/, const PanelRows* pr, int w,
                                                 const float* __restrict__ bias_perm, const float* __restrict__ rope,
                                                 bool small, int pos0, int len, int seq, int ntile, int tile0,
                                                 unsigned char* __restrict__ frag, __bf16* __restrict__ small_dst,
                                                 int which) {
    const int lane = lane_id(), hh = lane >> 5, tk = lane & 31;
    // All table reads of this epilogue (4 heads x 12 permuted biases, 2 rows x 12 rotary factors) are requested up
    // front as 16-byte loads: left inside the head loop each head paid its own L2 round trip (16 in sequence).
    f32x4 bq[4][3];
    {
        const f32x4* bp = reinterpret_cast<const f32x4*>(bias_perm + (w * 2 + hh) * 48);
#pragma unroll
        for (int hd = 0; hd < 4; ++hd)
#pragma unroll
            for (int c = 0; c < 3; ++c) bq[hd][c] = bp[hd * 3 + c];
    }
    int tokv[TT], posv[TT];
    f32x4 rq[TT][4];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        const int row = tt * 32 + tk;
        tokv[tt] = pr->tok[row];
        int pos = small ? (tokv[tt] >= 0 ? tokv[tt] % len : 0) : pos0 + row;
        if (pos > len) pos = len;  // padding rows: any in-table position (values are never used)
        posv[tt] = pos;
        if (ROPE) {
            const f32x4* rc = reinterpret_cast<const f32x4*>(rope + (long)pos * kRopeRow + 16 * hh);
#pragma unroll
            for (int i = 0; i < 4; ++i) rq[tt][i] = rc[i];
        }
    }
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        const int token = tokv[tt];
        const bool valid = token >= 0;
        float cs[6], sn[6];
        if (ROPE) {
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                cs[p] = rq[tt][p >> 2][p & 3];
                sn[p] = rq[tt][2 + (p >> 2)][p & 3];
            }
        }
        const int tile = tile0 + tt;
#pragma unroll
        for (int hd = 0; hd < 4; ++hd) {
            float e[12];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int ap = 3 * hd + c, ft = ap >> 2, a = ap & 3;
#pragma unroll
                for (int b = 0; b < 4; ++b) e[4 * c + b] = acc[ft * TT + tt][4 * a + b] + bq[hd][c][b];
            }
            if (ROPE) {
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                    const float x1 = e[2 * p], x2 = e[2 * p + 1];
                    e[2 * p] = x1 * cs[p] - x2 * sn[p];
                    e[2 * p + 1] = x2 * cs[p] + x1 * sn[p];
                }
            }
            uint32_t u[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) u[i] = pack_bf16(e[2 * i], e[2 * i + 1]);
            const int head = 4 * w + hd;
            if (small) {
                if (valid) {
                    u32x2* d = reinterpret_cast<u32x2*>(small_dst + (long)token * (3 * kC) + which * kC + head * kDH + hh * 12);
                    d[0] = u32x2{u[0], u[1]};
                    d[1] = u32x2{u[2], u[3]};
                    d[2] = u32x2{u[4], u[5]};
                }
            } else if (tile < ntile) {
                if (which == 1) {   // K: k-step 1 is a full 16-byte slot; slots 4, 5 = 1.0: they pick up Q's -M (bf16 pair)
                    unsigned char* base = frag + ((long)(seq * kH + head) * ntile + tile) * kFragK;
                    *reinterpret_cast<u32x4*>(base + lane * 16) = u32x4{u[0], u[1], u[2], u[3]};
                    *reinterpret_cast<u32x4*>(base + 1024 + lane * 16) = u32x4{u[4], u[5], 0x3f803f80u, 0u};
                } else {
                    unsigned char* base = frag + ((long)(seq * kH + head) * ntile + tile) * kFragQ;
                    *reinterpret_cast<u32x4*>(base + lane * 16) = u32x4{u[0], u[1], u[2], u[3]};
                    *reinterpret_cast<u32x2*>(base + 1024 + lane * 8) = u32x2{u[4], u[5]};
                }
            }
        }
    }
}

// V for the FLASH layout: non-transposed D[token][feature]; the accumulators are the V^T
// fragments of the P.V MFMA (lane = (d, half), register r = key slot) -- no data movement.
template <int TT = 2>
__device__ __forceinline__ void epilogue_v_flash(const f32x16* acc /*[TT][3 ft]*/, int w,
                                                 const float* __restrict__ bias_perm, int seq, int ntile, int tile0,
                                                 unsigned char* __restrict__ vf) {
    const int lane = lane_id(), hh = lane >> 5, n = lane & 31;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int col = 32 * j + n;
        const int hd = col / kDH, d = col - hd * kDH;
        const float b = bias_perm[w * 96 + col];
        const int head = 4 * w + hd;
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int tile = tile0 + tt;
            if (tile < ntile) {
                unsigned char* base = vf + ((long)(seq * kH + head) * ntile + tile) * kFragV + hh * 400 + d * 16;
                const f32x16 a = acc[tt * 3 + j];
                *reinterpret_cast<u32x4*>(base) = u32x4{pack_bf16(a[0] + b, a[1] + b), pack_bf16(a[2] + b, a[3] + b),
                                                         pack_bf16(a[4] + b, a[5] + b), pack_bf16(a[6] + b, a[7] + b)};
                *reinterpret_cast<u32x4*>(base + 800) =
                    u32x4{pack_bf16(a[8] + b, a[9] + b), pack_bf16(a[10] + b, a[11] + b),
                          pack_bf16(a[12] + b, a[13] + b), pack_bf16(a[14] + b, a[15] + b)};
            }
        }
    }
    if (lane < 32) {   // row 24 of every fragment this wave owns: 4 heads x 2 tiles x 2 key halves x 2 k-steps, all ones
        const int hd = lane & 3, tt = (lane >> 2) & 1, h2 = (lane >> 3) & 1, ks = lane >> 4;
        const int tile = tile0 + tt;
        if (tt < TT && tile < ntile)
            *reinterpret_cast<u32x4*>(vf + ((long)(seq * kH + 4 * w + hd) * ntile + tile) * kFragV + ks * 800 + h2 * 400 + kDH * 16) =
                u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    }
}

// The learned bias key / value (mha.py:265-268: one extra key at index `len`, rotated at position `len` like every
// key, :356-357; never masked) as a REAL entry of the K / V^T fragments: key slot len % 32 of tile len / 32 (the
// fragment layout covers len + 1 keys).  Written by the workgroup of a sequence's last panel, wave w for its heads
// 4w..4w+3, AFTER the same wave's K / V epilogues (which may have stored padding-row values into that slot: same wave,
// same address, program order).  The attention kernel then needs no special case for it.
// do_k / do_v: the K / V^T half only (k_ln_qkv8: the two halves are written by different waves).
// tpp: 32-position tiles per panel (2; 1 for the 32-row workgroups of k_ln_qkv8<true, true>)
__device__ __forceinline__ void write_bias_slots(const QkvParams& p, int seq, int w, bool do_k = true, bool do_v = true, int tpp = 2) {
    const int lane = lane_id();
    const int len = p.ax.len, nt = p.ax.ntile();
    const int kt = len >> 5, sl = len & 31;
    if (kt >= tpp * p.panels_per_seq) {
        // len is a multiple of 64: the bias key starts a tile of its own, which no panel epilogue has touched.  Its
        // other 31 key slots are masked, but the PV MFMA still multiplies their V^T entries by P = 0 -- stale bytes
        // that decode to NaN / inf would poison the sum -- and the all-ones row 24 must exist for the bias key's own P
        // to reach the denominator.  So: zero both fragments, then the ones row, then (below) the bias slots.
#pragma unroll
        for (int hd = 0; hd < 4; ++hd) {
            const long ft = (long)(seq * kH + 4 * w + hd) * nt + kt;
            u32x4* kd = reinterpret_cast<u32x4*>(p.kf + ft * kFragK);
            u32x4* vd = reinterpret_cast<u32x4*>(p.vf + ft * kFragV);
            const u32x4 z = {0u, 0u, 0u, 0u}, ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
            if (do_k) {
                kd[lane] = z;
                kd[64 + lane] = u32x4{0u, 0u, 0x3f803f80u, 0u};
            }
            const int r0 = lane, r1 = 64 + lane;   // 16-byte rows of the V^T fragment: [k-step 2][key half 2][25]
            if (do_v) {
                vd[r0] = (r0 % 25) == kDH ? ones : z;
                if (r1 < 100) vd[r1] = (r1 % 25) == kDH ? ones : z;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (do_k && lane < 8) {   // K: (half hh, head hd) -> 12 rotated values = 16 B (k-step 0) + 8 B (k-step 1) of key slot sl
        const int hd = lane & 3, hh = lane >> 2, head = 4 * w + hd;
        const float* bk = p.bias_k + head * kDH;
        const float* rc = p.rope + (long)len * kRopeRow + 16 * hh;
        float e[12];
#pragma unroll
        for (int pp = 0; pp < 6; ++pp) {
            const int i = 6 * hh + pp;
            const float x1 = bk[i], x2 = bk[i + 12], c = rc[pp], sn = rc[8 + pp];
            e[2 * pp] = x1 * c - x2 * sn;
            e[2 * pp + 1] = x2 * c + x1 * sn;
        }
        unsigned char* base = p.kf + ((long)(seq * kH + head) * nt + kt) * kFragK;
        *reinterpret_cast<u32x4*>(base + (hh * 32 + sl) * 16) =
            u32x4{pack_bf16(e[0], e[1]), pack_bf16(e[2], e[3]), pack_bf16(e[4], e[5]), pack_bf16(e[6], e[7])};
        *reinterpret_cast<u32x4*>(base + 1024 + (hh * 32 + sl) * 16) =
            u32x4{pack_bf16(e[8], e[9]), pack_bf16(e[10], e[11]), 0x3f803f80u, 0u};
    }
    if (do_v && lane < kDH) {   // V^T: row d = lane; key slot sl = register r of lane-half hk: r = (sl & 3) + 4 (sl >> 3)
        const int d = lane;
        const int dpsi = 12 * ((d >> 2) & 1) + 4 * (d >> 3) + (d & 3);   // feature of V^T row d (api.hip feat_vflash)
        const int hk = (sl >> 2) & 1, r = (sl & 3) + 4 * (sl >> 3);
#pragma unroll
        for (int hd = 0; hd < 4; ++hd) {
            const int head = 4 * w + hd;
            unsigned char* base = p.vf + ((long)(seq * kH + head) * nt + kt) * kFragV;
            const uint32_t v = pack_bf16(p.bias_v[head * kDH + dpsi], 0.f);
            *reinterpret_cast<uint16_t*>(base + (r >> 3) * 800 + hk * 400 + d * 16 + (r & 7) * 2) = (uint16_t)v;
        }
    }
}

#ifdef MDGEN_DEV_QKV_STAMPS   // (experiment build, scripts/micro/qkv_stamps.py: per-wave s_memtime stamps of k_ln_qkv<false>)
__device__ unsigned long long g_qkv_stamps[16384 * 8];
extern "C" int mdgen_dev_qkv_stamps(void* host, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_qkv_stamps), bytes);
}
#define QKV_STAMP(slot)                                                                                      \
    if (!SMALL && lane_id() == 0 && (long)blockIdx.x * 4 + wave_id() < 16384)                                  \
    g_qkv_stamps[((long)blockIdx.x * 4 + wave_id()) * 8 + (slot)] = __builtin_amdgcn_s_memtime()
// k_ln_qkv_attn4: [wave][16]: 0 start, 1 LN prologue, 2 Q GEMM, 3 Q epilogue, 4 K GEMM, 5 K RoPE, 6 scores + softmax, 7 V GEMM, 8 P V +
// panel, 9 out-projection GEMM, 10 end (scripts/r05/attn4_stamps.py)
__device__ unsigned long long g_attn4_stamps[8192 * 16];
extern "C" int mdgen_dev_attn4_stamps(void* host, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn4_stamps), bytes);
}
// (round 6: the stamps are held in SGPRs and written by ONE branch at the end of the kernel -- a store under `if (lane == 0)` per stamp
// splits the kernel into basic blocks around which hipcc spilled 60-190 registers, and a spilled kernel's phases are not the product's)
#define ATTN4_STAMP_DECL unsigned long long a4st[11] = {}
#define ATTN4_STAMP(slot)                          \
    a4st[slot] = __builtin_amdgcn_s_memtime();     \
    __builtin_amdgcn_sched_barrier(0)
#define ATTN4_STAMP_FLUSH                                                                                   \
    if (lane_id() == 0 && (long)blockIdx.x * 4 + wave_id() < 8192) {                                          \
        _Pragma("unroll") for (int k = 0; k < 11; ++k)                                                        \
            g_attn4_stamps[((long)blockIdx.x * 4 + wave_id()) * 16 + k] = a4st[k];                            \
    }
#else
#define QKV_STAMP(slot)
#define ATTN4_STAMP_DECL
#define ATTN4_STAMP(slot)
#define ATTN4_STAMP_FLUSH
#endif

// PRE (FLASH layout only): the panel first runs the PREVIOUS sub-layer's out-projection + gated residual for its 64 tokens
// (attention output rows p.obuf, weights p.wo / p.bo, gate chunk p.gate_chunk: what k_proj<0> does in a launch of its own,
// mha.py:397, latent_model.py:462), then normalises the rows it has just updated -- re-read from L2 instead of HBM.  Used for
// residue-axis out-projection -> temporal q, k, v (the verdict's item 3 (i)); the out-projection is
// token-local, so the panel may be cut along the NEXT sub-layer's axis.
template <bool SMALL, bool PRE = false>
__global__ __launch_bounds__(256, 2) void k_ln_qkv(const QkvParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[sizeof(PanelRows) + kPanelBytes];
    PanelRows* pr = reinterpret_cast<PanelRows*>(smem);
    unsigned char* panel = smem + sizeof(PanelRows);
    int seq = 0, pos0 = 0, tile0 = 0;
    if (SMALL) {
        setup_rows_linear(pr, (long)blockIdx.x * kPanel, p.nrows, p.mm);
    } else {
        seq = blockIdx.x / p.panels_per_seq;
        const int pn = blockIdx.x - seq * p.panels_per_seq;
        pos0 = pn * kPanel;
        tile0 = pn * 2;
        setup_rows_axis(pr, p.ax, seq, pos0, p.mm);
        // key-validity words of this panel's two tiles for the attention kernel (kernels.h flash_vmask_*): the 64
        // lanes of wave 0 are the panel's 64 positions.  The sequence's last panel also writes the words behind it:
        // zeros, except the bias key's when it starts a tile of its own (len a multiple of 64).
        if (wave_id() == 0) {
            const int lane = lane_id(), len = p.ax.len, pos = pos0 + lane;
            const float mv = p.mk.at(p.ax.token(seq, pos < len ? pos : len - 1));
            const unsigned long long bal = __ballot(pos == len || (pos < len && mv != 0.f));
            uint32_t* vm = p.vmask + (long)seq * p.vmask_stride;
            if (lane < 2) vm[tile0 + lane] = (uint32_t)(bal >> (32 * lane));
            if (pn == p.panels_per_seq - 1) {
                const int idx = tile0 + 2 + lane;
                if (idx < p.vmask_stride) vm[idx] = idx == (len >> 5) ? 1u << (len & 31) : 0u;
                if (idx + 64 < p.vmask_stride) vm[idx + 64] = 0u;   // (stride - 2 panels_per_seq can reach 65)
            }
        }
    }
    QKV_STAMP(0);
    __syncthreads();
    const int w = __builtin_amdgcn_readfirstlane(wave_id()), lane = lane_id();
    f32x16 acc[6];
    if (PRE) {
        prologue_bf16<kC>(panel, pr, p.obuf);
        __syncthreads();
        zero_acc<6>(acc);
        wave_gemm<2, 3, 24, false>(panel, kRowB, 0, 0, p.wo + (size_t)(3 * w) * 24 * 64 + lane, 24 * 64, acc);
        __syncthreads();   // every wave is done reading the panel: reuse it as four 12 KiB staging slabs
        epilogue_gate_residual_lds<3>(acc, pr, reinterpret_cast<float*>(panel) + w * (32 * 96), 96 * w, p.bo, p.mm, p.gate_chunk,
                                      true, p.h_rw);
        __syncthreads();   // (vmcnt(0) + barrier) the updated rows are in L2; the slabs are free
    }
    prologue_ln<false>(panel, pr, PRE ? p.h_rw : p.h, p.mm, p.shift_chunk, p.scale_chunk, 1e-6f);
    __syncthreads();
    QKV_STAMP(1);
    const int ntile = p.ax.ntile();
    const int len = p.ax.len;
    // ---- Q (heads 4w..4w+3), transposed
    zero_acc<6>(acc);
    wave_gemm<2, 3, 24, true>(panel, kRowB, 0, 0, p.wq + (size_t)(3 * w) * 24 * 64 + lane, 24 * 64, acc);
    QKV_STAMP(2);
    epilogue_heads_T<true>(acc, pr, w, p.bq, p.rope, SMALL, pos0, len, seq, ntile, tile0, p.qf, p.qkv_small, 0);
    __builtin_amdgcn_sched_barrier(0);
    QKV_STAMP(3);
    // ---- K
    zero_acc<6>(acc);
    wave_gemm<2, 3, 24, true>(panel, kRowB, 0, 0, p.wk + (size_t)(3 * w) * 24 * 64 + lane, 24 * 64, acc);
    QKV_STAMP(4);
    epilogue_heads_T<true>(acc, pr, w, p.bk, p.rope, SMALL, pos0, len, seq, ntile, tile0, p.kf, p.qkv_small, 1);
    __builtin_amdgcn_sched_barrier(0);
    QKV_STAMP(5);
    // ---- V
    zero_acc<6>(acc);
    if (SMALL) {
        wave_gemm<2, 3, 24, true>(panel, kRowB, 0, 0, p.wv + (size_t)(3 * w) * 24 * 64 + lane, 24 * 64, acc);
        epilogue_heads_T<false>(acc, pr, w, p.bv, nullptr, true, pos0, len, seq, ntile, tile0, nullptr, p.qkv_small, 2);
    } else {
        wave_gemm<2, 3, 24, false>(panel, kRowB, 0, 0, p.wv + (size_t)(3 * w) * 24 * 64 + lane, 24 * 64, acc);
        QKV_STAMP(6);
        epilogue_v_flash(acc, w, p.bv, seq, ntile, tile0, p.vf);
        if ((int)blockIdx.x - seq * p.panels_per_seq == p.panels_per_seq - 1) {   // the sequence's last panel
            __builtin_amdgcn_sched_barrier(0);   // after this wave's own K / V stores
            write_bias_slots(p, seq, w);
        }
        QKV_STAMP(7);
    }
}

// -------------------------------------------------------------------------------------------------
// Residue axis with L == 4 (tetrapeptides): LN -> QKV -> RoPE -> 5-key attention, all in one kernel.
//
// A panel is 64 consecutive tokens = 16 whole sequences of 4 residues, and in the transposed GEMM layout a lane
// IS a token: the four keys / values a query needs sit in the other three lanes of its DPP quad.  So q, k, v
// never leave the registers: scores are 12 quad-broadcast FMAs per (key, head) plus one half-wave swap (a lane
// holds 12 of a head's 24 features), softmax over 4 keys + the learned bias key is lane-local, and the output
// is written as the bf16 A-operand rows of the out-projection -- 49 MB per launch instead of the 147 MB SMALL
// layout, which the out-projection kernel then no longer has to read back (k_proj<0> instead of k_proj<2>).
// Same math as prologue_micro_attn (mha.py:258-268, 356-396); q and the bias key/value are rounded to bf16 as
// there, k and v stay fp32.
// -------------------------------------------------------------------------------------------------
template <int J>
__device__ __forceinline__ float quad_bcast(float v) {   // value of lane (quad base + J) in every lane of the quad
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), J * 0x55, 0xf, 0xf, true));
}
__device__ __forceinline__ float half_sum(float x) {   // x(lane) + x(lane ^ 32)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// the 12 values a lane holds for head hd of its token (tile tt of the wave's TT) in the transposed QKV accumulators, + bias
template <int TT = 2>
__device__ __forceinline__ void head_values(const f32x16* acc, int tt, int hd, const f32x4 (&bq)[3], float (&e)[12]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int ap = 3 * hd + c, ft = ap >> 2, a = ap & 3;
#pragma unroll
        for (int b = 0; b < 4; ++b) e[4 * c + b] = acc[ft * TT + tt][4 * a + b] + bq[c][b];
    }
}
__device__ __forceinline__ void load_head_bias(const float* bias_perm, int w, int hh, f32x4 (&bq)[4][3]) {
    const f32x4* bp = reinterpret_cast<const f32x4*>(bias_perm + (w * 2 + hh) * 48);
#pragma unroll
    for (int hd = 0; hd < 4; ++hd)
#pragma unroll
        for (int c = 0; c < 3; ++c) bq[hd][c] = bp[hd * 3 + c];
}

// (Weight-ring depth of the small launches' one-tile-per-wave GEMMs: 8 or 12 k-steps in flight instead of 4 change nothing -- they are
// bound by the 64 B/clk at which a CU's vector memory path returns data: 3 KiB per wave and k-step for 96 cycles of matrix-pipe time;
// profiles/r06_experiments.txt #17.)

// ---- LN -> q, k, v (FLASH layout) for launches of at most one workgroup per CU: eight waves per panel ------------------------------
// (see k_mlp8.)  The split keeps 64 rows per wave -- with one row tile per wave every weight fragment feeds one MFMA instead of two and
// the launch gets slower (profiles/r04_experiments.txt #17) -- and divides the three products instead: waves 0..3 compute q and k,
// waves 4..7 compute v (and its all-ones row); the LayerNorm prologue is split by row batches.
//
// SPLIT (launches of at most half a workgroup per CU: B = 1): a panel's three products go to TWO workgroups -- blockIdx 2 n: waves
// 0..3 q, waves 4..7 k (and the key-validity words); blockIdx 2 n + 1: waves 0..3 v, waves 4..7 leave after their share of the
// LayerNorm prologue (which both workgroups run in full) -- so the longest chain of a launch is one product instead of two.  The
// two workgroups write disjoint fragments; nothing is exchanged.
// HALF (with SPLIT; launches of at most one such workgroup per CU: B = 1; round 6, profiles/r06_experiments.txt #14 / #16): 32 positions per
// workgroup pair -- the panel's upper half stays empty, every wave computes one 32-token tile -- twice the workgroups, half the rows per SIMD.
template <bool SPLIT, bool HALF = false>
__global__ __launch_bounds__(512, 1) void k_ln_qkv8(const QkvParams p) {
    static_assert(!HALF || SPLIT, "32-row workgroups exist in the split form only");
    constexpr int TT = HALF ? 1 : 2, kRows = 32 * TT;
    __shared__ __attribute__((aligned(16))) unsigned char smem[sizeof(PanelRows) + kPanelBytes];
    PanelRows* pr = reinterpret_cast<PanelRows*>(smem);
    unsigned char* panel = smem + sizeof(PanelRows);
    const int blk = SPLIT ? blockIdx.x >> 1 : blockIdx.x, half = SPLIT ? blockIdx.x & 1 : 0;
    const int seq = blk / p.panels_per_seq;
    const int pn = blk - seq * p.panels_per_seq;
    const int pos0 = pn * kRows, tile0 = pn * TT;
    setup_rows_axis(pr, p.ax, seq, pos0, p.mm, kRows);
    if (wave_id() == 0 && half == 0) {   // key-validity words of this panel's tiles (as k_ln_qkv<false>)
        const int lane = lane_id(), len = p.ax.len, pos = pos0 + lane;
        const float mv = p.mk.at(p.ax.token(seq, pos < len ? pos : len - 1));
        const unsigned long long bal = __ballot(pos == len || (pos < len && mv != 0.f));
        uint32_t* vm = p.vmask + (long)seq * p.vmask_stride;
        if (lane < TT) vm[tile0 + lane] = (uint32_t)(bal >> (32 * lane));
        if (pn == p.panels_per_seq - 1) {
            const int idx = tile0 + TT + lane;
            if (idx < p.vmask_stride) vm[idx] = idx == (len >> 5) ? 1u << (len & 31) : 0u;
            if (idx + 64 < p.vmask_stride) vm[idx + 64] = 0u;
        }
    }
    __syncthreads();
    const int w8 = __builtin_amdgcn_readfirstlane(wave_id()), g = w8 >> 2, w = w8 & 3, lane = lane_id();
    if (HALF) {   // 32 rows: one batch of sixteen per wave group
        if (g == 0) prologue_ln<false, 0, 1>(panel, pr, p.h, p.mm, p.shift_chunk, p.scale_chunk, 1e-6f, w, lane);
        else prologue_ln<false, 1, 2>(panel, pr, p.h, p.mm, p.shift_chunk, p.scale_chunk, 1e-6f, w, lane);
    } else {
        if (g == 0) prologue_ln<false, 0, 2>(panel, pr, p.h, p.mm, p.shift_chunk, p.scale_chunk, 1e-6f, w, lane);
        else prologue_ln<false, 2, 4>(panel, pr, p.h, p.mm, p.shift_chunk, p.scale_chunk, 1e-6f, w, lane);
    }
    __syncthreads();
    const int ntile = p.ax.ntile(), len = p.ax.len;
    const bool last = pn == p.panels_per_seq - 1;
    f32x16 acc[3 * TT];
    // which products this wave group computes: q and k (do_q, do_k) or v
    const bool do_q = SPLIT ? half == 0 && g == 0 : g == 0, do_k = SPLIT ? half == 0 && g == 1 : g == 0;
    const bool do_v = SPLIT ? half == 1 && g == 0 : g == 1;
    if (do_q) {
        zero_acc<3 * TT>(acc);
        wave_gemm<TT, 3, 24, true>(panel, kRowB, 0, 0, p.wq + (size_t)(3 * w) * 24 * 64 + lane, 24 * 64, acc);
        epilogue_heads_T<true, TT>(acc, pr, w, p.bq, p.rope, false, pos0, len, seq, ntile, tile0, p.qf, p.qkv_small, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (do_k) {
        zero_acc<3 * TT>(acc);
        wave_gemm<TT, 3, 24, true>(panel, kRowB, 0, 0, p.wk + (size_t)(3 * w) * 24 * 64 + lane, 24 * 64, acc);
        epilogue_heads_T<true, TT>(acc, pr, w, p.bk, p.rope, false, pos0, len, seq, ntile, tile0, p.kf, p.qkv_small, 1);
        if (last) {   // the learned bias key, after this wave's own K stores
            __builtin_amdgcn_sched_barrier(0);
            write_bias_slots(p, seq, w, true, false, TT);
        }
    }
    if (do_v) {
        zero_acc<3 * TT>(acc);
        wave_gemm<TT, 3, 24, false>(panel, kRowB, 0, 0, p.wv + (size_t)(3 * w) * 24 * 64 + lane, 24 * 64, acc);
        epilogue_v_flash<TT>(acc, w, p.bv, seq, ntile, tile0, p.vf);
        if (last) {   // the learned bias value, after this wave's own V^T stores
            __builtin_amdgcn_sched_barrier(0);
            write_bias_slots(p, seq, w, false, true, TT);
        }
    }
}

// PROJ: also run the sub-layer's out-projection and gated residual update here (mha.py:397, latent_model.py:462):
// the attention output goes straight into the LDS panel as the A operand instead of through HBM.
// Weight-ring depth (k-steps in flight) of the K and V GEMMs, which run with q / P live.  Deeper rings compile without further
// spills (3 / 4) or with 8 more spilled dwords (4 / 4) and change nothing: 129.4-132.8 us per launch at cfg-2 for all four
// combinations on one box (profiles/r05_experiments.txt #9).
constexpr int kAttn4KPF = 2, kAttn4VPF = 3;
// HALF (with PROJ; launches of at most HALF a workgroup per CU: B = 1; round 6): a workgroup takes 32 rows instead of 64 -- the panel's
// upper half stays empty, every wave computes ONE 32-token tile (TT = 1) -- so the launch has twice the workgroups on twice the CUs and
// every SIMD half the work: a small launch lasts as long as one wave's chain (profiles/r06_experiments.txt #14: 87k cycles per wave at
// B = 1 and at cfg-3's shard alike), and the chain is LN prologue + four products + the 5-key attention + the residual epilogue of the
// rows the SIMD owns.  (Eight waves on the same 64 rows -- r06 #13 -- only put two waves on every SIMD: the same work per SIMD, a tie.)
// A weight fragment then feeds one MFMA instead of two, which only a launch that leaves CUs idle can afford.  q and P stay in registers.
template <bool PROJ, bool HALF = false>
__global__ __launch_bounds__(256, 2) void k_ln_qkv_attn4(const QkvParams p) {
    static_assert(!HALF || PROJ, "the half-panel form is the whole sub-layer");
    constexpr int TT = HALF ? 1 : 2, kRows = HALF ? 32 : kPanel;
    __shared__ __attribute__((aligned(16))) unsigned char smem[sizeof(PanelRows) + kPanelBytes];
    // q of the second 32-token tile waits in LDS while the K GEMM runs ([wave][value][lane]: conflict-free): with
    // all 48 packed q registers live across that GEMM hipcc spilled ~90 registers per lane -- 180 MB of scratch
    // traffic per launch, more than the q/k/v stores this kernel exists to avoid.
    __shared__ uint32_t qstash[HALF ? 1 : 4][24][64];
    PanelRows* pr = reinterpret_cast<PanelRows*>(smem);
    unsigned char* panel = smem + sizeof(PanelRows);
    ATTN4_STAMP_DECL;
    ATTN4_STAMP(0);
    {
        const long row0 = (long)blockIdx.x * kRows, rend = row0 + kRows;
        setup_rows_linear(pr, row0, rend < p.nrows ? rend : p.nrows, p.mm);   // (HALF: rows 32 .. 63 of the panel are padding rows)
    }
    __syncthreads();
    const int w = __builtin_amdgcn_readfirstlane(wave_id());
    constexpr int t0 = 0;
    if (HALF) prologue_ln<false, 0, 2>(panel, pr, p.h, p.mm, p.shift_chunk, p.scale_chunk, 1e-6f, w, lane_id());
    else prologue_ln<false>(panel, pr, p.h, p.mm, p.shift_chunk, p.scale_chunk, 1e-6f);
    __syncthreads();
    ATTN4_STAMP(1);
    const int lane = lane_id(), hh = lane >> 5, tk = lane & 31;
    constexpr int L = 4;
    // per-token constants: token id, key validity of the own token; the rotary factors (position = token % 4)
    // are re-read from the (L2-resident) table after each GEMM rather than held across it
    int tok[TT];
    float mval[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        tok[tt] = pr->tok[(t0 + tt) * 32 + tk];
        mval[tt] = p.mk.at(tok[tt] < 0 ? 0 : tok[tt]);
    }
    auto load_rope = [&](f32x4 (&rq)[TT][4]) {
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int tc = tok[tt] < 0 ? 0 : tok[tt];
            const f32x4* rc = reinterpret_cast<const f32x4*>(p.rope + (long)(tc & (L - 1)) * kRopeRow + 16 * hh);
#pragma unroll
            for (int i = 0; i < 4; ++i) rq[tt][i] = rc[i];
        }
    };
    f32x16 acc[3 * TT];
    f32x4 bb[4][3];
    f32x4 rq[TT][4];
    // ---- Q (heads 4w..4w+3): RoPE, keep as bf16 pairs (48 registers)
    zero_acc<3 * TT>(acc);
    wave_gemm<TT, 3, 24, true>(panel, kRowB, t0, 0, p.wq + (size_t)(3 * w) * 24 * 64 + lane, 24 * 64, acc);
    ATTN4_STAMP(2);
    load_head_bias(p.bq, w, hh, bb);
    load_rope(rq);
    uint32_t qp[TT][4][6];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int hd = 0; hd < 4; ++hd) {
            float e[12];
            head_values<TT>(acc, tt, hd, bb[hd], e);
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const float c = rq[tt][q >> 2][q & 3], sn = rq[tt][2 + (q >> 2)][q & 3];
                const float x1 = e[2 * q], x2 = e[2 * q + 1];
                qp[tt][hd][q] = pack_bf16(x1 * c - x2 * sn, x2 * c + x1 * sn);
            }
        }
    if constexpr (!HALF) {
#pragma unroll
        for (int hd = 0; hd < 4; ++hd)
#pragma unroll
            for (int q = 0; q < 6; ++q) qstash[w][hd * 6 + q][lane] = qp[TT - 1][hd][q];
    }
    ATTN4_STAMP(3);
    __builtin_amdgcn_sched_barrier(0);
    // ---- K: RoPE in place, then the scores of the 4 keys of the quad + the bias key; softmax -> P (40 registers)
    zero_acc<3 * TT>(acc);
    wave_gemm<TT, 3, 24, true, HALF ? 4 : kAttn4KPF>(panel, kRowB, t0, 0, p.wk + (size_t)(3 * w) * 24 * 64 + lane, 24 * 64, acc);   // (q is live)
    ATTN4_STAMP(4);
    load_head_bias(p.bk, w, hh, bb);
    load_rope(rq);
    // pass 1: bias + RoPE IN PLACE in the accumulators (frees the bias / rotary registers before the scores)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int hd = 0; hd < 4; ++hd) {
            float k[12];
            head_values<TT>(acc, tt, hd, bb[hd], k);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int ap = 3 * hd + c, ft = ap >> 2, a = ap & 3;
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) {   // values 4c + 2 b2, 4c + 2 b2 + 1 = rotary pair q = 2c + b2
                    const int q = 2 * c + b2;
                    const float cs = rq[tt][q >> 2][q & 3], sn = rq[tt][2 + (q >> 2)][q & 3];
                    const float x1 = k[2 * q], x2 = k[2 * q + 1];
                    acc[ft * TT + tt][4 * a + 2 * b2] = x1 * cs - x2 * sn;
                    acc[ft * TT + tt][4 * a + 2 * b2 + 1] = x2 * cs + x1 * sn;
                }
            }
        }
    ATTN4_STAMP(5);
    __builtin_amdgcn_sched_barrier(0);
    // pass 2: scores against the 4 keys of the quad + the bias key, softmax -> P (40 registers)
    float P[TT][4][5];
    {
        // learned bias key (mha.py:265-268), rotated at position L like every key (:356-357), rounded to bf16
        const float* rcL = p.rope + (long)L * kRopeRow + 16 * hh;
#pragma unroll
        for (int hd = 0; hd < 4; ++hd) {
            const float* bk = p.bias_k + (4 * w + hd) * kDH;
            float kb[12];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int i = 6 * hh + q;
                const float x1 = bk[i], x2 = bk[i + 12], c = rcL[q], sn = rcL[8 + q];
                kb[2 * q] = bf16_lo(pack_bf16(x1 * c - x2 * sn, 0.f));
                kb[2 * q + 1] = bf16_lo(pack_bf16(x2 * c + x1 * sn, 0.f));
            }
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                float k[12], qf[12];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int ap = 3 * hd + c, ft = ap >> 2, a = ap & 3;
#pragma unroll
                    for (int b2 = 0; b2 < 4; ++b2) k[4 * c + b2] = acc[ft * TT + tt][4 * a + b2];
                }
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const uint32_t u = (HALF || tt == 0) ? qp[tt][hd][q] : qstash[w][hd * 6 + q][lane];
                    qf[2 * q] = bf16_lo(u);
                    qf[2 * q + 1] = bf16_hi(u);
                }
                float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    s[0] += qf[i] * quad_bcast<0>(k[i]);
                    s[1] += qf[i] * quad_bcast<1>(k[i]);
                    s[2] += qf[i] * quad_bcast<2>(k[i]);
                    s[3] += qf[i] * quad_bcast<3>(k[i]);
                    s[4] += qf[i] * kb[i];
                }
#pragma unroll
                for (int j = 0; j < 5; ++j) s[j] = half_sum(s[j]);
                const float m0 = quad_bcast<0>(mval[tt]), m1 = quad_bcast<1>(mval[tt]), m2 = quad_bcast<2>(mval[tt]),
                            m3 = quad_bcast<3>(mval[tt]);
                s[0] = m0 != 0.f ? s[0] : -1e30f;
                s[1] = m1 != 0.f ? s[1] : -1e30f;
                s[2] = m2 != 0.f ? s[2] : -1e30f;
                s[3] = m3 != 0.f ? s[3] : -1e30f;
                const float mx = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), s[4]);   // the bias key is never masked
                float den = 0.f;
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    s[j] = s[j] > -1e29f ? __builtin_amdgcn_exp2f(s[j] - mx) : 0.f;
                    den += s[j];
                }
                const float inv = 1.0f / den;
#pragma unroll
                for (int j = 0; j < 5; ++j) P[tt][hd][j] = s[j] * inv;
            }
            __builtin_amdgcn_sched_barrier(0);   // one head at a time: keeps the scheduler from overlapping all four
        }
    }
    ATTN4_STAMP(6);
    // the 40 attention weights of this lane wait in the (now free) q stash as 20 bf16 pairs while the V GEMM runs
    // (bf16 weights: what the streaming kernel feeds its PV MFMA as well)
    // (the half-panel form: 20 weights as 10 packed registers, rounded to bf16 alike, held across the GEMM)
    uint32_t Pk[10];
    if constexpr (HALF) {
        const float* Pf = &P[0][0][0];
#pragma unroll
        for (int i = 0; i < 10; ++i) Pk[i] = pack_bf16(Pf[2 * i], Pf[2 * i + 1]);
    } else {
        const float* Pf = &P[0][0][0];
#pragma unroll
        for (int i = 0; i < 20; ++i) qstash[w][i][lane] = pack_bf16(Pf[2 * i], Pf[2 * i + 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- V (transposed as well: a lane holds features 12 hh .. 12 hh + 11 of each head of its token)
    zero_acc<3 * TT>(acc);
    wave_gemm<TT, 3, 24, true, HALF ? 4 : kAttn4VPF>(panel, kRowB, t0, 0, p.wv + (size_t)(3 * w) * 24 * 64 + lane, 24 * 64, acc);
    ATTN4_STAMP(7);
    load_head_bias(p.bv, w, hh, bb);
    {
        float* Pf = &P[0][0][0];
#pragma unroll
        for (int i = 0; i < 10 * TT; ++i) {
            const uint32_t u = HALF ? Pk[i % 10] : qstash[w][i][lane];
            Pf[2 * i] = bf16_lo(u);
            Pf[2 * i + 1] = bf16_hi(u);
        }
    }
    // PROJ: once every wave has finished its V GEMM the LN panel is dead, and the attention output is written
    // straight into it (bf16 [64][384], swizzled: the A operand of the out-projection, as k_proj<0> builds it)
    if (PROJ) __syncthreads();
#pragma unroll
    for (int hd = 0; hd < 4; ++hd) {
        const int head = 4 * w + hd;
        float bvv[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) bvv[i] = bf16_lo(pack_bf16(p.bias_v[head * kDH + 12 * hh + i], 0.f));
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            float v[12], o[12];
            head_values<TT>(acc, tt, hd, bb[hd], v);
#pragma unroll
            for (int i = 0; i < 12; ++i)
                o[i] = P[tt][hd][0] * quad_bcast<0>(v[i]) + P[tt][hd][1] * quad_bcast<1>(v[i]) +
                       P[tt][hd][2] * quad_bcast<2>(v[i]) + P[tt][hd][3] * quad_bcast<3>(v[i]) + P[tt][hd][4] * bvv[i];
            if (PROJ) {
                const bool ok = tok[tt] >= 0;   // padding rows enter the GEMM as zeros
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    *reinterpret_cast<u32x2*>(panel + panel_off((t0 + tt) * 32 + tk, head * 48 + hh * 24 + 8 * i, kRowB)) =
                        u32x2{ok ? pack_bf16(o[4 * i], o[4 * i + 1]) : 0u, ok ? pack_bf16(o[4 * i + 2], o[4 * i + 3]) : 0u};
            } else if (tok[tt] >= 0) {
                u32x2* d = reinterpret_cast<u32x2*>(p.obuf + (long)tok[tt] * kC + head * kDH + hh * 12);
                d[0] = u32x2{pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3])};
                d[1] = u32x2{pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7])};
                d[2] = u32x2{pack_bf16(o[8], o[9]), pack_bf16(o[10], o[11])};
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!PROJ) return;
    // ---- out-projection + gated residual, as k_proj<0>
    __syncthreads();   // the whole attention output is in the panel
    ATTN4_STAMP(8);
    // (the epilogue's first batch of residual rows requested ahead of this GEMM, as k_flash_proj does: measured, no gain here -- 130.6
    // against 128-130 us per launch, 125.9k against 126.4k frames/s; profiles/r06_experiments.txt #6)
    zero_acc<3 * TT>(acc);
    wave_gemm<TT, 3, 24, false>(panel, kRowB, t0, 0, p.wo + (size_t)(3 * w) * 24 * 64 + lane, 24 * 64, acc);
    ATTN4_STAMP(9);
    __syncthreads();   // every wave is done reading the panel: reuse it as four 12 KiB staging slabs
    if constexpr (HALF) {
        float* stage = reinterpret_cast<float*>(panel) + w * (32 * 96);
        epi_stage(acc, stage);
        epi_rmw<8>(0, pr, stage, 96 * w, p.bo, p.mm, p.gate_chunk, true, p.h_rw);
    } else {
        epilogue_gate_residual_lds<3>(acc, pr, reinterpret_cast<float*>(panel) + w * (32 * 96), 96 * w, p.bo, p.mm, p.gate_chunk,
                                      true, p.h_rw);
    }
    ATTN4_STAMP(10);
    ATTN4_STAMP_FLUSH
}

// =================================================================================================
// Attention output projection + gate + residual:  h += gate * (Wo o + bo)
// (mha.py:397 out_proj; latent_model.py:462,476 gated residual).  A-operand sources:
//   MODE 0: bf16 rows [token][384] written by the flash kernel
//   MODE 1: bf16 rows [token][256] (IPA concat features; gate = 1; ipa.py:250-254 + latent_model.py:373)
//   MODE 2: residue-axis micro-attention computed in the prologue from the SMALL q/k/v layout
//           (L <= 8: five keys incl. the learned bias key; mha.py:265-268, 359-396)
// =================================================================================================
// Every global load is unconditional (padding rows read token 0, key slots j >= L re-read key L-1 and are
// masked afterwards) and the loads of one (row, head) item are issued together: with loads under
// `if (token >= 0)` / `if (mask)` each key paid its own memory round trip (~13 in sequence per item).
template <int MAXL>
__device__ __forceinline__ void prologue_micro_attn(unsigned char* panel, const PanelRows* pr, const ProjParams& p) {
    const int L = p.ax.len;
    for (int item = threadIdx.x; item < kPanel * kH; item += 256) {
        const int row = item >> 4, head = item & 15;
        const int token = pr->tok[row];
        const int tkc = token < 0 ? 0 : token;
        const int seq = tkc / L;
        const __bf16* sbase = p.qkv_small + (long)seq * L * (3 * kC) + head * kDH;   // key/value j: + j*3C + {C, 2C}
        const u32x4* qp = reinterpret_cast<const u32x4*>(p.qkv_small + (long)tkc * (3 * kC) + head * kDH);
        u32x4 qv[3], kk[MAXL][3], vv[MAXL][3];
        float mk[MAXL];
#pragma unroll
        for (int i = 0; i < 3; ++i) qv[i] = qp[i];
#pragma unroll
        for (int j = 0; j < MAXL; ++j) {
            const int jc = j < L ? j : L - 1;
            mk[j] = p.mk.at((long)seq * L + jc);
            const u32x4* kp = reinterpret_cast<const u32x4*>(sbase + (long)jc * (3 * kC) + kC);
#pragma unroll
            for (int i = 0; i < 3; ++i) kk[j][i] = kp[i];
        }
        if (MAXL <= 4) {   // few keys: the values fit in registers too -> one round trip for everything
#pragma unroll
            for (int j = 0; j < MAXL; ++j) {
                const int jc = j < L ? j : L - 1;
                const u32x4* vp = reinterpret_cast<const u32x4*>(sbase + (long)jc * (3 * kC) + 2 * kC);
#pragma unroll
                for (int i = 0; i < 3; ++i) vv[j][i] = vp[i];
            }
        }
        float q[24];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                q[8 * i + 2 * j] = bf16_lo(qv[i][j]);
                q[8 * i + 2 * j + 1] = bf16_hi(qv[i][j]);
            }
        float s[MAXL + 1];
        float mx = -1e30f;
#pragma unroll
        for (int j = 0; j < MAXL; ++j) {
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    d += q[8 * i + 2 * jj] * bf16_lo(kk[j][i][jj]) + q[8 * i + 2 * jj + 1] * bf16_hi(kk[j][i][jj]);
            s[j] = (j < L && mk[j] != 0.f) ? d : -1e30f;
            mx = fmaxf(mx, s[j]);
        }
        if (MAXL > 4) {
#pragma unroll
            for (int j = 0; j < MAXL; ++j) {
                const int jc = j < L ? j : L - 1;
                const u32x4* vp = reinterpret_cast<const u32x4*>(sbase + (long)jc * (3 * kC) + 2 * kC);
#pragma unroll
                for (int i = 0; i < 3; ++i) vv[j][i] = vp[i];
            }
        }
        {   // learned bias key at position L, rotated there (mha.py:265-268 before :356-357); never masked
            const float* bk = p.bias_k + head * kDH;
            const float* rc = p.rope + (long)L * kRopeRow;
            float d = 0.f;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int pp = 0; pp < 6; ++pp) {
                    const int i = 6 * hh + pp;
                    const float x1 = bk[i], x2 = bk[i + 12], c = rc[16 * hh + pp], sn = rc[16 * hh + 8 + pp];
                    const float k1 = bf16_lo(pack_bf16(x1 * c - x2 * sn, 0.f));
                    const float k2 = bf16_lo(pack_bf16(x2 * c + x1 * sn, 0.f));
                    d += q[hh * 12 + 2 * pp] * k1 + q[hh * 12 + 2 * pp + 1] * k2;
                }
            s[MAXL] = d;
            mx = fmaxf(mx, d);
        }
        float o[24];
#pragma unroll
        for (int i = 0; i < 24; ++i) o[i] = 0.f;
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < MAXL; ++j) {
            const float pj = s[j] > -1e29f ? __builtin_amdgcn_exp2f(s[j] - mx) : 0.f;
            den += pj;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    o[8 * i + 2 * jj] += pj * bf16_lo(vv[j][i][jj]);
                    o[8 * i + 2 * jj + 1] += pj * bf16_hi(vv[j][i][jj]);
                }
        }
        {
            const float pj = __builtin_amdgcn_exp2f(s[MAXL] - mx);
            den += pj;
            const float* bv = p.bias_v + head * kDH;
#pragma unroll
            for (int i = 0; i < 24; ++i) o[i] += pj * bf16_lo(pack_bf16(bv[i], 0.f));
        }
        const float inv = token >= 0 ? 1.0f / den : 0.f;   // padding rows are written as zeros
#pragma unroll
        for (int i = 0; i < 3; ++i)
            *reinterpret_cast<u32x4*>(panel + panel_off(row, head * 48 + i * 16, kRowB)) =
                u32x4{pack_bf16(o[8 * i] * inv, o[8 * i + 1] * inv), pack_bf16(o[8 * i + 2] * inv, o[8 * i + 3] * inv),
                      pack_bf16(o[8 * i + 4] * inv, o[8 * i + 5] * inv), pack_bf16(o[8 * i + 6] * inv, o[8 * i + 7] * inv)};
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_proj(const ProjParams p) {
    constexpr int K = (MODE == 1) ? kIpaFeat : kC;
    constexpr int ROWB = K * 2;
    constexpr int KS = K / 16;
    __shared__ __attribute__((aligned(16))) unsigned char smem[sizeof(PanelRows) + kPanel * ROWB];
    PanelRows* pr = reinterpret_cast<PanelRows*>(smem);
    unsigned char* panel = smem + sizeof(PanelRows);
    setup_rows_linear(pr, (long)blockIdx.x * kPanel, p.nrows, p.mm);
    __syncthreads();
    if (MODE == 2) {
        if (p.ax.len <= 4) prologue_micro_attn<4>(panel, pr, p);
        else prologue_micro_attn<8>(panel, pr, p);
    } else {
        prologue_bf16<K>(panel, pr, p.a_bf16);
    }
    __syncthreads();
    const int w = __builtin_amdgcn_readfirstlane(wave_id()), lane = lane_id();
    f32x16 acc[6];
    zero_acc<6>(acc);
    wave_gemm<2, 3, KS, false>(panel, ROWB, 0, 0, p.w + (size_t)(3 * w) * KS * 64 + lane, KS * 64, acc);
    if (MODE == 1) {
        epilogue_gate_residual<2, 3>(acc, pr, 96 * w, p.bias, p.mm, p.gate_chunk, p.gated != 0, p.h);
    } else {
        __syncthreads();   // every wave is done reading the panel: reuse it as four 12 KiB staging slabs
        epilogue_gate_residual_lds<3>(acc, pr, reinterpret_cast<float*>(panel) + w * (32 * 96), 96 * w, p.bias, p.mm,
                                      p.gate_chunk, p.gated != 0, p.h);
    }
}

// =================================================================================================
// Fused MLP block:  h += gate_m * ( W2 gelu_erf( W1 (LN(h)(1+scale)+shift) + b1 ) + b2 )
// (latent_model.py:478-481, layers.py:77-84).  The 1536-wide hidden activation never leaves the CU:
// six chunks of 256 hidden units are produced (transposed MFMA -> 4 consecutive hidden units per
// lane -> exact-erf GELU -> bf16 -> LDS) and immediately consumed by the fc2 MFMAs.
// =================================================================================================
// (gelu_erf: common.h)
// Phase stamps for mdgen_profile_phase_trace (measurement only; p.trace is null in normal operation):
// slot 0 start, 1 after LN prologue, 2 after fc1(0) + GELU(0); chunk c = 1..11: 1+2c after the barrier + fc1(c), 2+2c
// after GELU(c) || fc2(c-1); 24 after fc2(11); 26 before the epilogue, 27 end, 28 HW_ID, 29 XCC_ID.
__device__ __forceinline__ void stamp(const MlpParams& p, int slot, unsigned long long v) {
    if (p.trace && lane_id() == 0) {
        const long i = ((long)blockIdx.x * 4 + wave_id()) * 32 + slot;
        if (i < p.trace_cap) p.trace[i] = v;
    }
}
__device__ __forceinline__ void stamp(const MlpParams& p, int slot) {
    if (p.trace) stamp(p, slot, __builtin_amdgcn_s_memtime());
}
// One chunk = 128 hidden units.  Per chunk a wave runs
//   X(c): fc1 for its 32 hidden units of chunk c                     48 MFMAs
//   Y(c): GELU of X(c)'s accumulators -> hbuf[c & 1]  INTERLEAVED with  fc2 of chunk c-1 from hbuf[(c-1) & 1]
//         8 k-steps of [6 MFMAs beside 4 GELU evaluations (~50 VALU, 8 of them transcendental) + 1 ds_write]
// and one LDS barrier (hbuf[c & 1] complete, everybody out of hbuf[(c-1) & 1]).
// Why: the GELU phase is VALU work of the same length as a GEMM phase, and the two workgroups of a CU run their
// phases in lockstep, so with GELU as a phase of its own the matrix pipe idled through it (PMC round 2: MFMA busy
// 38 %, VALU busy 42 %, next to nothing overlapped).  Inside ONE wave MFMAs and independent VALU work overlap
// almost perfectly when interleaved finely (profiles/r02_issue_rate.txt: 8 MFMAs + 48 FMAs take 292 cycles, the
// MFMAs alone 256) -- so the GELU of chunk c rides in the shadow of the fc2 MFMAs of chunk c-1 (double-buffered hbuf).
// PF1: weight prefetch depth (k-steps) of the fc1 stream.
constexpr int kHC = 128, kHRowB = kHC * 2, kNChunk = kF / kHC;

// GELU of one (a, tt) group: hidden units 32 w + 8 a + 4 hh .. +3 of token row tt * 32 + tk -> 8 bytes of hbuf
__device__ __forceinline__ void gelu_group(const f32x16* a1, const f32x4& ba, unsigned char* hw, int g, int w,
                                           int hh, int tk) {
    const int a = g >> 1, tt = g & 1;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = gelu_erf(a1[tt][4 * a + i] + ba[i]);
    *reinterpret_cast<u32x2*>(hw + panel_off(tt * 32 + tk, (32 * w + 8 * a + 4 * hh) * 2, kHRowB)) =
        u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
}

// Operands a stage needs for its first k-steps, requested during the LAST k-steps of the stage before it (weights
// are L2-resident but an L2 round trip is several hundred cycles, and both waves of a SIMD start their stages
// together: with the first loads issued at the stage's own start every stage -- 24 per panel -- began with that
// bubble; phase stamps showed both stage kinds at ~55 % of their MFMA / VALU bound).
constexpr int kPFX = 5;  // W1 fragments in flight per wave in the fc1 stage (see stage_x)
struct XPre {            // fc1 stage
    bf16x8 w[kPFX];      // W1 fragments of k-steps 0..kPFX-1
    bf16x8 a[2];         // panel fragments (two 32-token tiles) of k-step 0
};
struct YPre {            // fc2 stage
    bf16x8 w[2][3];      // W2 fragments of k-steps 0, 1 (three feature tiles each)
    bf16x8 a[2];         // hbuf fragments of k-step 0
};
constexpr int kW2S = 96 * 64;   // bf16x8 elements between two feature tiles of the packed W2

// X(c): a1 = fc1 of this wave's 32 hidden units of chunk c (transposed: D[hidden][token]), 24 k-steps.  Its last
// k-steps request what the following Y stage starts with.
// It also requests the fc1 bias of its own 32 hidden units (b1c), consumed by the GELU of that Y stage.
template <bool NEXT>
__device__ __forceinline__ void stage_x(const unsigned char* panel, const bf16x8* __restrict__ w1c, const XPre& pre, f32x16* a1,
                                        const float* b1c, f32x4& b0, const bf16x8* __restrict__ w2n,
                                        const unsigned char* hrn, YPre& nxt, unsigned long long* midstamp = nullptr) {
    // Weight fragments come from L2 with ~600 cycles of latency under load (in-kernel stamps: a k-step took 210 cycles
    // with three fragments in flight, 64 of them matrix-pipe time; with five 170), so the number in flight sets the
    // pace: kPFX, as many as the register file allows (six spill).
    constexpr int KS = 24, PF = kPFX;
    bf16x8 wring[PF + 1];
    bf16x8 aring[2][2];
#pragma unroll
    for (int i = 0; i < PF; ++i) wring[i] = pre.w[i];
    aring[0][0] = pre.a[0];
    aring[0][1] = pre.a[1];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (ks + PF < KS) wring[(ks + PF) % (PF + 1)] = w1c[(ks + PF) * 64];
        if (ks + 1 < KS) {
#pragma unroll
            for (int t = 0; t < 2; ++t) aring[(ks + 1) & 1][t] = panel_frag(panel, kRowB, t, ks + 1);
        }
        if (NEXT && (ks == KS - 3 || ks == KS - 2)) {
#pragma unroll
            for (int f = 0; f < 3; ++f) nxt.w[ks - (KS - 3)][f] = w2n[(size_t)f * kW2S + (ks - (KS - 3)) * 64];
        }
        if (NEXT && ks == KS - 1) {
#pragma unroll
            for (int t = 0; t < 2; ++t) nxt.a[t] = panel_frag(hrn, kHRowB, t, 0);
        }
        if (ks == KS - 4) b0 = *reinterpret_cast<const f32x4*>(b1c);   // bias of GELU group pair 0; the others just in time
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
            a1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wring[ks % (PF + 1)], aring[ks & 1][t], a1[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#ifdef MDGEN_DEV_MLP_STAMPX   // (experiment build, scripts/micro/mlp_stampx.py: a stamp half-way through one fc1 stage)
        if (midstamp && ks == 11) *midstamp = __builtin_amdgcn_s_memtime();
#endif
    }
}

// Y(c): y += hbuf[(c-1)&1] (64 x 128) . W2 slab^T, eight k-steps, each carrying one GELU group of chunk c (written to
// hbuf[c&1]); its last k-steps request what the following X stage starts with.
template <bool NEXT>
__device__ __forceinline__ void stage_y(const unsigned char* hr, const bf16x8* __restrict__ w2c, const YPre& pre, f32x16* y,
                                        const f32x16* a1, const f32x4& b0, const float* b1c, unsigned char* hw, int w, int hh,
                                        int tk, const unsigned char* panel, const bf16x8* __restrict__ w1n, XPre& nxt) {
    constexpr int KS = 8, PF = 2;
    bf16x8 wring[PF + 1][3];
    bf16x8 aring[2][2];
#pragma unroll
    for (int i = 0; i < PF; ++i)
#pragma unroll
        for (int f = 0; f < 3; ++f) wring[i][f] = pre.w[i][f];
    aring[0][0] = pre.a[0];
    aring[0][1] = pre.a[1];
    // fc1 bias of the GELU group pair in flight / of the next one, requested one k-step early: four live registers
    // instead of sixteen -- what pays for the deeper W1 ring of stage_x
    f32x4 bcur = b0, bnext = b0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if ((ks & 1) == 0 && ks) bcur = bnext;
        if ((ks & 1) && ks + 1 < KS) bnext = *reinterpret_cast<const f32x4*>(b1c + 8 * ((ks + 1) >> 1));
        if (ks + PF < KS) {
#pragma unroll
            for (int f = 0; f < 3; ++f) wring[(ks + PF) % (PF + 1)][f] = w2c[(size_t)f * kW2S + (ks + PF) * 64];
        }
        if (ks + 1 < KS) {
#pragma unroll
            for (int t = 0; t < 2; ++t) aring[(ks + 1) & 1][t] = panel_frag(hr, kHRowB, t, ks + 1);
        }
        if (NEXT && ks >= KS - kPFX) nxt.w[ks - (KS - kPFX)] = w1n[(ks - (KS - kPFX)) * 64];
        if (NEXT && ks == KS - 1) {
#pragma unroll
            for (int t = 0; t < 2; ++t) nxt.a[t] = panel_frag(panel, kRowB, t, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int f = 0; f < 3; ++f)
                y[t * 3 + f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aring[ks & 1][t], wring[ks % (PF + 1)][f], y[t * 3 + f], 0, 0, 0);
        gelu_group(a1, bcur, hw, ks, w, hh, tk);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x402, 9, 0);   // nine VALU / transcendental
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// PRE: the panel first runs the temporal attention sub-layer's out-projection + gated residual for its 64 tokens (k_proj<0>'s
// work, mha.py:397, latent_model.py:476) and normalises rows that are still in L2.
template <int PF1, bool PRE = false>
__global__ __launch_bounds__(256, 2) void k_mlp(const MlpParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[kPanelBytes + 2 * kPanel * kHRowB];
    unsigned char* panel = smem;
    unsigned char* hb0 = smem + kPanelBytes;
    unsigned char* hb1 = hb0 + kPanel * kHRowB;
    PanelRows* pr = reinterpret_cast<PanelRows*>(hb0);  // aliases hbuf 0: live only outside the chunk loop
    stamp(p, 0);
    if (p.trace) {
        stamp(p, 28, __builtin_amdgcn_s_getreg((31 << 11) | 4));    // HW_REG_HW_ID
        stamp(p, 29, __builtin_amdgcn_s_getreg((31 << 11) | 20));   // HW_REG_XCC_ID
    }
    setup_rows_linear(pr, (long)blockIdx.x * kPanel, p.nrows, p.mm);
    __syncthreads();
    const int w = __builtin_amdgcn_readfirstlane(wave_id());   // wave-uniform, and the compiler knows it
    const int lane = lane_id(), hh = lane >> 5, tk = lane & 31;
    if (PRE) {
        f32x16 acc[6];
        prologue_bf16<kC>(panel, pr, p.o);
        __syncthreads();
        zero_acc<6>(acc);
        wave_gemm<2, 3, 24, false>(panel, kRowB, 0, 0, p.wo + (size_t)(3 * w) * 24 * 64 + lane, 24 * 64, acc);
        __syncthreads();   // every wave is done reading the panel: reuse it as four 12 KiB staging slabs
        epilogue_gate_residual_lds<3>(acc, pr, reinterpret_cast<float*>(panel) + w * (32 * 96), 96 * w, p.bo, p.mm, p.gate_chunk_o,
                                      true, p.h);
        __syncthreads();   // (vmcnt(0) + barrier) the updated rows are in L2; the slabs are free
    }
    prologue_ln<false>(panel, pr, p.h, p.mm, p.shift_chunk, p.scale_chunk, 1e-6f);
    __syncthreads();
    stamp(p, 1);
    // per-lane views of the packed weights / bias: W1 fragment stream of chunk c starts at w1l + c * (4 * 24 * 64)
    const bf16x8* w1l = p.w1 + (size_t)w * 24 * 64 + lane;
    const bf16x8* w2l = p.w2 + (size_t)(3 * w) * 96 * 64 + lane;      // chunk c: + 8 c * 64
    const float* b1l = p.b1 + 32 * w + 4 * hh;                        // chunk c: + 128 c; group a: + 8 a
    constexpr size_t W1C = (size_t)4 * 24 * 64;
    f32x16 y[6];
    zero_acc<6>(y);
    f32x16 a1[2];
    XPre xp;
    YPre yp;
    {   // what X(0) starts with: nothing ran before it that could have prefetched
#pragma unroll
        for (int i = 0; i < kPFX; ++i) xp.w[i] = w1l[i * 64];
#pragma unroll
        for (int t = 0; t < 2; ++t) xp.a[t] = panel_frag(panel, kRowB, t, 0);
    }
    f32x4 b0;
    zero_acc<2>(a1);
    stage_x<false>(panel, w1l, xp, a1, b1l, b0, nullptr, nullptr, yp);
    {   // chunk 0: nothing to overlap its GELU with yet; request X(1)'s first operands under it
#pragma unroll
        for (int i = 0; i < kPFX; ++i) xp.w[i] = w1l[W1C + i * 64];
#pragma unroll
        for (int t = 0; t < 2; ++t) xp.a[t] = panel_frag(panel, kRowB, t, 0);
        f32x4 b4[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) b4[a] = *reinterpret_cast<const f32x4*>(b1l + 8 * a);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 8; ++g) gelu_group(a1, b4[g >> 1], hb0, g, w, hh, tk);
    }
    stamp(p, 2);
    lds_barrier();   // hbuf[0] is complete
#pragma unroll 1
    for (int c = 1; c < kNChunk; ++c) {
        unsigned char* hw = (c & 1) ? hb1 : hb0;           // GELU(c) writes it
        const unsigned char* hr = (c & 1) ? hb0 : hb1;     // fc2(c - 1) reads it
        // X(c): fc1 of chunk c; requests the first fc2(c - 1) operands
        zero_acc<2>(a1);
#ifdef MDGEN_DEV_MLP_STAMPX
        unsigned long long mid = 0;
        if (c == 5) stamp(p, 25);   // after the barrier, before fc1(5)
        stage_x<true>(panel, w1l + (size_t)c * W1C, xp, a1, b1l + c * kHC, b0, w2l + (size_t)8 * (c - 1) * 64, hr, yp,
                      c == 5 ? &mid : nullptr);
        if (c == 5) stamp(p, 30, mid);
#else
        stage_x<true>(panel, w1l + (size_t)c * W1C, xp, a1, b1l + c * kHC, b0, w2l + (size_t)8 * (c - 1) * 64, hr, yp);
#endif
        stamp(p, 1 + 2 * c);
        // Y(c): GELU(c) -> hw  ||  fc2(c - 1) from hr; requests the first operands of X(c + 1) (clamped at the end)
        const int cn = c + 1 < kNChunk ? c + 1 : c;
        stage_y<true>(hr, w2l + (size_t)8 * (c - 1) * 64, yp, y, a1, b0, b1l + c * kHC, hw, w, hh, tk, panel, w1l + (size_t)cn * W1C, xp);
        stamp(p, 2 + 2 * c);
        lds_barrier();   // hbuf[c & 1] is complete; every wave has left hbuf[(c - 1) & 1]
    }
    wave_gemm<2, 3, 8, false, 3>((kNChunk & 1) ? hb0 : hb1, kHRowB, 0, 0, w2l + (size_t)8 * (kNChunk - 1) * 64, kW2S, y);
    stamp(p, 24);
    __syncthreads();
    setup_rows_linear(pr, (long)blockIdx.x * kPanel, p.nrows, p.mm);
    __syncthreads();
    stamp(p, 26);
    epilogue_gate_residual_lds<3>(y, pr, reinterpret_cast<float*>(panel) + w * (32 * 96), 96 * w, p.b2, p.mm, p.gate_chunk,
                                  true, p.h);
    stamp(p, 27);
}

// ---- the same block for launches of at most one workgroup per CU (<= 256 panels: B = 1, the TPS shard) -------------------------------
// There a launch lasts exactly as long as ONE workgroup, and a lone wave per SIMD leaves the SIMD idle through every latency of its
// own chain (phase stamps, profiles/r04_experiments.txt #14: 7.0k cycles per chunk alone against 9.1k for TWO co-resident
// panels).  k_mlp8 gives the panel to eight waves: waves 0..3 run hidden chunks 0..5, waves 4..7 chunks 6..11 (each group with
// its own pair of hidden buffers, the same X / Y stages, the same barriers), i.e. two waves per SIMD working on one panel, and
// the two partial fc2 sums meet in LDS: group g hands the other group its partial of row tile 1 - g and finishes row tile g
// (bias, gate, residual).  The LayerNorm prologue is split by row batches, the out-projection prologue phase (PRE) by row tiles.
// Summation order differs from k_mlp's (two six-chunk partials instead of twelve chunks in sequence): fp32 rounding only.
//
// S > 1 (launches of at most a third of a workgroup per CU: B = 1, the IPA stack): the twelve hidden chunks are also divided over S
// WORKGROUPS (workgroup s, group g: chunks NC (2 s + g) .., NC = 12 / (2 S)), so that the chip's idle CUs take two thirds of the
// weight stream and of the MFMAs of a panel.  Each workgroup runs the prologue phases in full (out-projection: redundantly, into a
// PRIVATE copy p.hupd of the panel's updated residual rows -- nobody writes h while another workgroup may still read it), writes its
// fp32 partial of the fc2 product to p.part and bumps the panel's counter; the LAST ARRIVER adds the S partials in the order of s
// (the same bits whoever is last) and runs the gated residual epilogue from its private rows into h.
// Nobody waits for anybody.  Memory ordering: the partials are written and read with agent-scope atomic accesses (sc1: coherent across
// the XCDs' L2s element by element); producer: s_waitcnt vmcnt(0) on every wave, barrier, agent-scope atomic add; consumer: the
// same atomic add tells it that it is last, then agent-scope atomic loads.  Correct for any placement.  The S workgroups of a panel
// are nevertheless placed on ONE XCD (equal blockIdx % 8; the context's placement probe checks the residue -> XCD rule and the
// launcher only picks the form where it holds): that is where the form pays.
#ifdef MDGEN_DEV_MLP8_STAMPS   // (experiment build, scripts/r06/mlp8_stamps.py: per-wave s_memtime stamps of k_mlp8, held in SGPRs, one store branch)
__device__ unsigned long long g_mlp8_stamps[8192 * 16];
extern "C" int mdgen_dev_mlp8_stamps(void* host, size_t bytes) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_mlp8_stamps), bytes); }
#define MLP8_STAMP_DECL unsigned long long m8st[12] = {}
#define MLP8_STAMP(slot)                         \
    m8st[slot] = __builtin_amdgcn_s_memtime();   \
    __builtin_amdgcn_sched_barrier(0)
#define MLP8_STAMP_FLUSH                                                                      \
    if (lane_id() == 0 && (long)blockIdx.x * 8 + wave_id() < 8192) {                            \
        _Pragma("unroll") for (int k = 0; k < 12; ++k)                                          \
            g_mlp8_stamps[((long)blockIdx.x * 8 + wave_id()) * 16 + k] = m8st[k];               \
    }
#else
#define MLP8_STAMP_DECL
#define MLP8_STAMP(slot)
#define MLP8_STAMP_FLUSH
#endif
template <bool PRE, int S = 1>
__global__ __launch_bounds__(512, 1) void k_mlp8(const MlpParams p) {
    static_assert(kNChunk % (2 * S) == 0, "whole chunks per group");
    __shared__ __attribute__((aligned(16))) unsigned char smem[kPanelBytes + 4 * kPanel * kHRowB];
    __shared__ PanelRows prs;
    __shared__ PanelRows prs2;   // S > 1: the same rows in this workgroup's private copy (p.hupd) of the updated residual rows
    __shared__ int s_last;
    unsigned char* panel = smem;
    PanelRows* pr = &prs;
    const int w8 = __builtin_amdgcn_readfirstlane(wave_id());
    const int g = w8 >> 2, w = w8 & 3;
    const int lane = lane_id(), hh = lane >> 5, tk = lane & 31;
    unsigned char* hb0 = smem + kPanelBytes + (2 * g) * (kPanel * kHRowB);
    unsigned char* hb1 = hb0 + kPanel * kHRowB;
    float* slab = reinterpret_cast<float*>(smem) + w8 * (32 * 96);   // eight 12 KiB staging slabs over panel + hidden buffers
    int pn = blockIdx.x, sp = 0;
    if (S > 1) {
        const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
        sp = rest % S;
        pn = (rest / S) * 8 + xcd;
        if ((long)pn * kPanel >= p.nrows) return;
    }
    const int slot = pn * S + sp;   // this workgroup's slot in p.part / p.hupd
    MLP8_STAMP_DECL;
    MLP8_STAMP(0);
    setup_rows_linear(pr, (long)pn * kPanel, p.nrows, p.mm);
    constexpr bool PRIV = PRE && S > 1;   // residual rows come from / go to the private copy
    if (PRIV) {
        setup_rows_linear(&prs2, (long)pn * kPanel, p.nrows, p.mm);
        if (threadIdx.x < kPanel && prs2.tok[threadIdx.x] >= 0) prs2.tok[threadIdx.x] = slot * kPanel + threadIdx.x;   // (same thread wrote it)
    }
    __syncthreads();
    if (PRE) {
        if (threadIdx.x < 256) prologue_bf16<kC>(panel, pr, p.o);
        __syncthreads();
        MLP8_STAMP(1);
        f32x16 acc[3];
        zero_acc<3>(acc);
        wave_gemm<1, 3, 24, false>(panel, kRowB, g, 0, p.wo + (size_t)(3 * w) * 24 * 64 + lane, 24 * 64, acc);
        __syncthreads();   // every wave is done reading the panel
        MLP8_STAMP(2);
        epi_stage(acc, slab);
        if (PRIV) epi_rmw<8>(g, pr, slab, 96 * w, p.bo, p.mm, p.gate_chunk_o, true, p.h, &prs2, p.hupd);
        else epi_rmw<8>(g, pr, slab, 96 * w, p.bo, p.mm, p.gate_chunk_o, true, p.h);
        __syncthreads();   // (vmcnt(0) + barrier) the updated rows are in L2; the slabs are free
    }
    MLP8_STAMP(3);
    const PanelRows* prx = PRIV ? &prs2 : pr;          // where the MLP's input rows (and the residual of its epilogue) are read
    const float* hx = PRIV ? p.hupd : p.h;
    if (g == 0) prologue_ln<false, 0, 2>(panel, prx, hx, p.mm, p.shift_chunk, p.scale_chunk, 1e-6f, w, lane);
    else prologue_ln<false, 2, 4>(panel, prx, hx, p.mm, p.shift_chunk, p.scale_chunk, 1e-6f, w, lane);
    __syncthreads();
    MLP8_STAMP(4);
    const bf16x8* w1l = p.w1 + (size_t)w * 24 * 64 + lane;
    const bf16x8* w2l = p.w2 + (size_t)(3 * w) * 96 * 64 + lane;
    const float* b1l = p.b1 + 32 * w + 4 * hh;
    constexpr size_t W1C = (size_t)4 * 24 * 64;
    constexpr int NC = kNChunk / (2 * S);
    const int c0 = NC * (2 * sp + g);
    f32x16 y[6];
    zero_acc<6>(y);
    f32x16 a1[2];
    XPre xp;
    YPre yp;
    {
#pragma unroll
        for (int i = 0; i < kPFX; ++i) xp.w[i] = w1l[(size_t)c0 * W1C + i * 64];
#pragma unroll
        for (int t = 0; t < 2; ++t) xp.a[t] = panel_frag(panel, kRowB, t, 0);
    }
    f32x4 b0;
    zero_acc<2>(a1);
    stage_x<false>(panel, w1l + (size_t)c0 * W1C, xp, a1, b1l + c0 * kHC, b0, nullptr, nullptr, yp);
    {   // the group's first chunk: nothing to overlap its GELU with yet
#pragma unroll
        for (int i = 0; i < kPFX; ++i) xp.w[i] = w1l[(size_t)(c0 + 1) * W1C + i * 64];
#pragma unroll
        for (int t = 0; t < 2; ++t) xp.a[t] = panel_frag(panel, kRowB, t, 0);
        f32x4 b4[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) b4[a] = *reinterpret_cast<const f32x4*>(b1l + c0 * kHC + 8 * a);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) gelu_group(a1, b4[q >> 1], hb0, q, w, hh, tk);
    }
    lds_barrier();
    MLP8_STAMP(5);
#pragma unroll 1
    for (int i = 1; i < NC; ++i) {
        const int c = c0 + i;
        unsigned char* hw = (i & 1) ? hb1 : hb0;
        const unsigned char* hr = (i & 1) ? hb0 : hb1;
        zero_acc<2>(a1);
        stage_x<true>(panel, w1l + (size_t)c * W1C, xp, a1, b1l + c * kHC, b0, w2l + (size_t)8 * (c - 1) * 64, hr, yp);
        const int cn = i + 1 < NC ? c + 1 : c;
        stage_y<true>(hr, w2l + (size_t)8 * (c - 1) * 64, yp, y, a1, b0, b1l + c * kHC, hw, w, hh, tk, panel, w1l + (size_t)cn * W1C, xp);
        lds_barrier();
    }
    MLP8_STAMP(6);
    wave_gemm<2, 3, 8, false, 3>((NC & 1) ? hb0 : hb1, kHRowB, 0, 0, w2l + (size_t)8 * (c0 + NC - 1) * 64, kW2S, y);
    __syncthreads();   // panel and hidden buffers are dead: exchange area
    MLP8_STAMP(7);
    {   // this group's partial of the row tile the OTHER group finishes
        float* dst = reinterpret_cast<float*>(smem) + (size_t)(g * 4 + w) * (3 * 16 * 64);
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(f * 16 + r) * 64 + lane] = g == 0 ? y[3 + f][r] : y[f][r];   // (constant register indices)
    }
    __syncthreads();
    f32x16 z[3];
    {
        const float* src = reinterpret_cast<const float*>(smem) + (size_t)((1 - g) * 4 + w) * (3 * 16 * 64);
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float lo = g == 0 ? y[f][r] : src[(f * 16 + r) * 64 + lane];          // chunks 0..5
                const float hi = g == 0 ? src[(f * 16 + r) * 64 + lane] : y[3 + f][r];      // chunks 6..11
                z[f][r] = lo + hi;
            }
    }
    MLP8_STAMP(8);
    if (S > 1) {
        // this workgroup's partial: [slot][wave][3 tiles x 16 registers][lane] -- 256-byte runs per store
        float* mine = p.part + ((size_t)slot * 8 + w8) * (3 * 16 * 64) + lane;
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r)   // agent-scope atomic stores (global_store_dword ... sc1): written THROUGH this XCD's L2
                __hip_atomic_store(&mine[(f * 16 + r) * 64], z[f][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's partial stores have completed at agent scope
        __syncthreads();                                    // (also: exchange area read)
        if (threadIdx.x == 0) {
            unsigned* c = p.counters + pn;
            // Round 6: the hand-over no longer rests on where the S workgroups of a panel run.  The partials are written and read
            // with AGENT-scope atomic accesses (sc1: write-through stores, loads that bypass non-coherent cache lines), so they are
            // coherent across the XCDs' L2s element by element; every wave has waited for its stores (vmcnt(0)) before the barrier
            // above, and only then does the arrival counter move (agent-scope RMW).  A release fence instead (first form of this
            // round: __ATOMIC_ACQ_REL on the counter = buffer_wbl2, a write-back of the WHOLE L2) cost 4 us per launch at B = 1
            // (k_mlp8<true, 3> 45.3 -> 49.2 us, profiles/r06_experiments.txt #10).  The one-XCD placement stays a performance hint.
            const unsigned old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = old == (unsigned)(S - 1);
            if (old == (unsigned)(S - 1)) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
        __syncthreads();
        MLP8_STAMP(9);
        if (!s_last) {
            MLP8_STAMP_FLUSH
            return;
        }
        f32x16 zt[3];
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) zt[f][r] = 0.f;
#pragma unroll
        for (int o = 0; o < S; ++o) {   // fixed order (its own partial is read back like the others): the same bits whichever workgroup is last
            const float* src = p.part + ((size_t)(pn * S + o) * 8 + w8) * (3 * 16 * 64) + lane;
#pragma unroll
            for (int f = 0; f < 3; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r)   // agent-scope atomic loads (sc1): never a stale line of this CU's L1 or this XCD's L2
                    zt[f][r] += __hip_atomic_load(&src[(f * 16 + r) * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int f = 0; f < 3; ++f) z[f] = zt[f];
    } else {
        __syncthreads();   // exchange area read: the slabs may overwrite it
    }
    MLP8_STAMP(10);
    epi_stage(z, slab);
    if (PRIV) epi_rmw<8>(g, &prs2, slab, 96 * w, p.b2, p.mm, p.gate_chunk, true, p.hupd, pr, p.h);
    else epi_rmw<8>(g, pr, slab, 96 * w, p.b2, p.mm, p.gate_chunk, true, p.h);
    MLP8_STAMP(11);
    MLP8_STAMP_FLUSH
}

// =================================================================================================
// Affine LayerNorm (eps 1e-5) + the four IPA input projections in one GEMM:
//   out[token][672] = LN_affine(h) @ [linear_q | linear_kv | linear_q_points | linear_kv_points]^T + b
// (latent_model.py:373 ipa_norm; ipa.py:113-138).  fp32 output, row-major.
// =================================================================================================
__global__ __launch_bounds__(256, 2) void k_ln_linear(const LnLinearParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[sizeof(PanelRows) + kPanelBytes];
    PanelRows* pr = reinterpret_cast<PanelRows*>(smem);
    unsigned char* panel = smem + sizeof(PanelRows);
    setup_rows_linear(pr, (long)blockIdx.x * kPanel, p.nrows, p.mm);
    __syncthreads();
    prologue_ln<true>(panel, pr, p.h, p.mm, 1, 0, 1e-5f);   // mm.mod = [gamma(C) | beta(C)]: scale chunk 0, shift chunk 1
    __syncthreads();
    const int w = __builtin_amdgcn_readfirstlane(wave_id()), lane = lane_id(), hh = lane >> 5, n = lane & 31;
    const int ngroups = p.nout / 96;
    for (int g = w; g < ngroups; g += 4) {
        f32x16 acc[6];
        zero_acc<6>(acc);
        wave_gemm<2, 3, 24, false>(panel, kRowB, 0, 0, p.w + (size_t)(3 * g) * 24 * 64 + lane, 24 * 64, acc);
        float b[3];
#pragma unroll
        for (int f = 0; f < 3; ++f) b[f] = p.bias[96 * g + 32 * f + n];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tk = pr->tok[t * 32 + mfma_row(r, hh)];
                if (tk >= 0) {
                    float* op = p.out + (long)tk * p.nout + 96 * g + n;
#pragma unroll
                    for (int f = 0; f < 3; ++f) op[32 * f] = acc[t * 3 + f][r] + b[f];
                }
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// =================================================================================================
// FinalLayer (layers.py:70-74): v = Linear_{C->D}( LN(h)(1+scale)+shift ) fused with the Euler
// update x += dt*v (integrators.py:106, torchdiffeq fixed-grid Euler) when p.euler != 0.
// =================================================================================================
__global__ __launch_bounds__(256, 2) void k_final(const FinalParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[sizeof(PanelRows) + kPanelBytes];
    PanelRows* pr = reinterpret_cast<PanelRows*>(smem);
    unsigned char* panel = smem + sizeof(PanelRows);
    setup_rows_linear(pr, (long)blockIdx.x * kPanel, p.nrows, p.mm);
    __syncthreads();
    prologue_ln<false>(panel, pr, p.h, p.mm, p.shift_chunk, p.scale_chunk, 1e-6f);
    __syncthreads();
    const int w = __builtin_amdgcn_readfirstlane(wave_id()), lane = lane_id(), hh = lane >> 5, n = lane & 31;
    if (w < 2) {
        f32x16 acc[1];
        zero_acc<1>(acc);
        wave_gemm<1, 1, 24, false>(panel, kRowB, w, 0, p.w + lane, 24 * 64, acc);
        // Euler read-modify-write of x: all 16 loads unconditional and issued together (see prologue_ln)
        const bool col_ok = n < p.D;
        const int nc = col_ok ? n : 0;
        const float b = p.bias[nc];
        int tks[16];
        float xv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            tks[r] = pr->tok[w * 32 + mfma_row(r, hh)];
            xv[r] = 0.f;
        }
        if (p.euler) {
#pragma unroll
            for (int r = 0; r < 16; ++r) xv[r] = p.x[(long)(tks[r] < 0 ? 0 : tks[r]) * p.D + nc];
        }
        float* dst = p.euler ? p.x : p.out;
        const float dt = p.euler ? p.dt : 1.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (col_ok && tks[r] >= 0) dst[(long)tks[r] * p.D + n] = xv[r] + dt * (acc[0][r] + b);
    }
}

// ---- launchers ----------------------------------------------------------------------------------
// Four or eight waves per 64-row panel (k_mlp / k_mlp8, k_ln_qkv<false> / k_ln_qkv8): eight where a launch is at most one
// workgroup per CU, unless the caller forces one form (option panel_waves: 4 / 8).
int panel_waves_for(long grid, int forced, int ncu) { return forced == 4 || forced == 8 ? forced : (grid <= ncu ? 8 : 4); }

void launch_ln_qkv(const QkvParams& p, bool small, hipStream_t s, bool pre, int waves, bool split, bool half) {
    if (pre) {
        const int grid = p.ax.nseq * p.panels_per_seq;
        hipLaunchKernelGGL((k_ln_qkv<false, true>), dim3(grid), dim3(256), 0, s, p);
        return;
    }
    if (small) {
        const int grid = (int)((p.nrows + kPanel - 1) / kPanel);
        hipLaunchKernelGGL(k_ln_qkv<true>, dim3(grid), dim3(256), 0, s, p);
    } else {
        const int grid = p.ax.nseq * p.panels_per_seq;
        // (half: p.panels_per_seq counts 32-position panels)
        if (waves == 8 && split && half) hipLaunchKernelGGL((k_ln_qkv8<true, true>), dim3(2 * grid), dim3(512), 0, s, p);
        else if (waves == 8 && split) hipLaunchKernelGGL(k_ln_qkv8<true>, dim3(2 * grid), dim3(512), 0, s, p);
        else if (waves == 8) hipLaunchKernelGGL(k_ln_qkv8<false>, dim3(grid), dim3(512), 0, s, p);
        else hipLaunchKernelGGL(k_ln_qkv<false>, dim3(grid), dim3(256), 0, s, p);
    }
}
// placement probe (mdgen_ctx_create): the XCD each of `nblocks` workgroups ran on (HW_REG_XCC_ID, 4 bits)
__global__ void k_xcc_probe(int* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf);
}
void launch_xcc_probe(int* out, int nblocks, hipStream_t s) { hipLaunchKernelGGL(k_xcc_probe, dim3(nblocks), dim3(64), 0, s, out); }

void launch_ln_qkv_attn4(const QkvParams& p, bool fuse_proj, hipStream_t s, bool half) {
    const int grid = (int)((p.nrows + kPanel - 1) / kPanel);
#ifdef MDGEN_DEV_ATTN4_FULL   // (experiment build, A/B of the half-panel form: every launch takes 64-row panels, whatever the tag says)
    half = false;
#endif
    if (fuse_proj && half) hipLaunchKernelGGL((k_ln_qkv_attn4<true, true>), dim3((unsigned)((p.nrows + 31) / 32)), dim3(256), 0, s, p);
    else if (fuse_proj) hipLaunchKernelGGL((k_ln_qkv_attn4<true>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_ln_qkv_attn4<false>), dim3(grid), dim3(256), 0, s, p);
}
void launch_proj(const ProjParams& p, int mode, hipStream_t s) {
    const int grid = (int)((p.nrows + kPanel - 1) / kPanel);
    if (mode == 0) hipLaunchKernelGGL(k_proj<0>, dim3(grid), dim3(256), 0, s, p);
    else if (mode == 1) hipLaunchKernelGGL(k_proj<1>, dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(k_proj<2>, dim3(grid), dim3(256), 0, s, p);
}
void launch_mlp(const MlpParams& p, hipStream_t s, int waves) {
    const int grid = (int)((p.nrows + kPanel - 1) / kPanel);
    if (waves == 8 && !p.trace && p.part) {   // hidden chunks over kMlpSplit workgroups per panel (same XCD), last arriver finishes
        const int g3 = (grid + 7) / 8 * 8 * kMlpSplit;
        if (p.o) hipLaunchKernelGGL((k_mlp8<true, kMlpSplit>), dim3(g3), dim3(512), 0, s, p);
        else hipLaunchKernelGGL((k_mlp8<false, kMlpSplit>), dim3(g3), dim3(512), 0, s, p);
        return;
    }
    if (waves == 8 && !p.trace) {   // (the phase stamps stay with k_mlp: mdgen_profile_phase_trace selects the four-wave kernel)
        if (p.o) hipLaunchKernelGGL((k_mlp8<true>), dim3(grid), dim3(512), 0, s, p);
        else hipLaunchKernelGGL((k_mlp8<false>), dim3(grid), dim3(512), 0, s, p);
        return;
    }
    if (p.o) hipLaunchKernelGGL((k_mlp<3, true>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_mlp<3>), dim3(grid), dim3(256), 0, s, p);
}
void launch_ln_linear(const LnLinearParams& p, hipStream_t s) {
    const int grid = (int)((p.nrows + kPanel - 1) / kPanel);
    hipLaunchKernelGGL(k_ln_linear, dim3(grid), dim3(256), 0, s, p);
}
void launch_final(const FinalParams& p, hipStream_t s) {
    const int grid = (int)((p.nrows + kPanel - 1) / kPanel);
    hipLaunchKernelGGL(k_final, dim3(grid), dim3(256), 0, s, p);
}

}  // namespace mdg
