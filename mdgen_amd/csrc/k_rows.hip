// k_rows.hip -- row-owner token-local kernels (rows.h): the fused MLP block with activations in registers and the
// weights arriving as one LDS-DMA stream shared by the waves of a workgroup.  gfx950 only.
#include "kernels.h"
#include "rows.h"

namespace mdg {

// =================================================================================================
// Fused MLP block:  h += gate_m * ( W2 gelu_erf( W1 (LN(h)(1+scale)+shift) + b1 ) + b2 )
// (latent_model.py:478-481, layers.py:77-84), row-owner form.
//
// Per wave (32 tokens): X = LN rows as 24 B-operand fragments (96 registers); the hidden dimension is processed in 24
// chunks of 64 units: stage X(c) = fc1 of chunk c (2 hidden tiles x 24 k-steps = 48 MFMAs, accumulators a1[c & 1]),
// GELU(c) (32 values per lane -> the 4 B-operand fragments hf[c & 1] of fc2's k-steps 4c .. 4c + 3, no data
// movement), stage Y(c) = fc2 partial sums (4 k-steps x 12 feature tiles = 48 MFMAs into y[12], 192 registers).
// Software pipeline: iteration c runs the MFMAs of X(c + 1) and Y(c - 1) -- eight blocks of 12, alternating -- with
// GELU(c) riding beside them, one group of four values per block (~4 VALU per MFMA).  The weight stream
// (api.hip mlp_stream_table) is laid out in exactly that order:
//     [X(0)] [X(1)] { [X(c+1) ks 0-5] [Y(c-1) kk 0] [X ks 6-11] [Y kk 1] [X ks 12-17] [Y kk 2] [X ks 18-23] [Y kk 3] } c = 1..22 [Y(22)] [Y(23)]
// 2304 fragments = 96 ring slots; one iteration = 96 fragments = the whole ring, so every LDS address in the loop
// body is a compile-time constant.
// =================================================================================================
struct MlpPipe {
    f32x16 y[12];
    f32x16 a1[2][2];
    bf16x8 hf[2][4];
    bf16x8 wr[kWRing];
    f32x4 bias;       // fc1 bias (LDS copy of b1) the accumulator registers of the GELU group in flight are re-armed with
    float gx[4], gp[4], gq[4];   // GELU state of the four values of the group in flight
};

// GELU of group g (0..7) of a chunk = hidden tile g >> 2, accumulator registers 4 (g & 3) .. + 3 -> half of the fragment
// hf[2 (g >> 2) + ((g & 3) >> 1)].  gelu(x) = x * Phi(x) exactly as common.h gelu_erf, cut into TWELVE stages, each of
// which advances all four values by one operation: the instructions of a stage are independent of each other (a lone
// wave has nobody to hide a dependent VALU chain behind: with one value per stage the measured pace was 52 cycles per
// MFMA), at most two of them transcendental, and every MFMA of a block carries one stage (4-5 VALU instructions; the
// matrix pipe hides ~5, MI355X_MICROARCH "one wave per SIMD").  The fc1 bias is not added here: the accumulators
// START from it -- stages 7, 10, 11 re-arm the four registers with the bias of the chunk two further on (REARM).
__device__ __forceinline__ void acc_rearm(f32x16& t, int r, float b) {
    float z;
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(z) : "v"(b));
    t[r] = z;
}
// Phi as a logistic of an odd polynomial with THREE coefficients (the panel kernel's gelu_erf has five): max abs error of
// gelu 2.9e-5 (scripts/fit_gelu.py --rows), against the bf16 rounding applied right after (relative 2^-9).  The leading
// coefficient has the wrong sign for large |x|, hence the clamp of x^2 at 64 (|x| > 8: Phi is 0 or 1 in fp32 anyway).
// Constants carry the factor -log2(e): the exponential is a bare v_exp_f32.
constexpr float kGelu3C0 = -2.301208258e+00f, kGelu3C1 = -1.066924557e-01f, kGelu3C2 = 1.000115648e-03f;
template <int ST, bool REARM>
__device__ __forceinline__ void gelu_stage(MlpPipe& m, f32x16 (&a1r)[2], int g, bf16x8 (&hfw)[4]) {
    const int tile = g >> 2, a = g & 3;
    if (ST == 0) {
        // explicit accumulator-file reads: left to hipcc, the whole 16-register tuple is copied to VGPRs at its first use
#pragma unroll
        for (int j = 0; j < 4; ++j) asm("v_accvgpr_read_b32 %0, %1" : "=v"(m.gx[j]) : "a"(a1r[tile][4 * a + j]));
    } else if (ST == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) m.gq[j] = m.gx[j] * m.gx[j];
    } else if (ST == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) m.gq[j] = fminf(m.gq[j], 64.0f);
    } else if (ST == 3) {
#pragma unroll
        for (int j = 0; j < 4; ++j) m.gp[j] = kGelu3C2 * m.gq[j] + kGelu3C1;
    } else if (ST == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) m.gp[j] = m.gp[j] * m.gq[j] + kGelu3C0;
    } else if (ST == 5) {
#pragma unroll
        for (int j = 0; j < 4; ++j) m.gp[j] = m.gx[j] * m.gp[j];
    } else if (ST == 6) {
        m.gp[0] = __builtin_amdgcn_exp2f(m.gp[0]);
        m.gp[1] = __builtin_amdgcn_exp2f(m.gp[1]);
        if (REARM) {
            acc_rearm(a1r[tile], 4 * a + 0, m.bias[0]);
            acc_rearm(a1r[tile], 4 * a + 1, m.bias[1]);
        }
    } else if (ST == 7) {
        m.gp[2] = __builtin_amdgcn_exp2f(m.gp[2]);
        m.gp[3] = __builtin_amdgcn_exp2f(m.gp[3]);
        m.gp[0] = 1.0f + m.gp[0];
        m.gp[1] = 1.0f + m.gp[1];
    } else if (ST == 8) {
        m.gp[0] = __builtin_amdgcn_rcpf(m.gp[0]);
        m.gp[1] = __builtin_amdgcn_rcpf(m.gp[1]);
        m.gp[2] = 1.0f + m.gp[2];
        m.gp[3] = 1.0f + m.gp[3];
    } else if (ST == 9) {
        m.gp[2] = __builtin_amdgcn_rcpf(m.gp[2]);
        m.gp[3] = __builtin_amdgcn_rcpf(m.gp[3]);
        m.gx[0] = m.gx[0] * m.gp[0];
        m.gx[1] = m.gx[1] * m.gp[1];
    } else if (ST == 10) {
        m.gx[2] = m.gx[2] * m.gp[2];
        m.gx[3] = m.gx[3] * m.gp[3];
        if (REARM) {
            acc_rearm(a1r[tile], 4 * a + 2, m.bias[2]);
            acc_rearm(a1r[tile], 4 * a + 3, m.bias[3]);
        }
    } else {
        const int kk = 2 * tile + (a >> 1), e0 = 4 * (a & 1);
        hfw[kk][e0 + 0] = (__bf16)m.gx[0];   // two v_cvt_pk_bf16_f32
        hfw[kk][e0 + 1] = (__bf16)m.gx[1];
        hfw[kk][e0 + 2] = (__bf16)m.gx[2];
        hfw[kk][e0 + 3] = (__bf16)m.gx[3];
    }
}
template <bool REARM, int ST = 0>
__device__ __forceinline__ void gelu_stages_from(MlpPipe& m, f32x16 (&a1r)[2], int g, bf16x8 (&hfw)[4]) {
    gelu_stage<ST, REARM>(m, a1r, g, hfw);
    if constexpr (ST < 11) gelu_stages_from<REARM, ST + 1>(m, a1r, g, hfw);
}
// The four accumulator registers a GELU group has consumed start their next accumulation (the chunk two further on, which
// lands in the same registers) from the fc1 bias of that chunk: no zeroing, no bias add.
__device__ __forceinline__ void rearm(f32x16 (&a1r)[2], int g, const f32x4& b) {
    const int tile = g >> 2, a = g & 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc_rearm(a1r[tile], 4 * a + j, b[j]);
}
// all 32 accumulator registers of a chunk <- its fc1 bias (pipeline fill); b1c = LDS bias row of the chunk + 4 hh
__device__ __forceinline__ void arm_chunk(f32x16 (&a1)[2], const float* b1c) {
#pragma unroll
    for (int g = 0; g < 8; ++g) rearm(a1, g, *reinterpret_cast<const f32x4*>(b1c + 32 * (g >> 2) + 8 * (g & 3)));
}

// DMA number D0 + q / STRIDE of the wave's share of a slot (q is a constant after unrolling)
template <class WS, int D0, int STRIDE>
__device__ __forceinline__ void issue_q(const WS& ws, long slot, int q) {
    switch (q / STRIDE) {
        case 0: ws.template issue<D0 + 0>(slot); break;
        case 1: ws.template issue<D0 + 1>(slot); break;
        case 2: ws.template issue<D0 + 2>(slot); break;
        case 3: if constexpr (12 / STRIDE > 3) ws.template issue<D0 + 3>(slot); break;
        case 4: if constexpr (12 / STRIDE > 4) ws.template issue<D0 + 4>(slot); break;
        case 5: if constexpr (12 / STRIDE > 5) ws.template issue<D0 + 5>(slot); break;
        case 6: if constexpr (12 / STRIDE > 6) ws.template issue<D0 + 6>(slot); break;
        case 7: if constexpr (12 / STRIDE > 7) ws.template issue<D0 + 7>(slot); break;
        case 8: if constexpr (12 / STRIDE > 8) ws.template issue<D0 + 8>(slot); break;
        case 9: if constexpr (12 / STRIDE > 9) ws.template issue<D0 + 9>(slot); break;
        case 10: if constexpr (12 / STRIDE > 10) ws.template issue<D0 + 10>(slot); break;
        default: if constexpr (12 / STRIDE > 11) ws.template issue<D0 + 11>(slot); break;
    }
}

template <bool REARM>
__device__ __forceinline__ void gelu_stage_q(MlpPipe& m, f32x16 (&a1r)[2], int g, int q, bf16x8 (&hfw)[4]) {
    switch (q) {   // q is a constant after unrolling
        case 0: gelu_stage<0, REARM>(m, a1r, g, hfw); break;
        case 1: gelu_stage<1, REARM>(m, a1r, g, hfw); break;
        case 2: gelu_stage<2, REARM>(m, a1r, g, hfw); break;
        case 3: gelu_stage<3, REARM>(m, a1r, g, hfw); break;
        case 4: gelu_stage<4, REARM>(m, a1r, g, hfw); break;
        case 5: gelu_stage<5, REARM>(m, a1r, g, hfw); break;
        case 6: gelu_stage<6, REARM>(m, a1r, g, hfw); break;
        case 7: gelu_stage<7, REARM>(m, a1r, g, hfw); break;
        case 8: gelu_stage<8, REARM>(m, a1r, g, hfw); break;
        case 9: gelu_stage<9, REARM>(m, a1r, g, hfw); break;
        case 10: gelu_stage<10, REARM>(m, a1r, g, hfw); break;
        default: gelu_stage<11, REARM>(m, a1r, g, hfw); break;
    }
}

// One block of 12 fragments, one fenced scheduling region per fragment: [look-ahead LDS read of fragment I + PF]
// [MFMA of fragment I] [a slice of the GELU group] (+ this wave's share of the ring refill).
//   I0    ring-relative index of its first fragment (multiple of 12)
//   KIND  0: X block (k-steps 6 KI .. 6 KI + 5 of both hidden tiles -> a1w), 1: Y block (k-step KI of the chunk, 12 feature tiles),
//         2: out-projection block (k-step KI of the 24, 12 feature tiles, B operand = xf[KI])
//   GG    GELU group riding along (reads a1r, writes hfw), or -1.  REARM: afterwards its four accumulator registers are
//         re-armed with the bias at b1n (LDS; the same group of the chunk two further on)
//   BARVM >= 0: the block opens a ring slot: barrier with that vmcnt first
//   FILL  issue this wave's DMAs of slot `fill_slot` (half of them per block: a slot is two blocks)
//   NLOOK look-ahead reads are issued for the first NLOOK fragments only (end of the stream)
template <int NW, int I0, int KIND, int KI, int GG, bool REARM, int BARVM, bool FILL, int NLOOK = 12, class WS>
__device__ __forceinline__ void pipe_block(MlpPipe& m, const bf16x8 (&xf)[24], f32x16 (&a1w)[2], f32x16 (&a1r)[2],
                                           bf16x8 (&hfw)[4], const bf16x8 (&hfr)[4], const float* b1n,
                                           const unsigned char* ring_lane, const WS& ws, long fill_slot) {
    constexpr int FPW = WS::FPW, DPB = FPW / 2, STRIDE = 12 / DPB;
    if (BARVM >= 0) ring_barrier<BARVM>();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        const int I = I0 + q;
        if (FILL && q % STRIDE == 0) issue_q<WS, ((I0 / 12) & 1) * DPB, STRIDE>(ws, fill_slot, q);
        if (q < NLOOK) m.wr[(I + kWPF) % kWRing] = *reinterpret_cast<const bf16x8*>(ring_lane + ((I + kWPF) % kRingFrags) * 1024);
        if (GG >= 0 && REARM && q == 0) m.bias = *reinterpret_cast<const f32x4*>(b1n + 32 * (GG >> 2) + 8 * (GG & 3));
        if (KIND == 0) {
            const int ks = 6 * KI + (q >> 1), tile = q & 1;
            a1w[tile] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m.wr[I % kWRing], xf[ks], a1w[tile], 0, 0, 0);
        } else if (KIND == 1) {
            m.y[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m.wr[I % kWRing], hfr[KI], m.y[q], 0, 0, 0);
        } else {   // KIND 2: out-projection, k-step KI of the attention output rows (xf), 12 feature tiles
            m.y[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m.wr[I % kWRing], xf[KI], m.y[q], 0, 0, 0);
        }
#ifndef MDGEN_DEV_ROWS_NOGELU   // (experiment build, timing only: the main loop without its VALU work)
        if (GG >= 0) gelu_stage_q<REARM>(m, a1r, GG, q, hfw);
#endif
        __builtin_amdgcn_sched_barrier(0);
    }
}

// a whole GELU group without MFMAs beside it (pipeline fill / drain only)
template <bool REARM>
__device__ __forceinline__ void gelu_group_plain(MlpPipe& m, f32x16 (&a1r)[2], const float* b1n, int g, bf16x8 (&hfw)[4]) {
    if (REARM) m.bias = *reinterpret_cast<const f32x4*>(b1n + 32 * (g >> 2) + 8 * (g & 3));
    gelu_stages_from<REARM>(m, a1r, g, hfw);
}

// Phase stamps (measurement only, p.trace null in normal operation): s_memtime values are collected in SGPRs and
// written by ONE branch at the very end -- an `if (p.trace)` store per stamp splits the kernel into basic blocks around
// which hipcc's register allocator spilled 170 registers per lane.
#define ROWS_STAMP(i)                                  \
    st[i] = __builtin_amdgcn_s_memtime();              \
    __builtin_amdgcn_sched_barrier(0)

// out-projection blocks KS .. 23 of the fused form (one block = one k-step of the attention output rows x 12 feature tiles;
// two blocks per ring slot, slot s = global slot: barrier + refill of slot s + 3)
template <int NW, int KS, class WS>
__device__ __forceinline__ void proj_blocks(MlpPipe& m, const bf16x8 (&xf)[24], const unsigned char* ring_lane, const WS& ws) {
    constexpr int FPW = WS::FPW;
    pipe_block<NW, (12 * KS) % kRingFrags, 2, KS, -1, false, (KS & 1) == 0 ? FPW : -1, true>(
        m, xf, m.a1[0], m.a1[1], m.hf[0], m.hf[1], nullptr, ring_lane, ws, KS / 2 + 3);
    if constexpr (KS + 1 < 24) proj_blocks<NW, KS + 1>(m, xf, ring_lane, ws);
}

// PROJ: the temporal attention's out-projection + gated residual (mha.py:397, latent_model.py:476) runs in the same kernel,
// ahead of the MLP, on the same 32 rows per wave: the attention output rows are loaded straight into B-operand fragments,
// the 288 fragments of W_o lead the weight stream (12 ring slots = three ring revolutions), the updated residual rows are
// written once and KEPT in registers for the MLP's LayerNorm -- one kernel, one read of the rows less (98 MB at cfg-2), no
// launch boundary (k_proj<0> was an HBM-bound 58 us kernel of its own).
// MODLDS: every 32-row tile of the launch lies inside one modulation group (launch_mlp_rows checks the ModMap): the wave's
// scale / shift / gate chunks are DMA'd into LDS once and read from there (rows_norm_lds, rows_gate_residual_lds).
// FOLD (round 6; MODLDS, no PROJ): the launch's gate is one vector and is folded into the stream's W2 fragments and into b2'
// (p.b2g; k_pack_fold, once per call): the fc2 accumulators start from the residual rows + b2' (rows_norm_lds_fold) and the
// epilogue only stores -- one HBM read of the rows instead of two (98 MB of 332 per launch at cfg-2).
// TAIL (with FOLD; the trunk's last layer in a sampling call): the FinalLayer -- LN + modulate, Linear C -> D, Euler update of x
// (layers.py:57-74, integrators.py:106; k_final's work) -- runs on the updated rows straight from the accumulators (the row image is
// the LayerNorm image) and the rows are never stored.
template <int NW, bool PROJ, bool MODLDS, bool FOLD = false, bool TAIL = false>
__global__ __launch_bounds__(NW * 64, 1) void k_mlp_rows(const MlpRowsParams p) {
    static_assert(!FOLD || (MODLDS && !PROJ), "the folded form: one modulation group per launch, no fused out-projection");
    static_assert(!TAIL || FOLD, "the tail runs on the folded form's accumulators");
    // ring | fc1 bias | slack: the last re-arm reads the (non-existent) chunk 24 | per wave: scale, shift, gate chunks (2 KiB slots)
    // | TAIL: the final layer's shift, scale chunks (2 KiB slots, shared by the waves)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kRingBytes + 8192 + NW * 6144 + (TAIL ? 4096 : 0)];
    constexpr int NPRE = PROJ ? 12 : 0;   // ring slots of the out-projection ahead of the MLP stream
    using WS = WStream<NW, NPRE>;
    constexpr int FPW = WS::FPW;
    const int w = __builtin_amdgcn_readfirstlane(wave_id()), lane = lane_id(), hh = lane >> 5, n = lane & 31;
    unsigned long long st[8] = {};
    ROWS_STAMP(0);
    WS ws{p.wstream, p.wo_stream, lds_addr(smem), (unsigned)lane * 16u, w};
    // fc1 bias -> LDS, six 1 KiB DMAs (wave w: pieces w, w + NW, ...), BEFORE the stream's: barrier 0 then certifies them too.
    // (A load + ds_write loop here cost two serialised memory round trips ahead of the row loads.)
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (i % NW == w)
            dma_frag<0, true>(reinterpret_cast<const unsigned char*>(p.b1) + i * 1024, (unsigned)lane * 16u,
                              lds_addr(smem) + kRingBytes + i * 1024);
    if (TAIL) {   // final adaLN chunks 0 (shift), 1 (scale) -> LDS: wave c % NW brings chunk c (two 1 KiB DMAs, upper half into the slot's slack)
#pragma unroll
        for (int c = 0; c < 2; ++c)
            if (c % NW == w) {
                const unsigned char* src = reinterpret_cast<const unsigned char*>(p.tail_mod) + c * (kC * 4);
                dma_frag<0, true>(src, (unsigned)lane * 16u, lds_addr(smem) + kRingBytes + 8192 + NW * 6144 + c * 2048);
                dma_frag<1024, false>(src, (unsigned)lane * 16u, lds_addr(smem) + kRingBytes + 8192 + NW * 6144 + c * 2048);
            }
    }
    const long t = ((long)blockIdx.x * NW + w) * 32 + n;
    const int tok = t < p.nrows ? (int)t : -1;
    // The wave's modulation vectors (scale, shift, gate: 1536 B each) by DMA into LDS when all its rows share them (always,
    // unless a tile straddles two samples): two 1 KiB DMAs per chunk, the second one's upper half lands in the slot's slack.
    float* modl = reinterpret_cast<float*>(smem + kRingBytes + 8192 + w * 6144);
    {
        if (MODLDS) {
            const long t0 = ((long)blockIdx.x * NW + w) * 32;
            const unsigned char* mb = reinterpret_cast<const unsigned char*>(p.mm.mod + p.mm.row_off(t0 < p.nrows ? t0 : 0));
            const int chunk[3] = {p.scale_chunk, p.shift_chunk, p.gate_chunk};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                // (FOLD: the third slot holds b2' = gate * b2 instead of the gate)
                const unsigned char* src = FOLD && c == 2 ? reinterpret_cast<const unsigned char*>(p.b2g) : mb + (long)chunk[c] * (kC * 4);
                dma_frag<0, true>(src, (unsigned)lane * 16u, lds_addr(modl) + c * 2048);
                dma_frag<1024, false>(src, (unsigned)lane * 16u, lds_addr(modl) + c * 2048);
            }
        }
    }
    ws.issue_slot(0);
    ws.issue_slot(1);
    ws.issue_slot(2);
    float* b1s = reinterpret_cast<float*>(smem + kRingBytes);
    bf16x8 xf[24];
    MlpPipe m;
    const unsigned char* ring_lane = smem + lane * 16;
    if (PROJ) {
        rows_load_bf16(p.o, tok, xf);
        rows_acc_init(m.y, p.bo);
        ring_barrier<FPW>();   // barrier 0: slots 0 and 1 of the out-projection (and the fc1 bias table) have landed
#pragma unroll
        for (int i = 0; i < kWPF; ++i) m.wr[i] = *reinterpret_cast<const bf16x8*>(ring_lane + i * 1024);
        // the first block's barrier is the one just passed: blocks 0 .. 23 open slots 0 .. 11 at their even members
        pipe_block<NW, 0, 2, 0, -1, false, -1, true>(m, xf, m.a1[0], m.a1[1], m.hf[0], m.hf[1], nullptr, ring_lane, ws, 3);
        proj_blocks<NW, 1>(m, xf, ring_lane, ws);
        ROWS_STAMP(6);
        f32x4 v[48];
        rows_gate_residual_keep(m.y, tok, p.mm, p.gate_chunk_o, p.h, v);
        ROWS_STAMP(7);
        if (MODLDS) rows_norm_lds(v, tok, modl, modl + 512, 1e-6f, xf);
        else rows_norm(v, tok, p.mm, p.shift_chunk, p.scale_chunk, 1e-6f, xf);
    } else if (MODLDS) {
        f32x4 v[48];
        rows_load(p.h, tok, v);
        // the row loads AND this wave's modulation DMAs have landed (the asm keeps hipcc from lifting the LDS reads above it)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (FOLD) rows_norm_lds_fold(v, tok, modl, modl + 512, modl + 1024, 1e-6f, xf, m.y);
        else rows_norm_lds(v, tok, modl, modl + 512, 1e-6f, xf);
    } else {
        rows_ln(p.h, tok, p.mm, p.shift_chunk, p.scale_chunk, 1e-6f, xf);
    }
    ROWS_STAMP(1);
    if (!FOLD) rows_acc_init(m.y, p.b2);
    const float* b1l = b1s + 4 * hh;   // LDS bias row of chunk c: + 64 c (this lane half's four units of every group of 8)
    // ---- P0: X(0).  Barrier 0 certifies slots 0 and 1 (only the FPW DMAs of slot 2 may be in flight) and the LDS copy of b1.
    ring_barrier<FPW>();
#pragma unroll
    for (int i = 0; i < kWPF; ++i) m.wr[i] = *reinterpret_cast<const bf16x8*>(ring_lane + i * 1024);
    arm_chunk(m.a1[0], b1l);
    arm_chunk(m.a1[1], b1l + 64);
    pipe_block<NW, 0, 0, 0, -1, false, -1, true>(m, xf, m.a1[0], m.a1[1], m.hf[0], m.hf[1], b1l, ring_lane, ws, NPRE + 3);
    pipe_block<NW, 12, 0, 1, -1, false, -1, true>(m, xf, m.a1[0], m.a1[1], m.hf[0], m.hf[1], b1l, ring_lane, ws, NPRE + 3);
    pipe_block<NW, 24, 0, 2, -1, false, FPW, true>(m, xf, m.a1[0], m.a1[1], m.hf[0], m.hf[1], b1l, ring_lane, ws, NPRE + 4);
    pipe_block<NW, 36, 0, 3, -1, false, -1, true>(m, xf, m.a1[0], m.a1[1], m.hf[0], m.hf[1], b1l, ring_lane, ws, NPRE + 4);
    // ---- P1: X(1) -> a1[1] with GELU(0): a1[0] -> hf[0] (even groups ride in the blocks, odd ones run between them);
    //          a1[0] is re-armed with the bias of chunk 2
    pipe_block<NW, 48, 0, 0, 0, true, FPW, true>(m, xf, m.a1[1], m.a1[0], m.hf[0], m.hf[1], b1l + 128, ring_lane, ws, NPRE + 5);
    gelu_group_plain<true>(m, m.a1[0], b1l + 128, 1, m.hf[0]);
    pipe_block<NW, 60, 0, 1, 2, true, -1, true>(m, xf, m.a1[1], m.a1[0], m.hf[0], m.hf[1], b1l + 128, ring_lane, ws, NPRE + 5);
    gelu_group_plain<true>(m, m.a1[0], b1l + 128, 3, m.hf[0]);
    pipe_block<NW, 72, 0, 2, 4, true, FPW, true>(m, xf, m.a1[1], m.a1[0], m.hf[0], m.hf[1], b1l + 128, ring_lane, ws, NPRE + 6);
    gelu_group_plain<true>(m, m.a1[0], b1l + 128, 5, m.hf[0]);
    pipe_block<NW, 84, 0, 3, 6, true, -1, true>(m, xf, m.a1[1], m.a1[0], m.hf[0], m.hf[1], b1l + 128, ring_lane, ws, NPRE + 6);
    gelu_group_plain<true>(m, m.a1[0], b1l + 128, 7, m.hf[0]);
    ROWS_STAMP(2);
    // ---- iterations c = 1 .. 22 (two per trip: the register double buffers a1 / hf alternate)
#pragma unroll 1
    for (int c = 1; c < 23; c += 2) {
        {   // odd c: X(c + 1) -> a1[0], GELU(c): a1[1] -> hf[1] (a1[1] re-armed for chunk c + 2), Y(c - 1) <- hf[0]
            const long s0 = 4 * c;
            const float* bn = b1l + 64 * (c + 2);
            pipe_block<NW, 0, 0, 0, 0, true, FPW, true>(m, xf, m.a1[0], m.a1[1], m.hf[1], m.hf[0], bn, ring_lane, ws, NPRE + s0 + 3);
            pipe_block<NW, 12, 1, 0, 1, true, -1, true>(m, xf, m.a1[0], m.a1[1], m.hf[1], m.hf[0], bn, ring_lane, ws, NPRE + s0 + 3);
            pipe_block<NW, 24, 0, 1, 2, true, FPW, true>(m, xf, m.a1[0], m.a1[1], m.hf[1], m.hf[0], bn, ring_lane, ws, NPRE + s0 + 4);
            pipe_block<NW, 36, 1, 1, 3, true, -1, true>(m, xf, m.a1[0], m.a1[1], m.hf[1], m.hf[0], bn, ring_lane, ws, NPRE + s0 + 4);
            pipe_block<NW, 48, 0, 2, 4, true, FPW, true>(m, xf, m.a1[0], m.a1[1], m.hf[1], m.hf[0], bn, ring_lane, ws, NPRE + s0 + 5);
            pipe_block<NW, 60, 1, 2, 5, true, -1, true>(m, xf, m.a1[0], m.a1[1], m.hf[1], m.hf[0], bn, ring_lane, ws, NPRE + s0 + 5);
            pipe_block<NW, 72, 0, 3, 6, true, FPW, true>(m, xf, m.a1[0], m.a1[1], m.hf[1], m.hf[0], bn, ring_lane, ws, NPRE + s0 + 6);
            pipe_block<NW, 84, 1, 3, 7, true, -1, true>(m, xf, m.a1[0], m.a1[1], m.hf[1], m.hf[0], bn, ring_lane, ws, NPRE + s0 + 6);
        }
        {   // even c + 1: X(c + 2) -> a1[1], GELU(c + 1): a1[0] -> hf[0] (a1[0] re-armed for chunk c + 3), Y(c) <- hf[1]
            const long s0 = 4 * (c + 1);
            const float* bn = b1l + 64 * (c + 3);   // c + 3 = 24 on the last trip: slack behind the table, never consumed
            pipe_block<NW, 0, 0, 0, 0, true, FPW, true>(m, xf, m.a1[1], m.a1[0], m.hf[0], m.hf[1], bn, ring_lane, ws, NPRE + s0 + 3);
            pipe_block<NW, 12, 1, 0, 1, true, -1, true>(m, xf, m.a1[1], m.a1[0], m.hf[0], m.hf[1], bn, ring_lane, ws, NPRE + s0 + 3);
            pipe_block<NW, 24, 0, 1, 2, true, FPW, true>(m, xf, m.a1[1], m.a1[0], m.hf[0], m.hf[1], bn, ring_lane, ws, NPRE + s0 + 4);
            pipe_block<NW, 36, 1, 1, 3, true, -1, true>(m, xf, m.a1[1], m.a1[0], m.hf[0], m.hf[1], bn, ring_lane, ws, NPRE + s0 + 4);
            pipe_block<NW, 48, 0, 2, 4, true, FPW, true>(m, xf, m.a1[1], m.a1[0], m.hf[0], m.hf[1], bn, ring_lane, ws, NPRE + s0 + 5);
            pipe_block<NW, 60, 1, 2, 5, true, -1, true>(m, xf, m.a1[1], m.a1[0], m.hf[0], m.hf[1], bn, ring_lane, ws, NPRE + s0 + 5);
            pipe_block<NW, 72, 0, 3, 6, true, FPW, true>(m, xf, m.a1[1], m.a1[0], m.hf[0], m.hf[1], bn, ring_lane, ws, NPRE + s0 + 6);
            pipe_block<NW, 84, 1, 3, 7, true, -1, true>(m, xf, m.a1[1], m.a1[0], m.hf[0], m.hf[1], bn, ring_lane, ws, NPRE + s0 + 6);
        }
    }
    ROWS_STAMP(3);
    // ---- E0: Y(22) <- hf[0] with GELU(23): a1[1] -> hf[1] (slots 92, 93; slot 95 is the last one to fetch)
    pipe_block<NW, 0, 1, 0, 0, false, FPW, true>(m, xf, m.a1[0], m.a1[1], m.hf[1], m.hf[0], b1l, ring_lane, ws, NPRE + 95);
    gelu_group_plain<false>(m, m.a1[1], b1l, 1, m.hf[1]);
    pipe_block<NW, 12, 1, 1, 2, false, -1, true>(m, xf, m.a1[0], m.a1[1], m.hf[1], m.hf[0], b1l, ring_lane, ws, NPRE + 95);
    gelu_group_plain<false>(m, m.a1[1], b1l, 3, m.hf[1]);
    pipe_block<NW, 24, 1, 2, 4, false, FPW, false>(m, xf, m.a1[0], m.a1[1], m.hf[1], m.hf[0], b1l, ring_lane, ws, NPRE + 0);
    gelu_group_plain<false>(m, m.a1[1], b1l, 5, m.hf[1]);
    pipe_block<NW, 36, 1, 3, 6, false, -1, false>(m, xf, m.a1[0], m.a1[1], m.hf[1], m.hf[0], b1l, ring_lane, ws, NPRE + 0);
    gelu_group_plain<false>(m, m.a1[1], b1l, 7, m.hf[1]);
    // ---- E1: Y(23) <- hf[1] (slots 94, 95: everything has been requested; barrier 94 waits for all of it)
    pipe_block<NW, 48, 1, 0, -1, false, 0, false>(m, xf, m.a1[0], m.a1[1], m.hf[0], m.hf[1], b1l, ring_lane, ws, NPRE + 0);
    pipe_block<NW, 60, 1, 1, -1, false, -1, false>(m, xf, m.a1[0], m.a1[1], m.hf[0], m.hf[1], b1l, ring_lane, ws, NPRE + 0);
    pipe_block<NW, 72, 1, 2, -1, false, 0, false>(m, xf, m.a1[0], m.a1[1], m.hf[0], m.hf[1], b1l, ring_lane, ws, NPRE + 0);
    pipe_block<NW, 84, 1, 3, -1, false, -1, false, 12 - kWPF>(m, xf, m.a1[0], m.a1[1], m.hf[0], m.hf[1], b1l, ring_lane, ws, NPRE + 0);
    ROWS_STAMP(4);
    // ---- gated residual
    if (TAIL) {
        const float* fmod = reinterpret_cast<const float*>(smem + kRingBytes + 8192 + NW * 6144);   // [shift 512 floats | scale 512]
        float xnew[16];
        rows_final_tail(m.y, tok, fmod + 512, fmod, p.tail_w, p.tail_b, p.tail_D, p.tail_euler, p.tail_dt, p.tail_x, p.tail_out, xf, xnew);
        ROWS_STAMP(6);
        if (p.emb_base) {   // (uniform) the next step's token embedding, from the state just updated: that step launches no k_embed
            rows_embed_tail(m.y, tok, xnew, EmbedTail{p.emb_wl, p.emb_wc, p.emb_wl_hi, p.emb_wl_lo, p.emb_wc_hi, p.emb_wc_lo, p.emb_base, p.emb_mdelta, p.emb_xcond, p.emb_cmask, p.emb_T, p.emb_L, p.tail_D});
            ROWS_STAMP(7);
            rows_store<0, 12>(m.y, tok, p.h);
        }
    } else if (FOLD) {
        rows_store<0, 12>(m.y, tok, p.h);
    } else if (MODLDS) {
        rows_gate_residual_lds<0, 6>(m.y, tok, modl + 1024, p.h);
        rows_gate_residual_lds<6, 12>(m.y, tok, modl + 1024, p.h);
    } else {
        rows_gate_residual<0, 6>(m.y, tok, p.mm, p.gate_chunk, p.h);
        rows_gate_residual<6, 12>(m.y, tok, p.mm, p.gate_chunk, p.h);
    }
    ROWS_STAMP(5);
    if (p.trace && lane == 0) {
        const long i = ((long)blockIdx.x * NW + w) * 8;
        if (i + 8 <= p.trace_cap) {
#pragma unroll
            for (int k = 0; k < 8; ++k) p.trace[i + k] = st[k];
        }
    }
}

// ---- weight-stream packing ------------------------------------------------------------------------------------------
// dst fragment f (1 KiB = 64 lanes x 8 bf16) <- rows 32 tile .. + 31 of the matrix tab[f] names, K slice of k-step ks in
// kappa order (rows.h).  tab[f] = mat << 16 | tile << 8 | ks; only entries with mat == which are written.
// `kappa` = 0: natural K order k = 16 ks + 8 hh + j instead (operands whose B fragments are loaded from memory: the out-projection).
// rowmap (nullable): packed row r reads source row rowmap[r].
__global__ void k_pack_stream(const float* __restrict__ wsrc, int ld, int which, const int* __restrict__ tab, int nfrag,
                              float scale, int kappa, bf16x8* __restrict__ dst, const int* __restrict__ rowmap) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)nfrag * 64) return;
    const int lane = (int)(i & 63), f = (int)(i >> 6);
    const int e = tab[f];
    if ((e >> 16) != which) return;
    const int tile = (e >> 8) & 255, ks = e & 255;
    const int hh = lane >> 5;
    const int row = rowmap ? rowmap[tile * 32 + (lane & 31)] : tile * 32 + (lane & 31);
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int col = kappa == 1 ? 16 * ks + 8 * (j >> 2) + 4 * hh + (j & 3) : 16 * ks + 8 * hh + j;
        v[j] = (__bf16)(wsrc[(long)row * ld + col] * scale);
    }
    dst[i] = v;
}

// Gate fold (round 6): the MLP stream of (step s, layer l) with the step's gate folded into fc2, one workgroup row per (s, l):
//   fc1 fragments: copied from the layer's base stream;  fc2 fragment (feature tile, k-step): bf16(gate[row] * W2[row][kappa cols]),
//   rounded ONCE from the fp32 weight;  b2g[s][l][f] = gate[f] * b2[f].
// mod: adaLN table (row of step s at s * mod_step_stride); the layer's MLP gate chunk at goff[l] floats into the row.
struct PackFoldParams {
    const float* mod;
    long mod_step_stride;
    int nl;
    int goff[8];
    const float* w2[8];          // fp32 fc2.weight [384][1536] per layer
    const float* b2[8];
    const bf16x8* base[8];       // the layer's unfolded stream (fc1 fragments are copied from it)
    const int* tab;
    bf16x8* dst;                 // [S * nl][kMlpFrags * 64]
    float* b2g;                  // [S * nl][384]
};
__global__ __launch_bounds__(256) void k_pack_fold(const PackFoldParams p) {
    const int sl = blockIdx.y, s = sl / p.nl, l = sl % p.nl;
    const int i = blockIdx.x * 256 + threadIdx.x;            // < 2304 * 64
    const int lane = i & 63, f = i >> 6;
    const float* gate = p.mod + (long)s * p.mod_step_stride + p.goff[l];
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < kC; c += 256) p.b2g[(size_t)sl * kC + c] = gate[c] * p.b2[l][c];
    const int e = p.tab[f];
    bf16x8 v;
    if ((e >> 16) == 0) {
        v = p.base[l][i];
    } else {
        const int tile = (e >> 8) & 255, ks = e & 255, hh = lane >> 5;
        const int row = tile * 32 + (lane & 31);
        const float g = gate[row];
        const float* wr = p.w2[l] + (size_t)row * kF + 16 * ks + 4 * hh;   // kappa: cols 16 ks + 8 (j >> 2) + 4 hh + (j & 3)
        const f32x4 a = *reinterpret_cast<const f32x4*>(wr), b = *reinterpret_cast<const f32x4*>(wr + 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = (__bf16)(a[j] * g);
            v[4 + j] = (__bf16)(b[j] * g);
        }
    }
    p.dst[(size_t)sl * (2304 * 64) + i] = v;
}
void launch_pack_fold(const float* mod, long mod_step_stride, int S, int nl, const int* goff, const float* const* w2,
                      const float* const* b2, const bf16x8* const* base, const int* tab, bf16x8* dst, float* b2g, hipStream_t s) {
    PackFoldParams p{};
    p.mod = mod;
    p.mod_step_stride = mod_step_stride;
    p.nl = nl;
    for (int l = 0; l < nl; ++l) {
        p.goff[l] = goff[l];
        p.w2[l] = w2[l];
        p.b2[l] = b2[l];
        p.base[l] = base[l];
    }
    p.tab = tab;
    p.dst = dst;
    p.b2g = b2g;
    hipLaunchKernelGGL(k_pack_fold, dim3(2304 * 64 / 256, (unsigned)(S * nl)), dim3(256), 0, s, p);
}

__global__ void k_pack_embed_rows(const float* __restrict__ w, int D, float* __restrict__ pack) {
    const int i = blockIdx.x * 256 + threadIdx.x;   // < 12 * NK4 * 64
    const int nk4 = D <= 24 ? 3 : 4;
    if (i >= 12 * nk4 * 64) return;
    const int lane = i & 63, q = (i >> 6) % nk4, ft = (i >> 6) / nk4;
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int f = 8 * q + 4 * (lane >> 5) + k;
        v[k] = f < D ? w[(size_t)(32 * ft + (lane & 31)) * D + f] : 0.f;
    }
    reinterpret_cast<f32x4*>(pack)[i] = v;
}
__global__ void k_sub_f32(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ dst, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = a[i] - b[i];
}
void launch_sub_f32(const float* a, const float* b, float* dst, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_sub_f32, dim3((n + 255) / 256), dim3(256), 0, s, a, b, dst, n);
}
void launch_pack_embed_rows(const float* w, int D, float* pack, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_embed_rows, dim3(12), dim3(256), 0, s, w, D, pack);
}
__global__ __launch_bounds__(96) void k_embed_base(const float* __restrict__ bl, const float* __restrict__ bc, const float* __restrict__ mask_emb,
                                                   const float* __restrict__ pos_embed, const float* __restrict__ ipa_out, int BL, int L,
                                                   float* __restrict__ base) {
    const long row = blockIdx.x;            // s * BL + bl
    const int l = (int)(row % BL) % L, c = threadIdx.x * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(bl + c);
    const f32x4 a = *reinterpret_cast<const f32x4*>(bc + c), m0 = *reinterpret_cast<const f32x4*>(mask_emb + c),
                ip = *reinterpret_cast<const f32x4*>(ipa_out + row * kC + c);
    f32x4 pe = {0.f, 0.f, 0.f, 0.f};
    if (pos_embed) pe = *reinterpret_cast<const f32x4*>(pos_embed + (size_t)l * kC + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ((v[k] + a[k]) + m0[k]) + (pe[k] + ip[k]);
    *reinterpret_cast<f32x4*>(base + row * kC + c) = v;
}
void launch_embed_base(const float* bl, const float* bc, const float* mask_emb, const float* pos_embed, const float* ipa_out, int S,
                       int BL, int L, float* base, hipStream_t s) {
    hipLaunchKernelGGL(k_embed_base, dim3((unsigned)((long)S * BL)), dim3(96), 0, s, bl, bc, mask_emb, pos_embed, ipa_out, BL, L, base);
}

void launch_pack_stream(const float* w, int ld, int which, const int* tab, int nfrag, float scale, int kappa, bf16x8* dst,
                        hipStream_t s, const int* rowmap) {
    const long total = (long)nfrag * 64;
    hipLaunchKernelGGL(k_pack_stream, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, ld, which, tab, nfrag, scale, kappa, dst,
                       rowmap);
}

template <int NW>
static void launch_mlp_rows_nw(const MlpRowsParams& p, long tiles, hipStream_t s) {
    const bool proj = p.o != nullptr;
    // a 32-row tile never straddles two modulation groups: groups are whole tiles, or every group reads the same row
    const bool uni = p.mm.tokens_per_group % 32 == 0 || (p.mm.step_stride == 0 && p.mm.group_stride == 0);
    const dim3 g((unsigned)((tiles + NW - 1) / NW)), b(NW * 64);
    if (proj) {
        if (uni) hipLaunchKernelGGL((k_mlp_rows<NW, true, true>), g, b, 0, s, p);
        else hipLaunchKernelGGL((k_mlp_rows<NW, true, false>), g, b, 0, s, p);
    } else {
        if (p.b2g && p.tail_w) hipLaunchKernelGGL((k_mlp_rows<NW, false, true, true, true>), g, b, 0, s, p);
        else if (p.b2g) hipLaunchKernelGGL((k_mlp_rows<NW, false, true, true>), g, b, 0, s, p);   // (the caller has checked: one modulation group)
        else if (uni) hipLaunchKernelGGL((k_mlp_rows<NW, false, true>), g, b, 0, s, p);
        else hipLaunchKernelGGL((k_mlp_rows<NW, false, false>), g, b, 0, s, p);
    }
}
void launch_mlp_rows(const MlpRowsParams& p, int nw, hipStream_t s) {
    const long tiles = (p.nrows + 31) / 32;
    // (the orchestration only uses four waves; the one- and two-wave instantiations of rounds 3-5 were dead weight in the build)
    if (nw == 4) launch_mlp_rows_nw<4>(p, tiles, s);
    else g_k32_launch_error = "launch_mlp_rows: only the four-wave workgroup is built";
}

}  // namespace mdg
