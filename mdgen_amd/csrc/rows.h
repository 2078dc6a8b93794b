// rows.h -- device building blocks of the "row-owner" token-local kernels (gfx950).
//
// Design (DESIGN.md "row-owner family"): a WAVE owns 32 token rows for the whole kernel and keeps every activation of
// those rows in registers, in MFMA operand layout:
//   * all GEMMs are computed transposed, D[feature][token] = W (A operand) x X^T (B operand): a lane IS a token
//     (n = lane & 31) and holds, per 32-feature tile, the features 8a + 4hh + i (a = r >> 2, i = r & 3, hh = lane >> 5);
//   * the K dimension of every weight matrix is packed in the order
//         kappa(ks, hh, j) = 16 ks + 8 (j >> 2) + 4 hh + (j & 3)          (k-step ks, lane half hh, element j of 8)
//     which makes the accumulator registers 8s .. 8s + 7 of an output tile -- after the activation function and a
//     bf16 pack -- BE the B operand of k-step s of the next GEMM.  Activations never visit LDS, there is no
//     cross-wave exchange and hence no barrier tied to the data flow;
//   * one wave per SIMD with the whole 512-entry register file (256 arch VGPRs + 256 accumulation registers);
//   * the weights are one linear stream of 1 KiB MFMA fragments in consumption order.  The NW waves of a workgroup
//     consume the SAME stream in lock step, so it is fetched once per workgroup: LDS-DMA (global_load_lds_dwordx4,
//     no registers involved) into a ring of four 24 KiB slots, each wave issuing 1/NW of every slot; `ds_read_b128`
//     (lane-linear, conflict-free) feeds the MFMAs.  One `s_barrier` per slot (24 MFMAs per wave).
//
// Ring protocol.  Slot s (global index) lives at ring position s & 3.  Before the first MFMA of slot s every wave runs
//     s_waitcnt vmcnt(FPW) lgkmcnt(0); s_barrier            ("barrier s")
// which certifies (a) slot s + 1 has landed for every wave (each wave's own DMAs complete in order; only the FPW
// DMAs of slot s + 2 may still be in flight), (b) every wave has finished reading slot s - 1 (lgkmcnt(0): the
// look-ahead `ds_read`s run at most PF fragments ahead of the MFMAs, i.e. inside slots s and s + 1).  After barrier s
// the waves issue the DMAs of slot s + 3 into the position slot s - 1 has just vacated.  The DMAs are inline asm
// (hipcc neither counts nor drains them); the main loop contains no other vector-memory instruction, so the hand
// count is exact.
#pragma once
#include "common.h"

namespace mdg {

constexpr int kSlotFrags = 24;                       // 1 KiB weight fragments per ring slot
constexpr int kSlotBytes = kSlotFrags * 1024;
constexpr int kRingSlots = 4;
constexpr int kRingFrags = kRingSlots * kSlotFrags;  // 96: one pipeline iteration of the MLP kernel
constexpr int kRingBytes = kRingSlots * kSlotBytes;  // 96 KiB
constexpr int kWPF = 5;                              // look-ahead of the LDS -> register fragment reads (fragments)
constexpr int kWRing = kWPF + 1;                     // register ring; divides 12, so ring indices are static

typedef __attribute__((address_space(3))) unsigned char lds_u8;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(const lds_u8*)p; }

__device__ __forceinline__ float half_sum2(float x) {   // x(lane) + x(lane ^ 32)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// A zero written straight into an accumulation register (common.h opaque_zero yields a VGPR: 192 of them at once, waiting
// to be copied into the accumulator file, is exactly the register pressure the row-owner kernels cannot afford).
__device__ __forceinline__ float acc_zero() {
    float z;
    asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(z));
    return z;
}

// One LDS-DMA: 64 lanes x 16 bytes, global (uniform base + per-lane 32-bit offset + OFF) -> LDS [M0 + OFF + lane * 16]:
// the instruction offset moves BOTH ends (measured: scripts/micro/dma_offset.hip), so consecutive fragments of a slot
// need one M0 write per four DMAs.  SETM0: write M0 (= LDS destination of the group) in the same statement.  M0 is NOT
// saved / restored: hipcc reserves it but uses it for nothing in these kernels (no LDS-direct, GWS or movrel);
// build.py check_isa fails the build if it ever emits an M0 access of its own in k_rows.hip.
template <int OFF, bool SETM0>
__device__ __forceinline__ void dma_frag(const unsigned char* src, unsigned voff, unsigned lds_dst) {
    static_assert(OFF >= 0 && OFF < 4096, "13-bit signed instruction offset");
    if (SETM0)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(voff), "s"(src), "s"(lds_dst), "n"(OFF) : "memory");
    else
        asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(src), "n"(OFF) : "memory");
}

template <int VM>
__device__ __forceinline__ void ring_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VM) : "memory");
}

// NW waves share the stream; the first NPRE slots come from `pre` (a second matrix in front of the main stream: the
// out-projection ahead of the MLP), the rest from `src`.  Slot numbers are GLOBAL (prefix included); NPRE is a multiple of
// the ring size, so a slot's ring position does not depend on it.
template <int NW, int NPRE = 0>
struct WStream {
    static_assert(NPRE % kRingSlots == 0, "the prefix must cover whole ring revolutions");
    static constexpr int FPW = kSlotFrags / NW;   // DMAs per wave per slot
    const unsigned char* src;   // main stream (uniform)
    const unsigned char* pre;   // prefix stream (uniform; unused when NPRE == 0)
    unsigned ring;              // LDS byte address of the ring
    unsigned voff;              // lane * 16
    int w;                      // wave index (uniform)
    // the D-th (0 .. FPW - 1) DMA of this wave for global slot `slot`; DMAs are grouped in fours that share an M0 value
    // and a source base (the statements of a group must follow each other with no other M0 writer in between)
    template <int D>
    __device__ __forceinline__ void issue(long slot) const {
        const int f = w * FPW + (D & ~3);
        const unsigned char* base = (NPRE > 0 && slot < NPRE) ? pre + slot * kSlotBytes : src + (slot - NPRE) * kSlotBytes;
        dma_frag<(D & 3) * 1024, (D & 3) == 0>(base + f * 1024, voff, ring + ((unsigned)slot & 3u) * kSlotBytes + f * 1024);
    }
    template <int D0 = 0>
    __device__ __forceinline__ void issue_slot(long slot) const {
        issue<D0>(slot);
        if constexpr (D0 + 1 < FPW) issue_slot<D0 + 1>(slot);
    }
};

// fragment I (ring-relative index; the ring holds exactly kRingFrags consecutive fragments of the stream)
template <int I>
__device__ __forceinline__ bf16x8 ring_frag(const unsigned char* ring_lane) {
    return *reinterpret_cast<const bf16x8*>(ring_lane + (I % kRingFrags) * 1024);
}

// ---- LayerNorm + adaLN modulate of the wave's 32 rows, straight into B-operand fragments -------------------------
// y = LN(x) * (1 + scale) + shift (layers.py:14-15; no affine, eps) for token `tok` (-1: padding row -> zeros).
// Lane (n, hh) reads the 192 features kappa(ks, hh, .) of its row as 48 unconditional 16-byte loads (all in flight
// together: 48 KiB per wave), reduces them locally and with its partner lane n + 32 (one permlane32 swap per
// statistic), and packs xf[ks] = the B operand of k-step ks.  The modulation vectors are read through per-lane row
// offsets, so a tile may straddle samples.
// rows_load: v[2 ks + q][0..3] = features 16 ks + 8 q + 4 hh + (0..3) of the lane's row (the register image the epilogues
// produce as well: accumulator register 4 a + i of tile ft is v[4 ft + a][i]).  rows_norm: statistics + modulate + pack.
__device__ __forceinline__ void rows_load(const float* __restrict__ x, int tok, f32x4 (&v)[48]) {
    const int hh = lane_id() >> 5;
    const unsigned tokc = tok < 0 ? 0u : (unsigned)tok;
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(x);
    const unsigned off = tokc * (unsigned)(kC * 4) + (unsigned)hh * 16u;
#ifdef MDGEN_DEV_ROWS_COALESCED   // (experiment build: the same 48 KiB per wave read as 48 fully coalesced 1 KiB requests -- WRONG values,
                                  // timing only: does the 32-byte-per-row request pattern cost HBM efficiency?)
    const unsigned offc = (tokc & ~31u) * (unsigned)(kC * 4) + (unsigned)lane_id() * 16u;
#pragma unroll
    for (int i = 0; i < 48; ++i) v[i] = *reinterpret_cast<const f32x4*>(xb + offc + 1024u * i);
#else
#pragma unroll
    for (int i = 0; i < 48; ++i) v[i] = *reinterpret_cast<const f32x4*>(xb + off + 32u * i);
#endif
}
__device__ __forceinline__ void rows_norm(const f32x4 (&v)[48], int tok, const ModMap mm, int shift_chunk, int scale_chunk,
                                          float eps, bf16x8 (&xf)[24]) {
    const int hh = lane_id() >> 5;
    const unsigned tokc = tok < 0 ? 0u : (unsigned)tok;
    const unsigned mo = tok < 0 ? 0u : (unsigned)mm.row_off(tokc);
    const unsigned char* mb = reinterpret_cast<const unsigned char*>(mm.mod);
    const unsigned osc = (mo + (unsigned)(scale_chunk * kC)) * 4u + (unsigned)hh * 16u;
    const unsigned osh = (mo + (unsigned)(shift_chunk * kC)) * 4u + (unsigned)hh * 16u;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 48; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = half_sum2(s) * (1.0f / kC);
    // (the loaded tuples are never modified: an in-place `v -= mean` makes hipcc rename 192 registers tuple by tuple and
    // spill the originals; the centred values are recomputed where they are used)
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 48; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = v[i][j] - mean;
            q += d * d;
        }
    }
    const float rstd = tok < 0 ? 0.f : 1.0f / sqrtf(half_sum2(q) * (1.0f / kC) + eps);
    const float live = tok < 0 ? 0.f : 1.f;   // padding rows enter every GEMM as zeros
    float mean2 = mean;
    asm volatile("" : "+v"(mean2));   // opaque copy: keeps hipcc from holding on to the 192 centred values of the variance pass
    // modulate + pack, k-step by k-step; the (L2-resident) modulation vectors are requested kModPF k-steps ahead and the
    // loop is fenced so that hipcc neither sinks those loads to their use nor hoists all 96 of them (384 registers)
    constexpr int kModPF = 2;
    f32x4 sc[kModPF + 1][2], sh[kModPF + 1][2];
#pragma unroll
    for (int ks = 0; ks < kModPF; ++ks)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            sc[ks][h2] = *reinterpret_cast<const f32x4*>(mb + osc + 32u * (2 * ks + h2));
            sh[ks][h2] = *reinterpret_cast<const f32x4*>(mb + osh + 32u * (2 * ks + h2));
        }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 24; ++ks) {
        if (ks + kModPF < 24) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                sc[(ks + kModPF) % (kModPF + 1)][h2] = *reinterpret_cast<const f32x4*>(mb + osc + 32u * (2 * (ks + kModPF) + h2));
                sh[(ks + kModPF) % (kModPF + 1)][h2] = *reinterpret_cast<const f32x4*>(mb + osh + 32u * (2 * (ks + kModPF) + h2));
            }
        }
        uint32_t u[4];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const f32x4 c = sc[ks % (kModPF + 1)][h2], d = sh[ks % (kModPF + 1)][h2];
            const f32x4 a = v[2 * ks + h2];
            const float y0 = (a[0] - mean2) * (rstd * c[0] + rstd) + live * d[0];
            const float y1 = (a[1] - mean2) * (rstd * c[1] + rstd) + live * d[1];
            const float y2 = (a[2] - mean2) * (rstd * c[2] + rstd) + live * d[2];
            const float y3 = (a[3] - mean2) * (rstd * c[3] + rstd) + live * d[3];
            u[2 * h2] = pack_bf16(y0, y1);
            u[2 * h2 + 1] = pack_bf16(y2, y3);
        }
        u32x4 t = {u[0], u[1], u[2], u[3]};
        asm volatile("" : "+v"(t));   // pinned: hipcc otherwise sinks the normalisation to each fragment's first use inside the
                                      // following GEMM and keeps the raw row image alive (spilled) until then
        xf[ks] = __builtin_bit_cast(bf16x8, t);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The same with the modulation vectors in LDS (`sc`, `sh`: the wave's scale / shift chunk, 384 floats each; every row of the
// wave shares them): 16-byte broadcast reads instead of 96 L2 round trips per lane (~12k cycles of a 28k-cycle prologue).
__device__ __forceinline__ void rows_norm_lds(const f32x4 (&v)[48], int tok, const float* sc, const float* sh, float eps,
                                              bf16x8 (&xf)[24]) {
    const int hh = lane_id() >> 5;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 48; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = half_sum2(s) * (1.0f / kC);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 48; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = v[i][j] - mean;
            q += d * d;
        }
    }
    const float rstd = tok < 0 ? 0.f : 1.0f / sqrtf(half_sum2(q) * (1.0f / kC) + eps);
    const float live = tok < 0 ? 0.f : 1.f;
    float mean2 = mean;
    asm volatile("" : "+v"(mean2));   // (see rows_norm)
    const f32x4* scp = reinterpret_cast<const f32x4*>(sc + 4 * hh);
    const f32x4* shp = reinterpret_cast<const f32x4*>(sh + 4 * hh);
#pragma unroll
    for (int ks = 0; ks < 24; ++ks) {
        uint32_t u[4];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const f32x4 c = scp[2 * (2 * ks + h2)], d = shp[2 * (2 * ks + h2)];   // features 16 ks + 8 h2 + 4 hh ..
            const f32x4 a = v[2 * ks + h2];
            const float y0 = (a[0] - mean2) * (rstd * c[0] + rstd) + live * d[0];
            const float y1 = (a[1] - mean2) * (rstd * c[1] + rstd) + live * d[1];
            const float y2 = (a[2] - mean2) * (rstd * c[2] + rstd) + live * d[2];
            const float y3 = (a[3] - mean2) * (rstd * c[3] + rstd) + live * d[3];
            u[2 * h2] = pack_bf16(y0, y1);
            u[2 * h2 + 1] = pack_bf16(y2, y3);
        }
        xf[ks] = __builtin_bit_cast(bf16x8, u32x4{u[0], u[1], u[2], u[3]});
    }
}

// rows_norm_lds that ALSO starts the fc2 accumulators from the residual rows (round 6, "gate fold"): sampling shares t across the
// batch (integrators.py:99), so the MLP's gate is one vector per (layer, step) and is folded into the weights once per call --
// W2' = diag(gate) W2, b2' = gate * b2 (k_pack_fold) -- which turns  h + gate * (W2 u + b2)  (latent_model.py:481) into
// y0 = h + b2',  y += W2' u:  the accumulator registers 4 a + j of tile ft start as v[4 ft + a][j] + b2'[..] (the row image IS
// the accumulator image, see rows_load) and the epilogue is a plain store -- the rows are read from HBM once, not twice.
// `b2g`: b2' in LDS (384 floats, shared by the wave's rows).
__device__ __forceinline__ void rows_norm_lds_fold(const f32x4 (&v)[48], int tok, const float* sc, const float* sh, const float* b2g,
                                                   float eps, bf16x8 (&xf)[24], f32x16 (&y)[12]) {
    const int hh = lane_id() >> 5;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 48; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = half_sum2(s) * (1.0f / kC);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 48; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = v[i][j] - mean;
            q += d * d;
        }
    }
    const float rstd = tok < 0 ? 0.f : 1.0f / sqrtf(half_sum2(q) * (1.0f / kC) + eps);
    const float live = tok < 0 ? 0.f : 1.f;
    float mean2 = mean;
    asm volatile("" : "+v"(mean2));   // (see rows_norm)
    const f32x4* scp = reinterpret_cast<const f32x4*>(sc + 4 * hh);
    const f32x4* shp = reinterpret_cast<const f32x4*>(sh + 4 * hh);
    const f32x4* bgp = reinterpret_cast<const f32x4*>(b2g + 4 * hh);
#pragma unroll
    for (int ks = 0; ks < 24; ++ks) {
        uint32_t u[4];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int i = 2 * ks + h2;
            const f32x4 c = scp[2 * i], d = shp[2 * i], b = bgp[2 * i];   // features 16 ks + 8 h2 + 4 hh ..
            const f32x4 a = v[i];
            const float y0 = (a[0] - mean2) * (rstd * c[0] + rstd) + live * d[0];
            const float y1 = (a[1] - mean2) * (rstd * c[1] + rstd) + live * d[1];
            const float y2 = (a[2] - mean2) * (rstd * c[2] + rstd) + live * d[2];
            const float y3 = (a[3] - mean2) * (rstd * c[3] + rstd) + live * d[3];
            u[2 * h2] = pack_bf16(y0, y1);
            u[2 * h2 + 1] = pack_bf16(y2, y3);
#pragma unroll
            for (int j = 0; j < 4; ++j) y[i >> 2][4 * (i & 3) + j] = a[j] + b[j];
        }
        xf[ks] = __builtin_bit_cast(bf16x8, u32x4{u[0], u[1], u[2], u[3]});
    }
}

// ... and its epilogue: h[tok][32 ft + 8 a + 4 hh + j] = y[ft][4 a + j], stores only
template <int FT0, int FT1>
__device__ __forceinline__ void rows_store(const f32x16 (&y)[12], int tok, float* __restrict__ h) {
    const int hh = lane_id() >> 5;
    const unsigned tokc = tok < 0 ? 0u : (unsigned)tok;
    unsigned char* hb = reinterpret_cast<unsigned char*>(h);
    const unsigned off = tokc * (unsigned)(kC * 4) + (unsigned)hh * 16u;
    if (tok >= 0) {
#pragma unroll
        for (int i = FT0 * 4; i < FT1 * 4; ++i) {
            const int ft = i >> 2, a = i & 3;
            const f32x4 o = {y[ft][4 * a], y[ft][4 * a + 1], y[ft][4 * a + 2], y[ft][4 * a + 3]};
            *reinterpret_cast<f32x4*>(hb + off + 32u * (unsigned)i) = o;
        }
    }
}

// FinalLayer on the wave's 32 rows, from the fc2 accumulators (k_mlp_rows<., TAIL>): v = W (LN(h)(1 + scale) + shift) + b,
// x += dt v (euler) or out = v  (layers.py:70-74; integrators.py:106).  `y` holds the updated residual rows (accumulator image =
// row image); sc / sh: the final layer's scale / shift chunks in LDS; wfin: 24 fragments (one 32-row tile, rows >= D zero, kappa K
// order); bfin [32].  D[feature][token]: lane (n, hh) ends with features 8 a + 4 hh + i of its token in register 4 a + i.
__device__ __forceinline__ void rows_final_tail(const f32x16 (&y)[12], int tok, const float* sc, const float* sh,
                                                const bf16x8* __restrict__ wfin, const float* __restrict__ bfin, int D, int euler,
                                                float dt, float* __restrict__ x, float* __restrict__ out, bf16x8 (&xf)[24],
                                                float (&xnew)[16]) {   // xnew[4 a + i]: the updated state value of feature 8 a + 4 hh + i
    const int lane = lane_id(), hh = lane >> 5;
    const unsigned tokc = tok < 0 ? 0u : (unsigned)tok;
    // the state values and the bias are requested first; the weight tile (24 KiB, L2) once the row image is dead (192 + 96 registers
    // do not fit beside it)
    f32x4 b4[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) b4[a] = *reinterpret_cast<const f32x4*>(bfin + 8 * a + 4 * hh);
    float xv[16];
    float* xp = (euler ? x : out) + (size_t)tokc * (unsigned)D;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = 8 * (r >> 2) + 4 * hh + (r & 3);
        xv[r] = euler ? xp[f < D ? f : 0] : 0.f;   // (unconditional, clamped)
    }
    f32x4 v[48];
#pragma unroll
    for (int i = 0; i < 48; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[i][j]) : "a"(y[i >> 2][4 * (i & 3) + j]));
    rows_norm_lds(v, tok, sc, sh, 1e-6f, xf);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 wf[24];
#pragma unroll
    for (int ks = 0; ks < 24; ++ks) wf[ks] = wfin[ks * 64 + lane];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = b4[r >> 2][r & 3];
#pragma unroll
    for (int ks = 0; ks < 24; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], xf[ks], acc, 0, 0, 0);
    const float s = euler ? dt : 1.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = 8 * (r >> 2) + 4 * hh + (r & 3);
        xnew[r] = f < D ? xv[r] + s * acc[r] : 0.f;
        if (tok >= 0 && f < D) xp[f] = xnew[r];
    }
}

// The NEXT step's token embedding from the state the tail has just updated (k_embed's work, latent_model.py:233-246):
//   h0[tok] = W_l x_new + [W_c x_cond] + (b_l + b_c + mask_to_emb[0] + pos_embed[l] + ipa_out[step + 1][b, l])  + [mask delta]
// computed transposed on the fp32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32 products as k_embed): A = the weight (lane m = output
// feature of the tile, k = hh), B = the state -- and the B operand of k-step j IS xnew[j]: the weights are packed with their K
// dimension in the order the FinalLayer's accumulators hold the features (api.hip launch_pack_embed_rows).  The accumulators
// start from the per-(step, b, l) base row (k_embed_base, once per call) loaded as a row image; result stored as rows of h.
// wl / wc: [12 ft][NK4][64 lanes][4] floats, NK4 = 3 (D <= 24) or 4; base: rows of 384 floats, row = b * L + l of the launch's view.
// The same product on the bf16 MFMA with both operands split into a bf16 pair hi + lo (16 mantissa bits each side; the lo x lo
// term, 2^-18 of the product, is dropped): 3 x 2 v_mfma_f32_32x32x16_bf16 per feature tile instead of 12 v_mfma_f32_32x32x2_f32 --
// the fp32 MFMA runs at a quarter of its nominal rate on gfx950 (profiles/r06_experiments.txt #5: the 144 of them were 30 us of this
// launch).  whi / wlo: [12 ft][2 k-steps][64 lanes] bf16x8, K (padded to 32) in kappa order: k-step s of a lane half IS xnew[8 s .. 8 s + 7].
__device__ __forceinline__ void rows_embed_gemm_split(f32x16 (&y)[12], const bf16x8* __restrict__ whi, const bf16x8* __restrict__ wlo,
                                                      const float (&xv)[16]) {
    const int lane = lane_id();
    bf16x8 xh[2], xl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const __bf16 h = (__bf16)xv[8 * s + j];
            xh[s][j] = h;
            xl[s][j] = (__bf16)(xv[8 * s + j] - (float)h);
        }
    bf16x8 wh[2][2], wl[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        wh[0][s] = whi[s * 64 + lane];
        wl[0][s] = wlo[s * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ft = 0; ft < 12; ++ft) {
        if (ft + 1 < 12) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                wh[(ft + 1) & 1][s] = whi[((ft + 1) * 2 + s) * 64 + lane];
                wl[(ft + 1) & 1][s] = wlo[((ft + 1) * 2 + s) * 64 + lane];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            y[ft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[ft & 1][s], xh[s], y[ft], 0, 0, 0);   // small terms first
            y[ft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[ft & 1][s], xl[s], y[ft], 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) y[ft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[ft & 1][s], xh[s], y[ft], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int NK4>
__device__ __forceinline__ void rows_embed_gemm(f32x16 (&y)[12], const float* __restrict__ wrows, const float (&xv)[16]) {
    const int lane = lane_id();
    const f32x4* wp = reinterpret_cast<const f32x4*>(wrows) + lane;
    f32x4 w[2][NK4];
#pragma unroll
    for (int q = 0; q < NK4; ++q) w[0][q] = wp[q * 64];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ft = 0; ft < 12; ++ft) {
        if (ft + 1 < 12) {
#pragma unroll
            for (int q = 0; q < NK4; ++q) w[(ft + 1) & 1][q] = wp[((ft + 1) * NK4 + q) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NK4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) y[ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ft & 1][q][i], xv[4 * q + i], y[ft], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}
struct EmbedTail {
    const float *wl, *wc;        // packed latent_to_emb / cond_to_emb weights (rows order, fp32: the exact form)
    const bf16x8 *wl_hi, *wl_lo, *wc_hi, *wc_lo;   // ... as bf16 pairs for rows_embed_gemm_split (null: the exact form runs)
    const float* base;           // base rows of the NEXT step for this launch's view: [B_view * L][384]
    const float* mdelta;         // mask_to_emb[1] - mask_to_emb[0]  [384]
    const float* x_cond;         // [N][D]
    const int64_t* x_cond_mask;  // [N]
    int T, L, D;
};
// (computes into y; the caller stores the rows: rows_store)
__device__ __forceinline__ void rows_embed_tail(f32x16 (&y)[12], int tok, const float (&xnew)[16], const EmbedTail e) {
    const int lane = lane_id(), hh = lane >> 5;
    const unsigned tokc = tok < 0 ? 0u : (unsigned)tok;
    const unsigned row = (tokc / (unsigned)(e.T * e.L)) * (unsigned)e.L + tokc % (unsigned)e.L;
    // everything requested up front: base row image, the conditioning values and mask of the token
    const unsigned char* bb = reinterpret_cast<const unsigned char*>(e.base) + (size_t)row * (kC * 4) + hh * 16;
#pragma unroll
    for (int i = 0; i < 48; ++i) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bb + 32u * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) y[i >> 2][4 * (i & 3) + j] = b[j];
    }
    float xc[16];
    const float* cp = e.x_cond + (size_t)tokc * (unsigned)e.D;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = 8 * (r >> 2) + 4 * hh + (r & 3);
        const float v = cp[f < e.D ? f : 0];
        xc[r] = f < e.D && tok >= 0 ? v : 0.f;
    }
    const bool cm = tok >= 0 && e.x_cond_mask[tokc] != 0;
    if (e.wl_hi) rows_embed_gemm_split(y, e.wl_hi, e.wl_lo, xnew);
    else if (e.D <= 24) rows_embed_gemm<3>(y, e.wl, xnew);
    else rows_embed_gemm<4>(y, e.wl, xnew);
    bool anyc = false;
#pragma unroll
    for (int r = 0; r < 16; ++r) anyc |= xc[r] != 0.f;
    if (__builtin_amdgcn_ballot_w64(anyc) != 0) {   // some token of the tile is conditioned (k_embed skips the product likewise)
        if (e.wc_hi) rows_embed_gemm_split(y, e.wc_hi, e.wc_lo, xc);
        else if (e.D <= 24) rows_embed_gemm<3>(y, e.wc, xc);
        else rows_embed_gemm<4>(y, e.wc, xc);
    }
    if (__builtin_amdgcn_ballot_w64(cm) != 0) {     // mask_to_emb[1] instead of [0] for the tokens whose x_cond_mask is set
        const unsigned char* mb = reinterpret_cast<const unsigned char*>(e.mdelta) + hh * 16;
        const float sel = cm ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < 48; ++i) {
            const f32x4 d = *reinterpret_cast<const f32x4*>(mb + 32u * i);
#pragma unroll
            for (int j = 0; j < 4; ++j) y[i >> 2][4 * (i & 3) + j] += sel * d[j];
        }
    }
}

__device__ __forceinline__ void rows_ln(const float* __restrict__ x, int tok, const ModMap mm, int shift_chunk,
                                        int scale_chunk, float eps, bf16x8 (&xf)[24]) {
    f32x4 v[48];
    rows_load(x, tok, v);
    rows_norm(v, tok, mm, shift_chunk, scale_chunk, eps, xf);
}

// ---- accumulators that START from the bias: y[ft][4 a + j] = bias[32 ft + 8 a + 4 hh + j] -----------------------------------
// (48 16-byte loads of an L2-resident vector, once per GEMM: the epilogue then needs neither the bias loads nor the adds)
__device__ __forceinline__ void rows_acc_init(f32x16 (&y)[12], const float* __restrict__ bias) {
    const unsigned char* bb = reinterpret_cast<const unsigned char*>(bias) + (lane_id() >> 5) * 16;
#pragma unroll
    for (int ft = 0; ft < 12; ++ft)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bb + 32u * (4 * ft + a));
#pragma unroll
            for (int j = 0; j < 4; ++j) y[ft][4 * a + j] = b[j];
        }
}

// ---- gated residual epilogue of the wave's 32 rows ----------------------------------------------------------------
// h[tok][32 ft + 8 a + 4 hh + i] += gate * y[ft][4 a + i]   (latent_model.py:462,476,481; y already holds the bias): 16-byte
// accesses, the same address pattern as rows_load; loads unconditional, stores predicated on the row being real.
template <int FT0, int FT1>
__device__ __forceinline__ void rows_gate_residual(const f32x16 (&y)[12], int tok, const ModMap mm, int gate_chunk,
                                                   float* __restrict__ h) {
    const int hh = lane_id() >> 5;
    const unsigned tokc = tok < 0 ? 0u : (unsigned)tok;
    unsigned char* hb = reinterpret_cast<unsigned char*>(h);
    const unsigned off = tokc * (unsigned)(kC * 4) + (unsigned)hh * 16u;
    const unsigned mo = tok < 0 ? 0u : (unsigned)mm.row_off(tokc);
    const unsigned char* mb = reinterpret_cast<const unsigned char*>(mm.mod);
    const unsigned og = (mo + (unsigned)(gate_chunk * kC)) * 4u + (unsigned)hh * 16u;
    constexpr int NV = (FT1 - FT0) * 4;
    f32x4 hv[NV], g[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const unsigned fo = 32u * (unsigned)(FT0 * 4 + i);   // byte offset of feature 32 ft + 8 a (+ 4 hh via off / og)
        hv[i] = *reinterpret_cast<const f32x4*>(hb + off + fo);
        g[i] = *reinterpret_cast<const f32x4*>(mb + og + fo);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int ft = FT0 + (i >> 2), a = i & 3;
        f32x4 o = hv[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += g[i][j] * y[ft][4 * a + j];
        if (tok >= 0) *reinterpret_cast<f32x4*>(hb + off + 32u * (unsigned)(FT0 * 4 + i)) = o;
    }
}

// ... with the gate chunk in LDS (`gate`: 384 floats shared by every row of the wave)
template <int FT0, int FT1>
__device__ __forceinline__ void rows_gate_residual_lds(const f32x16 (&y)[12], int tok, const float* gate, float* __restrict__ h) {
    const int hh = lane_id() >> 5;
    const unsigned tokc = tok < 0 ? 0u : (unsigned)tok;
    unsigned char* hb = reinterpret_cast<unsigned char*>(h);
    const unsigned off = tokc * (unsigned)(kC * 4) + (unsigned)hh * 16u;
    const f32x4* gp = reinterpret_cast<const f32x4*>(gate + 4 * hh);
    constexpr int NV = (FT1 - FT0) * 4;
    f32x4 hv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) hv[i] = *reinterpret_cast<const f32x4*>(hb + off + 32u * (unsigned)(FT0 * 4 + i));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int ft = FT0 + (i >> 2), a = i & 3;
        const f32x4 g = gp[2 * (FT0 * 4 + i)];
        f32x4 o = hv[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += g[j] * y[ft][4 * a + j];
        if (tok >= 0) *reinterpret_cast<f32x4*>(hb + off + 32u * (unsigned)(FT0 * 4 + i)) = o;
    }
}

// The same update, keeping the updated rows in registers (rows_load's image) for the LayerNorm of the NEXT sub-layer: the
// residual stream is written (the next epilogue needs it back) but not read again by the next prologue.  The gate vectors
// are requested one feature tile ahead (fenced: left alone hipcc either serialises a round trip per tile or hoists all 48).
__device__ __forceinline__ void rows_gate_residual_keep(const f32x16 (&y)[12], int tok, const ModMap mm, int gate_chunk,
                                                        float* __restrict__ h, f32x4 (&v)[48]) {
    const int hh = lane_id() >> 5;
    const unsigned tokc = tok < 0 ? 0u : (unsigned)tok;
    unsigned char* hb = reinterpret_cast<unsigned char*>(h);
    const unsigned off = tokc * (unsigned)(kC * 4) + (unsigned)hh * 16u;
    const unsigned mo = tok < 0 ? 0u : (unsigned)mm.row_off(tokc);
    const unsigned char* mb = reinterpret_cast<const unsigned char*>(mm.mod);
    const unsigned og = (mo + (unsigned)(gate_chunk * kC)) * 4u + (unsigned)hh * 16u;
    f32x4 g[2][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) g[0][a] = *reinterpret_cast<const f32x4*>(mb + og + 32u * a);
#pragma unroll
    for (int i = 0; i < 48; ++i) v[i] = *reinterpret_cast<const f32x4*>(hb + off + 32u * i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ft = 0; ft < 12; ++ft) {
        if (ft + 1 < 12) {
#pragma unroll
            for (int a = 0; a < 4; ++a) g[(ft + 1) & 1][a] = *reinterpret_cast<const f32x4*>(mb + og + 32u * (4 * (ft + 1) + a));
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            f32x4 o = v[4 * ft + a];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += g[ft & 1][a][j] * y[ft][4 * a + j];
            v[4 * ft + a] = o;
            if (tok >= 0) *reinterpret_cast<f32x4*>(hb + off + 32u * (4 * ft + a)) = o;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// bf16 rows [token][384] (the attention kernel's output) -> B-operand fragments in NATURAL k order: lane (n, hh), k-step ks
// = the 16 bytes at feature 16 ks + 8 hh of its row.  24 unconditional 16-byte loads.
__device__ __forceinline__ void rows_load_bf16(const __bf16* __restrict__ o, int tok, bf16x8 (&xf)[24]) {
    const int hh = lane_id() >> 5;
    const unsigned tokc = tok < 0 ? 0u : (unsigned)tok;
    const unsigned char* ob = reinterpret_cast<const unsigned char*>(o);
    const unsigned off = tokc * (unsigned)(kC * 2) + (unsigned)hh * 16u;
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int ks = 0; ks < 24; ++ks) {
        const u32x4 t = *reinterpret_cast<const u32x4*>(ob + off + 32u * ks);
        xf[ks] = __builtin_bit_cast(bf16x8, tok >= 0 ? t : z);
    }
}

}  // namespace mdg
