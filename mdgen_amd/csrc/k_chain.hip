// k_chain.hip -- the token-local chain between two temporal attention kernels of a tetrapeptide trunk layer, as ONE
// row-owner kernel (rows.h):
//     residue-axis attention sub-layer (L == 4):  h += gate_l * Wo_l attn(LN(h)...)          latent_model.py:457-462, mha.py:258-397
//     temporal-axis front half:                   LN -> modulate -> q, k, v -> RoPE -> attention operand fragments
//                                                                                                latent_model.py:465-475, mha.py:258-268, 356-357
// It replaces k_ln_qkv_attn4<true> + k_ln_qkv<false> (two launches, the residual stream crossing HBM three times) for
// launches that fill the chip: the rows of a wave are read once and written once, q / k / v / o of the residue axis never
// leave the registers, and the 167 MB of temporal fragments are stored from the epilogues of the GEMM stages that produce them.
//
// A wave owns 32 consecutive tokens = 8 frames x 4 residues of one sample; lane n <-> frame 4 ((n >> 2) & 1) + (n >> 3),
// residue n & 3, so that (a) the four residues of a frame are a DPP quad (the 5-key attention is quad-local, as in
// k_ln_qkv_attn4) and (b) in a NON-transposed product D[token][feature] the accumulator registers 4a + i of lane half hh are
// the frames 4 hh + a of residue i: four consecutive key slots of ONE temporal sequence, i.e. an aligned 8-byte piece of its
// V^T fragment -- no cross-lane movement anywhere.
//
// Every GEMM runs as "stages" of 3 output tiles x 24 k-steps (72 weight fragments = 3 ring slots, streamed by LDS-DMA as in
// k_mlp_rows; fragment order k-step major, tile minor): 12 stages q | k | v of the residue axis (head group by head group),
// 4 stages of its out-projection, 12 stages q, k, v of the temporal axis = 84 slots.  Phase 1 of this kernel: epilogues run
// between the stages (not yet interleaved with the next stage's MFMAs).
//
// vmcnt discipline.  The ring barrier waits `vmcnt(6)`: at most this wave's 6 LDS-DMAs of slot s + 2 may be in flight.  Loads
// return in order, so the bound is safe whatever else is outstanding: global STORES issued by the epilogues can only make it
// over-wait.  Ordinary global LOADS inside the streaming part (the re-read of h for the residual, LayerNorm's modulation
// vectors) make hipcc wait for them with a count that ignores the DMAs -- a drain of the ring's prefetch depth, one L2 round
// trip each; they are kept to four places per kernel.  Tables an epilogue needs per stage live in LDS.
#include "kernels.h"
#include "rows.h"

namespace mdg {

constexpr int kChSlotsL = 36, kChSlotsO = 12, kChSlotsT = 36, kChSlots = kChSlotsL + kChSlotsO + kChSlotsT;
// LDS tables behind the ring (float offsets)
constexpr int TB_BQL = 0, TB_BKL = 384, TB_BVL = 768, TB_BOL = 1152, TB_BQT = 1536, TB_BKT = 1920, TB_BVT = 2304,
              TB_KBL = 2688, TB_BVLR = 3072, TB_ROPEL = 3456, TB_GATE = 3616, TB_ROPET = TB_GATE + 4 * 384,
              TB_END = TB_ROPET + 4 * 8 * 36;
constexpr int kChTabBytes = TB_END * 4;
constexpr int kChStashBytes = 4 * 24 * 64 * 4;
constexpr int kChSmemBytes = kRingBytes + kChTabBytes + kChStashBytes;
static_assert(kChSmemBytes <= 160 * 1024, "LDS budget");

// three-segment weight stream (slots are global): [0, 36) residue q|k|v, [36, 48) residue out-projection, [48, 84) temporal
// q, k, v; slots >= 84 (the ring's look-ahead past the end) re-fetch the last segment's head: never consumed.
struct ChStream {
    const unsigned char *sl, *so, *st;
    unsigned ring, voff;
    int w;
    __device__ __forceinline__ const unsigned char* slot_src(int slot) const {
        return slot < kChSlotsL ? sl + (long)slot * kSlotBytes
             : slot < kChSlotsL + kChSlotsO ? so + (long)(slot - kChSlotsL) * kSlotBytes
             : slot < kChSlots ? st + (long)(slot - kChSlotsL - kChSlotsO) * kSlotBytes
             : st;
    }
    __device__ __forceinline__ void issue_slot(int slot) const {   // this wave's 6 DMAs (two M0 groups: 4 + 2)
        const unsigned char* base = slot_src(slot) + w * 6 * 1024;
        const unsigned dst = ring + ((unsigned)slot & 3u) * kSlotBytes + w * 6 * 1024;
        dma_frag<0, true>(base, voff, dst);
        dma_frag<1024, false>(base, voff, dst);
        dma_frag<2048, false>(base, voff, dst);
        dma_frag<3072, false>(base, voff, dst);
        dma_frag<0, true>(base + 4096, voff, dst + 4096);
        dma_frag<1024, false>(base + 4096, voff, dst + 4096);
    }
};

// The slices of an epilogue that ride beside the MFMAs of the next stage are pure VALU work: left alone, hipcc's instruction
// selection orders them by their USES (everything lands in front of the stores at the end) whatever the source order says.
// An empty volatile asm on a slice's results pins it where it is written.
#define CH_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define CH_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

struct ChRing {
    const unsigned char* lane_base;   // smem + lane * 16
    int slot;                         // next slot to consume
    bf16x8 wr[6];                     // wr[i] = fragment i of that slot for i < 5 (look-ahead of five fragments)
};

// One stage: 3 output tiles x 24 k-steps.  SWAP == false: D[feature][token] = W (A) x X^T (B) (transposed product: a lane is
// a token); SWAP == true: D[token][feature] = X (A) x W^T (B) (a lane is a feature).  `fill(s)`, s = 0 .. 71, is called behind
// MFMA s in the same fenced scheduling region: a slice of the PREVIOUS stage's epilogue (which reads the other accumulator
// buffer) -- a lone wave has nobody else to fill its matrix pipe's shadow with (rows.h, k_mlp_rows' GELU stages).
struct ChNoFill {
    __device__ __forceinline__ void operator()(int) const {}
};
template <bool SWAP, class Fill>
__device__ __forceinline__ void ch_stage(ChRing& r, const ChStream& ws, f32x16 (&acc)[3], const bf16x8 (&xf)[24], Fill& fill) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        ring_barrier<6>();            // slot r.slot + 1 has landed for everybody, everybody has left slot r.slot - 1
        ws.issue_slot(r.slot + 3);    // ... whose ring position is refilled
        const unsigned char* cur = r.lane_base + (r.slot & 3) * kSlotBytes;
        const unsigned char* nxt = r.lane_base + ((r.slot + 1) & 3) * kSlotBytes;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            const int idx = q + 5;
            r.wr[idx % 6] = *reinterpret_cast<const bf16x8*>(idx < 24 ? cur + idx * 1024 : nxt + (idx - 24) * 1024);
            const int f = 24 * j + q, ks = f / 3, tile = f % 3;
            if (SWAP) acc[tile] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[ks], r.wr[q % 6], acc[tile], 0, 0, 0);
            else acc[tile] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.wr[q % 6], xf[ks], acc[tile], 0, 0, 0);
            fill(f);
            __builtin_amdgcn_sched_barrier(0);
        }
        r.slot += 1;
    }
}
template <bool SWAP>
__device__ __forceinline__ void ch_stage(ChRing& r, const ChStream& ws, f32x16 (&acc)[3], const bf16x8 (&xf)[24]) {
    ChNoFill nf;
    ch_stage<SWAP>(r, ws, acc, xf, nf);
}

__device__ __forceinline__ void ch_zero(f32x16 (&acc)[3]) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = acc_zero();
}

template <int J>
__device__ __forceinline__ float ch_quad(float v) {   // value of lane (quad base + J) in every lane of the quad
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), J * 0x55, 0xf, 0xf, true));
}

// the 12 values lane half hh holds for head hd (0..3 of the stage's group) of its token, + bias
template <int HD>
__device__ __forceinline__ void ch_head12(const f32x16 (&acc)[3], const f32x4* bq3, float (&e)[12]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int ap = 3 * HD + c, ft = ap >> 2, a = ap & 3;
        const f32x4 b = bq3[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[4 * c + j] = acc[ft][4 * a + j] + b[j];
    }
}

__device__ __forceinline__ void ch_rope12(float (&e)[12], const f32x4 (&rq)[4]) {
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        const float cs = rq[p >> 2][p & 3], sn = rq[2 + (p >> 2)][p & 3];
        const float x1 = e[2 * p], x2 = e[2 * p + 1];
        e[2 * p] = x1 * cs - x2 * sn;
        e[2 * p + 1] = x2 * cs + x1 * sn;
    }
}

// ---- residue axis (L = 4): the attention of one head group as three step machines, each riding beside the MFMAs of the
// stage that follows the one whose accumulators it reads -------------------------------------------------------------------
//   QE (q stage's epilogue, beside the k stage): per head 13 steps: bias request | 3 x (4 values + bias) | 6 rotary pairs |
//        3 x (pack two pairs -> the wave's LDS stash)
//   SC (k stage's, beside the v stage): per head 36 micro-steps, two per slice: bias + q + bias-key requests | k values |
//        rotary pairs | q unpack | 12 x (one feature: 4 quad-broadcast FMAs + the bias key's) | half-wave sums | mask | max |
//        exp | denominator | P = five attention weights of the head
//   PV (v stage's, beside the next q stage): per head 18 steps: bias requests | v values | 12 x (one feature of o: 5 FMAs) |
//        2 x (pack three pairs -> ob)
// Same arithmetic as k_ln_qkv_attn4 (k_gemm.hip): q and the bias key / value rounded to bf16, k and v fp32, softmax in fp32
// (1 / den through v_rcp_f32 here).  The quad broadcast is folded into the FMA (v_fmac_f32_dpp, inline asm: hipcc emits a
// v_mov_b32_dpp + v_fmac pair): its DPP source is always written at least one slice (an MFMA and an LDS read) earlier, which
// covers the two wait states the hardware asks for between a VALU write and a DPP read of the same register.
#define CH_FMAC_QUAD(J)                                                                                              \
    template <>                                                                                                       \
    __device__ __forceinline__ void ch_fmac_quad<J>(float& acc, float k, float q) {                                   \
        asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[" #J "," #J "," #J "," #J "] row_mask:0xf bank_mask:0xf"           \
            : "+v"(acc) : "v"(k), "v"(q));                                                                           \
    }
template <int J>
__device__ __forceinline__ void ch_fmac_quad(float& acc, float k, float q);   // acc += (k of quad lane J) * q
CH_FMAC_QUAD(0) CH_FMAC_QUAD(1) CH_FMAC_QUAD(2) CH_FMAC_QUAD(3)
#undef CH_FMAC_QUAD

struct ChAttn {
    f32x4 rq[4];          // rotary factors of the residue position
    float mq[4];          // key validity of the four residues of the frame (quad)
    f32x4 b[3];           // requested biases of the head in flight
    f32x4 kb[3];          // SC: rotated bias key of the head; PV: bias value of the head
    uint32_t qu[6];       // SC: packed q of the head (from the stash)
    float e[12];          // values of the head in flight (q, k or v)
    float qf[12];         // SC: q of the head
    float sc[5];          // SC: scores / exponentials
    float t0, t1;         // SC: max, 1 / den
    float P[20];          // attention weights of the group's four heads
    float o[12];          // PV: output features of the head
};

__device__ __forceinline__ void ch_load4(const f32x16 (&acc)[3], ChAttn& s, int hd, int c) {
    const int ap = 3 * hd + c, ft = ap >> 2, a = ap & 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) s.e[4 * c + j] = acc[ft][4 * a + j] + s.b[c][j];
    CH_PIN4(s.e[4 * c], s.e[4 * c + 1], s.e[4 * c + 2], s.e[4 * c + 3]);
}
__device__ __forceinline__ void ch_rope1(ChAttn& s, int pp) {
    const float cs = s.rq[pp >> 2][pp & 3], sn = s.rq[2 + (pp >> 2)][pp & 3];
    const float x1 = s.e[2 * pp], x2 = s.e[2 * pp + 1];
    s.e[2 * pp] = x1 * cs - x2 * sn;
    s.e[2 * pp + 1] = x2 * cs + x1 * sn;
    CH_PIN2(s.e[2 * pp], s.e[2 * pp + 1]);
}

// bias3: LDS, the group's lane-ordered q biases of this lane half; stash_lane: the wave's stash + lane
__device__ __forceinline__ void ch_qe_step(const f32x16 (&acc)[3], const f32x4* bias3, uint32_t* stash_lane, ChAttn& s, int hd, int step) {
    if (step == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) s.b[c] = bias3[hd * 3 + c];
    } else if (step < 4) {
        ch_load4(acc, s, hd, step - 1);
    } else if (step < 10) {
        ch_rope1(s, step - 4);
    } else if (step < 13) {
        const int i = 2 * (step - 10);
        stash_lane[(hd * 6 + i) * 64] = pack_bf16(s.e[2 * i], s.e[2 * i + 1]);
        stash_lane[(hd * 6 + i + 1) * 64] = pack_bf16(s.e[2 * i + 2], s.e[2 * i + 3]);
    }
}
struct ChQeFill {   // slice s of 72: head s / 18, step s % 18
    const f32x16 (&acc)[3];
    const f32x4* bias3;
    uint32_t* stash_lane;
    ChAttn& st;
    __device__ __forceinline__ void operator()(int s) const {
        if (s % 18 < 13) ch_qe_step(acc, bias3, stash_lane, st, s / 18, s % 18);
    }
};

// bias3: the group's k biases; kb3: LDS, rotated bias key of the group's head 0 for this lane half (+ 6 f32x4 per head)
__device__ __forceinline__ void ch_sc_step(const f32x16 (&acc)[3], const f32x4* bias3, const f32x4* kb3, const uint32_t* stash_lane,
                                           ChAttn& s, int hd, int m) {
    if (m == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) s.b[c] = bias3[hd * 3 + c];
#pragma unroll
        for (int q = 0; q < 6; ++q) s.qu[q] = stash_lane[(hd * 6 + q) * 64];
    } else if (m < 4) {
        ch_load4(acc, s, hd, m - 1);
    } else if (m < 10) {
        ch_rope1(s, m - 4);
    } else if (m < 13) {
        const int c = m - 10;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            s.qf[4 * c + 2 * j] = bf16_lo(s.qu[2 * c + j]);
            s.qf[4 * c + 2 * j + 1] = bf16_hi(s.qu[2 * c + j]);
        }
        CH_PIN4(s.qf[4 * c], s.qf[4 * c + 1], s.qf[4 * c + 2], s.qf[4 * c + 3]);
    } else if (m == 13) {
#pragma unroll
        for (int c = 0; c < 3; ++c) s.kb[c] = kb3[hd * 6 + c];
#pragma unroll
        for (int j = 0; j < 5; ++j) s.sc[j] = 0.f;
        CH_PIN4(s.sc[0], s.sc[1], s.sc[2], s.sc[3]);
    } else if (m < 26) {
        const int i = m - 14;
        ch_fmac_quad<0>(s.sc[0], s.e[i], s.qf[i]);
        ch_fmac_quad<1>(s.sc[1], s.e[i], s.qf[i]);
        ch_fmac_quad<2>(s.sc[2], s.e[i], s.qf[i]);
        ch_fmac_quad<3>(s.sc[3], s.e[i], s.qf[i]);
        s.sc[4] += s.qf[i] * s.kb[i >> 2][i & 3];
        CH_PIN4(s.sc[0], s.sc[1], s.sc[2], s.sc[3]);
    } else if (m == 26) {
        s.sc[0] = half_sum2(s.sc[0]);
        s.sc[1] = half_sum2(s.sc[1]);
        s.sc[2] = half_sum2(s.sc[2]);
        CH_PIN2(s.sc[0], s.sc[1]);
    } else if (m == 27) {
        s.sc[3] = half_sum2(s.sc[3]);
        s.sc[4] = half_sum2(s.sc[4]);
        CH_PIN2(s.sc[3], s.sc[4]);
    } else if (m == 28) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s.sc[j] = s.mq[j] != 0.f ? s.sc[j] : -1e30f;
        CH_PIN4(s.sc[0], s.sc[1], s.sc[2], s.sc[3]);
    } else if (m == 29) {
        s.t0 = fmaxf(fmaxf(fmaxf(s.sc[0], s.sc[1]), fmaxf(s.sc[2], s.sc[3])), s.sc[4]);   // the bias key is never masked
        CH_PIN2(s.t0, s.sc[4]);
    } else if (m == 30) {
        float d[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) d[j] = __builtin_amdgcn_exp2f(s.sc[j] - s.t0);
#pragma unroll
        for (int j = 0; j < 4; ++j) s.sc[j] = s.mq[j] != 0.f ? d[j] : 0.f;
        s.sc[4] = d[4];
        CH_PIN4(s.sc[0], s.sc[1], s.sc[2], s.sc[3]);
    } else if (m == 31) {
        s.t1 = __builtin_amdgcn_rcpf(((s.sc[0] + s.sc[1]) + (s.sc[2] + s.sc[3])) + s.sc[4]);
        CH_PIN2(s.t1, s.sc[4]);
    } else if (m == 32) {
#pragma unroll
        for (int j = 0; j < 5; ++j) s.P[5 * hd + j] = s.sc[j] * s.t1;
        CH_PIN4(s.P[5 * hd], s.P[5 * hd + 1], s.P[5 * hd + 2], s.P[5 * hd + 3]);
    }
}
struct ChScFill {   // slice s of 72: head s / 18, micro-steps 2 (s % 18), 2 (s % 18) + 1
    const f32x16 (&acc)[3];
    const f32x4 *bias3, *kb3;
    const uint32_t* stash_lane;
    ChAttn& st;
    __device__ __forceinline__ void operator()(int s) const {
        ch_sc_step(acc, bias3, kb3, stash_lane, st, s / 18, 2 * (s % 18));
        ch_sc_step(acc, bias3, kb3, stash_lane, st, s / 18, 2 * (s % 18) + 1);
    }
};

// bias3: the group's v biases (lane order); bv3: LDS, bf16-rounded bias value of the group's head 0 for this lane half (+ 6 per head)
__device__ __forceinline__ void ch_pv_step(const f32x16 (&acc)[3], const f32x4* bias3, const f32x4* bv3, ChAttn& s, uint32_t (&ob)[24], int hd,
                                           int step) {
    if (step == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            s.b[c] = bias3[hd * 3 + c];
            s.kb[c] = bv3[hd * 6 + c];
        }
    } else if (step < 4) {
        ch_load4(acc, s, hd, step - 1);
    } else if (step < 16) {
        const int i = step - 4;
        float o = s.P[5 * hd + 4] * s.kb[i >> 2][i & 3];
        ch_fmac_quad<0>(o, s.e[i], s.P[5 * hd + 0]);
        ch_fmac_quad<1>(o, s.e[i], s.P[5 * hd + 1]);
        ch_fmac_quad<2>(o, s.e[i], s.P[5 * hd + 2]);
        ch_fmac_quad<3>(o, s.e[i], s.P[5 * hd + 3]);
        s.o[i] = o;
        asm volatile("" : "+v"(s.o[i]));
    } else if (step < 18) {
        const int q0 = 3 * (step - 16);
#pragma unroll
        for (int q = q0; q < q0 + 3; ++q) ob[hd * 6 + q] = pack_bf16(s.o[2 * q], s.o[2 * q + 1]);
    }
}
struct ChPvFill {   // slice s of 72: head s / 18, step s % 18
    const f32x16 (&acc)[3];
    const f32x4 *bias3, *bv3;
    ChAttn& st;
    uint32_t (&ob)[24];
    __device__ __forceinline__ void operator()(int s) const { ch_pv_step(acc, bias3, bv3, st, ob, s / 18, s % 18); }
};
__device__ __forceinline__ void ch_pv_plain(const f32x16 (&acc)[3], const f32x4* bias3, const f32x4* bv3, ChAttn& st, uint32_t (&ob)[24]) {
#pragma unroll
    for (int s = 0; s < 72; ++s) ch_pv_step(acc, bias3, bv3, st, ob, s / 18, s % 18);
}

// ---- temporal q / k epilogue of one stage (4 heads of head group g) as a step machine: 15 steps per head ----------------
//   0 request the head's 12 biases (LDS) | 1..3 four accumulator values + bias | 4..9 one rotary pair each | 10..12 pack two
//   pairs each | 13, 14 the two stores.  Branch-free: a wave without real rows stores to a dump (p0 == p1 == dump, hs == 0).
// p0 / p1: the lane's slot in k-step 0 / k-step 1 of head 0's fragment of its (sequence, tile); q: k-step 1 is an 8-byte
// slot; k: a 16-byte slot whose entries 4, 5 hold the constant 1.0 (they pick up the softmax shift riding in q, k_flash.hip).
struct ChQkEpi {
    f32x4 rq[4];                 // rotary factors of the token's frame
    unsigned char *p0, *p1;      // this stage's group: head 4 g's slots
    long hs;                     // head stride of the fragment buffer
    f32x4 b[3];
    float e[12];
    uint32_t u[6];
};
template <bool ISK>
__device__ __forceinline__ void ch_qk_step(const f32x16 (&acc)[3], const f32x4* bias3, ChQkEpi& s, int hd, int step) {
    if (step == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) s.b[c] = bias3[hd * 3 + c];
    } else if (step < 4) {
        const int c = step - 1, ap = 3 * hd + c, ft = ap >> 2, a = ap & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) s.e[4 * c + j] = acc[ft][4 * a + j] + s.b[c][j];
        CH_PIN4(s.e[4 * c], s.e[4 * c + 1], s.e[4 * c + 2], s.e[4 * c + 3]);
    } else if (step < 10) {
        const int pp = step - 4;
        const float cs = s.rq[pp >> 2][pp & 3], sn = s.rq[2 + (pp >> 2)][pp & 3];
        const float x1 = s.e[2 * pp], x2 = s.e[2 * pp + 1];
        s.e[2 * pp] = x1 * cs - x2 * sn;
        s.e[2 * pp + 1] = x2 * cs + x1 * sn;
        CH_PIN2(s.e[2 * pp], s.e[2 * pp + 1]);
    } else if (step < 13) {
        const int i = 2 * (step - 10);
        s.u[i] = pack_bf16(s.e[2 * i], s.e[2 * i + 1]);
        s.u[i + 1] = pack_bf16(s.e[2 * i + 2], s.e[2 * i + 3]);
        CH_PIN2(s.u[i], s.u[i + 1]);
    } else if (step == 13) {
        *reinterpret_cast<u32x4*>(s.p0 + hd * s.hs) = u32x4{s.u[0], s.u[1], s.u[2], s.u[3]};
    } else if (step == 14) {
        if (ISK) *reinterpret_cast<u32x4*>(s.p1 + hd * s.hs) = u32x4{s.u[4], s.u[5], 0x3f803f80u, 0u};
        else *reinterpret_cast<u32x2*>(s.p1 + hd * s.hs) = u32x2{s.u[4], s.u[5]};
    }
}
template <bool ISK>
struct ChQkFill {   // slice s of 72: head s / 18, step s % 18
    const f32x16 (&acc)[3];
    const f32x4* bias3;          // LDS: lane-ordered biases of the stage's head group, this lane half
    ChQkEpi& st;
    __device__ __forceinline__ void operator()(int s) const {
        if (s % 18 < 15) ch_qk_step<ISK>(acc, bias3, st, s / 18, s % 18);
    }
};

// ---- temporal v epilogue (non-transposed stage): 12 items (tile j, residue i), three steps each: bias request | add | pack +
//      store.  The lane's three columns (feature columns 32 j + n of the head group) sit in heads hd[j], V^T rows d[j].
struct ChVEpi {
    unsigned char* vdst;         // this stage's group: V^T fragment of (sequence (b, 0), head 4 g, this tile) + the wave's piece
    long hs, ss;                 // head stride, sequence stride (bytes)
    int off[3];                  // hd[j] * hs + d[j] * 16 of the lane's three columns
    float bv[3];
    float e[4];
};
__device__ __forceinline__ void ch_v_step(const f32x16 (&acc)[3], const float* bias_n, ChVEpi& s, int item, int step) {
    const int j = item >> 2, i = item & 3;
    if (step == 0) {
        if (i == 0) s.bv[j] = bias_n[32 * j];
    } else if (step == 1) {
#pragma unroll
        for (int a = 0; a < 4; ++a) s.e[a] = acc[j][4 * a + i] + s.bv[j];
        CH_PIN4(s.e[0], s.e[1], s.e[2], s.e[3]);
    } else {
        const u32x2 v = {pack_bf16(s.e[0], s.e[1]), pack_bf16(s.e[2], s.e[3])};
        *reinterpret_cast<u32x2*>(s.vdst + s.off[j] + i * s.ss) = v;
    }
}
struct ChVFill {   // slice s of 72: every second slice one of the 36 steps
    const f32x16 (&acc)[3];
    const float* bias_n;         // LDS: biases of the stage's head group, + lane & 31
    ChVEpi& st;
    __device__ __forceinline__ void operator()(int s) const {
        if (s % 2 == 0) ch_v_step(acc, bias_n, st, (s / 2) / 3, (s / 2) % 3);
    }
};
__device__ __forceinline__ void ch_v_plain(const f32x16 (&acc)[3], const float* bias_n, ChVEpi& st) {
#pragma unroll
    for (int item = 0; item < 12; ++item)
#pragma unroll
        for (int step = 0; step < 3; ++step) ch_v_step(acc, bias_n, st, item, step);
}

// =================================================================================================
template <int NW>
__global__ __launch_bounds__(NW * 64, 1) void k_chain_l4(const ChainParams p) {
    static_assert(NW == 4, "the LDS tables and the DMA split assume four waves");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kChSmemBytes];
    float* tab = reinterpret_cast<float*>(smem + kRingBytes);
    const int w = __builtin_amdgcn_readfirstlane(wave_id()), lane = lane_id(), hh = lane >> 5, n = lane & 31;
    const int l = n & 3, tt = 4 * ((n >> 2) & 1) + (n >> 3);
    unsigned long long st[10] = {};
#define CH_STAMP(i)                              \
    st[i] = __builtin_amdgcn_s_memtime();        \
    __builtin_amdgcn_sched_barrier(0)
    CH_STAMP(0);
    ChStream ws{p.ws_l, p.ws_o, p.ws_t, lds_addr(smem), (unsigned)lane * 16u, w};
    ws.issue_slot(0);
    ws.issue_slot(1);
    ws.issue_slot(2);
    // ---- geometry: the wave's 32 rows = frames t0 .. t0 + 7 of sample b, four residues each
    const long tile = (long)blockIdx.x * NW + w;
#ifdef MDGEN_DEV_CHAIN_NOSTORE   // (experiment build, timing only)
    const bool live = tile * 32 < p.nrows && p.nrows < 0;
#else
    const bool live = tile * 32 < p.nrows;
#endif
    const int base_tok = __builtin_amdgcn_readfirstlane(live ? (int)(tile * 32) : 0);
    const int tok = base_tok + 4 * tt + l;
    const int f0 = base_tok >> 2;
    const int b = f0 / p.T, t0 = f0 - b * p.T;
    const int ltok = live ? tok : -1;
    // ---- LDS tables (plain loads: hipcc's waits for them also land the three slots requested above)
    {
        const int tid = threadIdx.x;
        for (int i = tid; i < kC; i += NW * 64) {
            tab[TB_BQL + i] = p.bq_l[i];
            tab[TB_BKL + i] = p.bk_l[i];
            tab[TB_BVL + i] = p.bv_l[i];
            tab[TB_BOL + i] = p.bo_l[i];
            tab[TB_BQT + i] = p.bq_t[i];
            tab[TB_BKT + i] = p.bk_t[i];
            tab[TB_BVT + i] = p.bv_t[i];
            tab[TB_BVLR + i] = bf16_lo(pack_bf16(p.bias_v_l[i], 0.f));
            // learned bias key of the residue axis, rotated at position 4 like every key (mha.py:265-268, 356-357), rounded to
            // bf16, in lane order [(head, half)][12]: slot e of half h is feature 6 h + e / 2 (+ 12 for odd e)
            const int head = i / 24, h2 = (i / 12) & 1, e = i % 12, q = e >> 1;
            const float* bk = p.bias_k_l + head * kDH;
            const float* rc = p.rope + 4 * kRopeRow + 16 * h2;
            const float x1 = bk[6 * h2 + q], x2 = bk[6 * h2 + q + 12], cs = rc[q], sn = rc[8 + q];
            tab[TB_KBL + i] = bf16_lo(pack_bf16((e & 1) ? x2 * cs + x1 * sn : x1 * cs - x2 * sn, 0.f));
        }
        for (int i = tid; i < 5 * kRopeRow; i += NW * 64) tab[TB_ROPEL + i] = p.rope[i];
        const long mo = p.mm.row_off(base_tok);
#pragma unroll
        for (int k = 0; k < 6; ++k) tab[TB_GATE + w * kC + lane + 64 * k] = p.mm.mod[mo + (long)p.gate_l * kC + lane + 64 * k];
        {   // rotary rows of the wave's 8 frames (row stride 36 floats: conflict-free 16-byte reads)
            const int row = lane >> 3, piece = lane & 7;
            const f32x4 v = *reinterpret_cast<const f32x4*>(p.rope + (long)(t0 + row) * kRopeRow + piece * 4);
            *reinterpret_cast<f32x4*>(tab + TB_ROPET + w * 288 + row * 36 + piece * 4) = v;
        }
    }
    const float mval = p.mk.at(tok);
    bf16x8 xf[24];
    rows_ln(p.h, ltok, p.mm, p.shift_l, p.scale_l, 1e-6f, xf);
    __syncthreads();
    ChRing r;
    r.lane_base = smem + lane * 16;
    r.slot = 0;
    ring_barrier<12>();   // slot 0 has landed
#pragma unroll
    for (int i = 0; i < 5; ++i) r.wr[i] = *reinterpret_cast<const bf16x8*>(r.lane_base + i * 1024);
    uint32_t* stash_lane = reinterpret_cast<uint32_t*>(smem + kRingBytes + kChTabBytes) + w * 24 * 64 + lane;
    f32x16 acc[3];
    CH_STAMP(1);
    f32x16 accb[3];   // second accumulator buffer: stage n + 1 runs while stage n's epilogue reads stage n's
    // ================= residue axis: q, k, v head group by head group; the 5-key attention in registers =================
    bf16x8 of[24];   // attention output rows as the B operand of the out-projection (k order: lane's own value list)
    {
        ChAttn at;
        uint32_t ob[24];
        {
            const f32x4* rc = reinterpret_cast<const f32x4*>(tab + TB_ROPEL + l * kRopeRow + 16 * hh);
#pragma unroll
            for (int i = 0; i < 4; ++i) at.rq[i] = rc[i];
        }
        at.mq[0] = ch_quad<0>(mval);
        at.mq[1] = ch_quad<1>(mval);
        at.mq[2] = ch_quad<2>(mval);
        at.mq[3] = ch_quad<3>(mval);
#pragma unroll
        for (int i = 0; i < 20; ++i) at.P[i] = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) at.e[i] = 0.f;
        ch_zero(accb);
        const f32x4* bq3 = reinterpret_cast<const f32x4*>(tab + TB_BQL + hh * 48);    // + g * 24 (f32x4 units)
        const f32x4* bk3 = reinterpret_cast<const f32x4*>(tab + TB_BKL + hh * 48);
        const f32x4* bv3 = reinterpret_cast<const f32x4*>(tab + TB_BVL + hh * 48);
        const f32x4* kbl = reinterpret_cast<const f32x4*>(tab + TB_KBL + hh * 12);    // + head * 6
        const f32x4* bvr = reinterpret_cast<const f32x4*>(tab + TB_BVLR + hh * 12);   // + head * 6
        // stage n = 3 g + {0 q, 1 k, 2 v} accumulates into buffer n & 1 and carries the epilogue of stage n - 1:
        //   q_g: PV of group g - 1 | k_g: QE of g | v_g: SC of g.      Two groups per trip: the buffers alternate statically.
#define CH_PUT(G)                                                                                                                \
    if (gp == G) {                                                                                                               \
        _Pragma("unroll") for (int i = 0; i < 6; ++i)                                                                            \
            of[6 * G + i] = __builtin_bit_cast(bf16x8, u32x4{ob[4 * i], ob[4 * i + 1], ob[4 * i + 2], ob[4 * i + 3]}); \
    }
#pragma unroll 1
        for (int t2 = 0; t2 < 2; ++t2) {
            const int g0 = 2 * t2, g1 = g0 + 1;
            {   // q of g0 -> A; PV of g0 - 1 reads B (first trip: zeros, result unused)
                const int gq = g0 == 0 ? 0 : g0 - 1;
                ChPvFill f{accb, bv3 + 24 * gq, bvr + 24 * gq, at, ob};
                ch_zero(acc);
                ch_stage<false>(r, ws, acc, xf, f);
                const int gp = g0 - 1;
                CH_PUT(1)
            }
            {   // k of g0 -> B; QE of g0 reads A
                ChQeFill f{acc, bq3 + 24 * g0, stash_lane, at};
                ch_zero(accb);
                ch_stage<false>(r, ws, accb, xf, f);
            }
            {   // v of g0 -> A; SC of g0 reads B
                ChScFill f{accb, bk3 + 24 * g0, kbl + 24 * g0, stash_lane, at};
                ch_zero(acc);
                ch_stage<false>(r, ws, acc, xf, f);
            }
            {   // q of g1 -> B; PV of g0 reads A
                ChPvFill f{acc, bv3 + 24 * g0, bvr + 24 * g0, at, ob};
                ch_zero(accb);
                ch_stage<false>(r, ws, accb, xf, f);
                const int gp = g0;
                CH_PUT(0) CH_PUT(2)
            }
            {   // k of g1 -> A; QE of g1 reads B
                ChQeFill f{accb, bq3 + 24 * g1, stash_lane, at};
                ch_zero(acc);
                ch_stage<false>(r, ws, acc, xf, f);
            }
            {   // v of g1 -> B; SC of g1 reads A
                ChScFill f{acc, bk3 + 24 * g1, kbl + 24 * g1, stash_lane, at};
                ch_zero(accb);
                ch_stage<false>(r, ws, accb, xf, f);
            }
        }
        {   // the last group's PV has no stage to ride beside: the out-projection needs all of `of`
            ch_pv_plain(accb, bv3 + 72, bvr + 72, at, ob);
            const int gp = 3;
            CH_PUT(3)
        }
#undef CH_PUT
    }
    CH_STAMP(2);
    // ================= residue axis: out-projection + gated residual =================
    {
        unsigned char* hb = reinterpret_cast<unsigned char*>(p.h) + (unsigned)tok * (unsigned)(kC * 4) + (unsigned)hh * 16u;
        const float* gate = tab + TB_GATE + w * kC + 4 * hh;
        const float* bo = tab + TB_BOL + 4 * hh;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 hq[24];
#pragma unroll
            for (int i = 0; i < 24; ++i) hq[i] = *reinterpret_cast<const f32x4*>(hb + 32u * (24 * half + i));
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                ch_zero(acc);
                ch_stage<false>(r, ws, acc, of);
#pragma unroll
                for (int tl = 0; tl < 3; ++tl)
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int il = 12 * s2 + 4 * tl + a, i = 24 * half + il;   // 16-byte piece: features 8 i + 4 hh ..
                        const f32x4 gv = *reinterpret_cast<const f32x4*>(gate + 8 * i);
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(bo + 8 * i);
                        f32x4 o = hq[il];
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] += gv[j] * (acc[tl][4 * a + j] + bv[j]);
                        if (live) *reinterpret_cast<f32x4*>(hb + 32u * i) = o;
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    CH_STAMP(3);
    // ================= temporal axis: LayerNorm of the updated rows, q / k / v -> fragments =================
    // The updated rows come back from L2 (this wave's own stores, acknowledged by the wait; a 192-register image kept across
    // the out-projection spills): 48 loads, then the same LayerNorm as the prologue.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    rows_ln(p.h, ltok, p.mm, p.shift_t, p.scale_t, 1e-6f, xf);
    CH_STAMP(4);
    const int ntile = p.ntile;
    const int tl5 = t0 >> 5, s0 = t0 & 31;                   // key tile and first key slot of the wave's 8 frames (uniform)
    const int seq = b * 4 + l;                               // the lane's temporal sequence
    const int slot = s0 + tt;
    const int g8 = s0 >> 3;
    const long seq0 = (long)b * 4 * kH * ntile + tl5;        // fragment index of (sequence (b, 0), head 0, this tile)
    {
        ChQkEpi qe;
        {
            const f32x4* rc = reinterpret_cast<const f32x4*>(tab + TB_ROPET + w * 288 + tt * 36 + 16 * hh);
#pragma unroll
            for (int i = 0; i < 4; ++i) qe.rq[i] = rc[i];
        }
        // a wave without real rows computes like the others (it shares the ring) and stores into the attention-output buffer,
        // which nobody reads before the attention kernel rewrites it
        unsigned char* dump = reinterpret_cast<unsigned char*>(p.dump) + lane * 16;
        const long fq = ((long)seq * kH * ntile + tl5);
        unsigned char* q0 = live ? p.qf + fq * kFragQ + (hh * 32 + slot) * 16 : dump;
        unsigned char* q1 = live ? p.qf + fq * kFragQ + 1024 + (hh * 32 + slot) * 8 : dump;
        unsigned char* k0 = live ? p.kf + fq * kFragK + (hh * 32 + slot) * 16 : dump;
        unsigned char* k1 = live ? p.kf + fq * kFragK + 1024 + (hh * 32 + slot) * 16 : dump;
        const long qhs = live ? (long)ntile * kFragQ : 0, khs = live ? (long)ntile * kFragK : 0;
        const f32x4* bq3 = reinterpret_cast<const f32x4*>(tab + TB_BQT + hh * 48);   // + g * 24 (f32x4 units)
        const f32x4* bk3 = reinterpret_cast<const f32x4*>(tab + TB_BKT + hh * 48);
        ChVEpi ve;
        // 8-byte piece of the V^T fragment: [k-step g8 >> 1][key half hh][row d][slots 4 (g8 & 1) .. + 3]
        unsigned char* v0 = live ? p.vf + seq0 * kFragV + (long)(g8 >> 1) * 800 + hh * 400 + (g8 & 1) * 8 : dump;
        ve.hs = live ? (long)ntile * kFragV : 0;
        ve.ss = live ? (long)kH * ntile * kFragV : 0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int col = 32 * j + n, hd = col / kDH, d = col - hd * kDH;
            ve.off[j] = live ? (int)(hd * ve.hs) + d * 16 : 0;
        }
        const float* bvn = tab + TB_BVT + n;   // + g * 96
        // stage n:   0 q0 -> A | 1 q1 -> B, q0's epilogue | 2 q2 -> A, q1's | 3 q3 -> B | 4 k0 -> A, q3's | 5 .. 7 k1 .. k3 |
        //            8 v0 -> A, k3's epilogue | 9 v1 -> B, v0's | 10 v2 -> A | 11 v3 -> B, v2's | then v3's epilogue
        ch_zero(acc);
        ch_stage<false>(r, ws, acc, xf);
        qe.hs = qhs;
#pragma unroll 1
        for (int n2 = 0; n2 < 2; ++n2) {   // stages 2 n2 + 1 (-> B), 2 n2 + 2 (-> A): the epilogues of q stages 2 n2, 2 n2 + 1
            {
                qe.p0 = q0 + (long)(8 * n2) * qhs;
                qe.p1 = q1 + (long)(8 * n2) * qhs;
                ChQkFill<false> f{acc, bq3 + 48 * n2, qe};
                ch_zero(accb);
                ch_stage<false>(r, ws, accb, xf, f);
            }
            {
                qe.p0 = q0 + (long)(8 * n2 + 4) * qhs;
                qe.p1 = q1 + (long)(8 * n2 + 4) * qhs;
                ChQkFill<false> f{accb, bq3 + 48 * n2 + 24, qe};
                ch_zero(acc);
                ch_stage<false>(r, ws, acc, xf, f);
            }
        }
        qe.hs = khs;
        {   // stages 5 (-> B), 6 (-> A): k0's, k1's epilogues
            qe.p0 = k0;
            qe.p1 = k1;
            ChQkFill<true> f{acc, bk3, qe};
            ch_zero(accb);
            ch_stage<false>(r, ws, accb, xf, f);
        }
        {
            qe.p0 = k0 + 4 * khs;
            qe.p1 = k1 + 4 * khs;
            ChQkFill<true> f{accb, bk3 + 24, qe};
            ch_zero(acc);
            ch_stage<false>(r, ws, acc, xf, f);
        }
        {   // stages 7 (-> B), 8 (-> A; the first v stage: non-transposed product): k2's, k3's epilogues
            qe.p0 = k0 + 8 * khs;
            qe.p1 = k1 + 8 * khs;
            ChQkFill<true> f{acc, bk3 + 48, qe};
            ch_zero(accb);
            ch_stage<false>(r, ws, accb, xf, f);
        }
        {
            qe.p0 = k0 + 12 * khs;
            qe.p1 = k1 + 12 * khs;
            ChQkFill<true> f{accb, bk3 + 72, qe};
            ch_zero(acc);
            ch_stage<true>(r, ws, acc, xf, f);
        }
        CH_STAMP(5);
        {
            ve.vdst = v0;
            ChVFill f{acc, bvn, ve};
            ch_zero(accb);
            ch_stage<true>(r, ws, accb, xf, f);
        }
        {
            ve.vdst = v0 + 4 * ve.hs;
            ChVFill f{accb, bvn + 96, ve};
            ch_zero(acc);
            ch_stage<true>(r, ws, acc, xf, f);
        }
        {
            ve.vdst = v0 + 8 * ve.hs;
            ChVFill f{acc, bvn + 192, ve};
            ch_zero(accb);
            ch_stage<true>(r, ws, accb, xf, f);
        }
        CH_STAMP(6);
        ve.vdst = v0 + 12 * ve.hs;
        ch_v_plain(accb, bvn + 288, ve);
    }
    {
        CH_STAMP(7);
        // the all-ones row 24 of every fragment the wave touched: 4 sequences x 16 heads x 2 key halves, 8 bytes each
        if (live) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int idx = lane + 64 * k, li = idx & 3, head = (idx >> 2) & 15, hk = idx >> 6;
                *reinterpret_cast<u32x2*>(p.vf + (seq0 + ((long)li * kH + head) * ntile) * kFragV + (long)(g8 >> 1) * 800 + hk * 400 +
                                          kDH * 16 + (g8 & 1) * 8) = u32x2{0x3f803f80u, 0x3f803f80u};
            }
        }
        // key-validity bits of the wave's 8 slots: one BYTE of each sequence's word of this tile (kernels.h flash_vmask)
        const unsigned long long bal = __ballot(mval != 0.f);
        const uint32_t lo = (uint32_t)bal;   // lanes 0..31: bit n = token (frame 4 ((n >> 2) & 1) + (n >> 3), residue n & 3)
        if (live && lane < 4) {
            uint32_t byte = 0;
#pragma unroll
            for (int f = 0; f < 8; ++f) byte |= ((lo >> (8 * (f & 3) + 4 * (f >> 2) + lane)) & 1u) << f;
            unsigned char* vm = reinterpret_cast<unsigned char*>(p.vmask + (long)(b * 4 + lane) * p.vmask_stride);
            vm[tl5 * 4 + g8] = (unsigned char)byte;
        }
        // ---- the wave that owns a sample's LAST 8 frames finishes its four sequences: the learned bias key / value as a
        //      real entry (key slot T) of the K / V^T fragments, the validity words behind the last real key
        if (live && t0 + 8 == p.T) {
            const int len = p.T, kt = len >> 5, sl = len & 31;
            if (sl == 0) {
                // the bias key opens a tile of its own, which no stage has touched: zero it (stale bytes x P = 0 would still
                // poison the PV sum), then the ones row
                for (int it = lane; it < 4 * kH * (kFragK / 16); it += 64) {
                    const int sh = it / (kFragK / 16), o16 = it % (kFragK / 16);
                    const long fi = ((long)(b * 4 + (sh >> 4)) * kH + (sh & 15)) * ntile + kt;
                    // (k-step 1 keeps the constant 1.0 pair in entries 4, 5 of every slot, as k_ln_qkv writes it)
                    *reinterpret_cast<u32x4*>(p.kf + fi * kFragK + o16 * 16) = u32x4{0u, 0u, o16 >= 64 ? 0x3f803f80u : 0u, 0u};
                }
                for (int it = lane; it < 4 * kH * (kFragV / 16); it += 64) {
                    const int sh = it / (kFragV / 16), o16 = it % (kFragV / 16);
                    const long fi = ((long)(b * 4 + (sh >> 4)) * kH + (sh & 15)) * ntile + kt;
                    const uint32_t v = (o16 % 25) == kDH ? 0x3f803f80u : 0u;
                    *reinterpret_cast<u32x4*>(p.vf + fi * kFragV + o16 * 16) = u32x4{v, v, v, v};
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                // query slots behind the last frame, up to the end of the 64-query chunk a k_flash wave loads: finite values
                // (a NaN pattern in an unused query column would send the whole wave to the robust loop; k_ln_qkv's
                // padding rows store finite values there as well)
                const int pend = min(ntile * 32, ((len + 63) >> 6) << 6), np_ = pend - len;
                for (int it = lane; it < 4 * kH * 2 * np_; it += 64) {
                    const int pz = len + it % np_, h2 = (it / np_) & 1, sh = it / (2 * np_);
                    unsigned char* base = p.qf + (((long)(b * 4 + (sh >> 4)) * kH + (sh & 15)) * ntile + (pz >> 5)) * kFragQ;
                    *reinterpret_cast<u32x4*>(base + (h2 * 32 + (pz & 31)) * 16) = u32x4{0u, 0u, 0u, 0u};
                    *reinterpret_cast<u32x2*>(base + 1024 + (h2 * 32 + (pz & 31)) * 8) = u32x2{0u, 0u};
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {   // K: (sequence li, head, half h2) -> 12 rotated values of key slot sl
                const int idx = lane + 64 * k, li = idx & 3, head = (idx >> 2) & 15, h2 = idx >> 6;
                const float* bk = p.bias_k_t + head * kDH;
                const float* rc = p.rope + (long)len * kRopeRow + 16 * h2;
                float e[12];
#pragma unroll
                for (int pp = 0; pp < 6; ++pp) {
                    const float x1 = bk[6 * h2 + pp], x2 = bk[6 * h2 + pp + 12], cs = rc[pp], sn = rc[8 + pp];
                    e[2 * pp] = x1 * cs - x2 * sn;
                    e[2 * pp + 1] = x2 * cs + x1 * sn;
                }
                unsigned char* base = p.kf + (((long)(b * 4 + li) * kH + head) * ntile + kt) * kFragK;
                *reinterpret_cast<u32x4*>(base + (h2 * 32 + sl) * 16) =
                    u32x4{pack_bf16(e[0], e[1]), pack_bf16(e[2], e[3]), pack_bf16(e[4], e[5]), pack_bf16(e[6], e[7])};
                *reinterpret_cast<u32x4*>(base + 1024 + (h2 * 32 + sl) * 16) =
                    u32x4{pack_bf16(e[8], e[9]), pack_bf16(e[10], e[11]), 0x3f803f80u, 0u};
            }
            {   // V^T: row d of (sequence li, head): key slot sl = register r of key half hk (k_gemm.hip write_bias_slots)
                const int hk = (sl >> 2) & 1, rr = (sl & 3) + 4 * (sl >> 3);
                // (row 24, the all-ones row, gets its 1.0 in that slot too: the bias key's P must reach the denominator)
                for (int it = lane; it < 4 * kH * (kDH + 1); it += 64) {
                    const int d = it % (kDH + 1), sh = it / (kDH + 1), li = sh >> 4, head = sh & 15;
                    const int dpsi = 12 * ((d >> 2) & 1) + 4 * (d >> 3) + (d & 3);
                    const uint32_t v = d == kDH ? 0x3f80u : pack_bf16(p.bias_v_t[head * kDH + (d == kDH ? 0 : dpsi)], 0.f);
                    unsigned char* base = p.vf + (((long)(b * 4 + li) * kH + head) * ntile + kt) * kFragV;
                    *reinterpret_cast<uint16_t*>(base + (rr >> 3) * 800 + hk * 400 + d * 16 + (rr & 7) * 2) = (uint16_t)v;
                }
            }
            if (lane < 4) {   // validity: the bias key's bit, zeros behind it
                unsigned char* vm = reinterpret_cast<unsigned char*>(p.vmask + (long)(b * 4 + lane) * p.vmask_stride);
                for (int by = kt * 4 + (sl >> 3); by < kt * 4 + 4; ++by) vm[by] = by == kt * 4 + (sl >> 3) ? 1 : 0;
                uint32_t* vw = p.vmask + (long)(b * 4 + lane) * p.vmask_stride;
                for (int wi = kt + 1; wi < p.vmask_stride; ++wi) vw[wi] = 0u;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's look-ahead DMAs must not outlive the workgroup's LDS
    CH_STAMP(8);
    // phase stamps (measurement only, p.trace null in normal operation): collected in SGPRs, written by ONE branch
    if (p.trace && lane == 0) {
        const long i = ((long)blockIdx.x * NW + w) * 10;
        if (i + 10 <= p.trace_cap) {
#pragma unroll
            for (int k = 0; k < 9; ++k) p.trace[i + k] = st[k];
        }
    }
#undef CH_STAMP
}

void launch_chain_l4(const ChainParams& p, hipStream_t s) {
    const long tiles = (p.nrows + 31) / 32;
    hipLaunchKernelGGL((k_chain_l4<4>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, p);
}

}  // namespace mdg
