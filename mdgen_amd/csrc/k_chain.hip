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

template <int HD>
__device__ __forceinline__ void ch_attn_q(const f32x16 (&acc)[3], const float* tab, int g, int hh, const f32x4 (&rq)[4],
                                          uint32_t* stash_lane) {
    float e[12];
    ch_head12<HD>(acc, reinterpret_cast<const f32x4*>(tab + TB_BQL + (g * 2 + hh) * 48) + HD * 3, e);
    ch_rope12(e, rq);
#pragma unroll
    for (int q = 0; q < 6; ++q) stash_lane[(HD * 6 + q) * 64] = pack_bf16(e[2 * q], e[2 * q + 1]);
}

// scores of head HD against the four keys of the quad + the learned bias key, softmax -> P[5]
template <int HD>
__device__ __forceinline__ void ch_attn_scores(const f32x16 (&acc)[3], const float* tab, int g, int hh, const f32x4 (&rq)[4],
                                               const uint32_t* stash_lane, float mval, float (&P)[5]) {
    float k[12], qf[12], kb[12];
    ch_head12<HD>(acc, reinterpret_cast<const f32x4*>(tab + TB_BKL + (g * 2 + hh) * 48) + HD * 3, k);
    ch_rope12(k, rq);
    {
        const f32x4* kp = reinterpret_cast<const f32x4*>(tab + TB_KBL + ((4 * g + HD) * 2 + hh) * 12);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f32x4 v = kp[c];
#pragma unroll
            for (int j = 0; j < 4; ++j) kb[4 * c + j] = v[j];
        }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const uint32_t u = stash_lane[(HD * 6 + q) * 64];
        qf[2 * q] = bf16_lo(u);
        qf[2 * q + 1] = bf16_hi(u);
    }
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        s[0] += qf[i] * ch_quad<0>(k[i]);
        s[1] += qf[i] * ch_quad<1>(k[i]);
        s[2] += qf[i] * ch_quad<2>(k[i]);
        s[3] += qf[i] * ch_quad<3>(k[i]);
        s[4] += qf[i] * kb[i];
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) s[j] = half_sum2(s[j]);
    const float m0 = ch_quad<0>(mval), m1 = ch_quad<1>(mval), m2 = ch_quad<2>(mval), m3 = ch_quad<3>(mval);
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
    for (int j = 0; j < 5; ++j) P[j] = s[j] * inv;
}

// o of head HD = sum_j P_j v_j + P_4 bias_v -> six packed bf16 pairs ob[6 HD .. 6 HD + 5]
template <int HD>
__device__ __forceinline__ void ch_attn_out(const f32x16 (&acc)[3], const float* tab, int g, int hh, const float (&Pw)[5],
                                            uint32_t (&ob)[24]) {
    float v[12], o[12];
    ch_head12<HD>(acc, reinterpret_cast<const f32x4*>(tab + TB_BVL + (g * 2 + hh) * 48) + HD * 3, v);
    const f32x4* bp = reinterpret_cast<const f32x4*>(tab + TB_BVLR + (4 * g + HD) * kDH + 12 * hh);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const f32x4 bv = bp[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 4 * c + j;
            o[i] = Pw[0] * ch_quad<0>(v[i]) + Pw[1] * ch_quad<1>(v[i]) + Pw[2] * ch_quad<2>(v[i]) + Pw[3] * ch_quad<3>(v[i]) +
                   Pw[4] * bv[j];
        }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) ob[HD * 6 + q] = pack_bf16(o[2 * q], o[2 * q + 1]);
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
    // rotary factors of the residue axis: position = residue index
    f32x4 rql[4];
    // ================= residue axis: q, k, v head group by head group; the 5-key attention in registers =================
    bf16x8 of[24];   // attention output rows as the B operand of the out-projection (k order: lane's own value list)
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {
        ch_zero(acc);
        ch_stage<false>(r, ws, acc, xf);
        {
            const f32x4* rc = reinterpret_cast<const f32x4*>(tab + TB_ROPEL + l * kRopeRow + 16 * hh);
#pragma unroll
            for (int i = 0; i < 4; ++i) rql[i] = rc[i];
        }
        ch_attn_q<0>(acc, tab, g, hh, rql, stash_lane);
        ch_attn_q<1>(acc, tab, g, hh, rql, stash_lane);
        ch_attn_q<2>(acc, tab, g, hh, rql, stash_lane);
        ch_attn_q<3>(acc, tab, g, hh, rql, stash_lane);
        __builtin_amdgcn_sched_barrier(0);
        ch_zero(acc);
        ch_stage<false>(r, ws, acc, xf);
        float P0[5], P1[5], P2[5], P3[5];
        ch_attn_scores<0>(acc, tab, g, hh, rql, stash_lane, mval, P0);
        __builtin_amdgcn_sched_barrier(0);
        ch_attn_scores<1>(acc, tab, g, hh, rql, stash_lane, mval, P1);
        __builtin_amdgcn_sched_barrier(0);
        ch_attn_scores<2>(acc, tab, g, hh, rql, stash_lane, mval, P2);
        __builtin_amdgcn_sched_barrier(0);
        ch_attn_scores<3>(acc, tab, g, hh, rql, stash_lane, mval, P3);
        __builtin_amdgcn_sched_barrier(0);
        // the 20 attention weights wait in the (now free) q stash as bf16 pairs while the V stage runs
        {
            const float Pf[20] = {P0[0], P0[1], P0[2], P0[3], P0[4], P1[0], P1[1], P1[2], P1[3], P1[4],
                                  P2[0], P2[1], P2[2], P2[3], P2[4], P3[0], P3[1], P3[2], P3[3], P3[4]};
#pragma unroll
            for (int i = 0; i < 10; ++i) stash_lane[i * 64] = pack_bf16(Pf[2 * i], Pf[2 * i + 1]);
        }
        ch_zero(acc);
        ch_stage<false>(r, ws, acc, xf);
        uint32_t ob[24];
        {
            float Pf[20];
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const uint32_t u = stash_lane[i * 64];
                Pf[2 * i] = bf16_lo(u);
                Pf[2 * i + 1] = bf16_hi(u);
            }
            const float Q0[5] = {Pf[0], Pf[1], Pf[2], Pf[3], Pf[4]}, Q1[5] = {Pf[5], Pf[6], Pf[7], Pf[8], Pf[9]},
                        Q2[5] = {Pf[10], Pf[11], Pf[12], Pf[13], Pf[14]}, Q3[5] = {Pf[15], Pf[16], Pf[17], Pf[18], Pf[19]};
            ch_attn_out<0>(acc, tab, g, hh, Q0, ob);
            ch_attn_out<1>(acc, tab, g, hh, Q1, ob);
            ch_attn_out<2>(acc, tab, g, hh, Q2, ob);
            ch_attn_out<3>(acc, tab, g, hh, Q3, ob);
        }
        // group g = the lane's values 48 g .. 48 g + 47 = k-steps 6 g .. 6 g + 5 of the out-projection
#define CH_PUT(G)                                                                                                \
    if (g == G) {                                                                                                \
        _Pragma("unroll") for (int i = 0; i < 6; ++i)                                                            \
            of[6 * G + i] = __builtin_bit_cast(bf16x8, u32x4{ob[4 * i], ob[4 * i + 1], ob[4 * i + 2], ob[4 * i + 3]}); \
    }
        CH_PUT(0) CH_PUT(1) CH_PUT(2) CH_PUT(3)
#undef CH_PUT
        __builtin_amdgcn_sched_barrier(0);
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
    f32x16 accb[3];                                          // second accumulator buffer: stage n + 1 runs while n's epilogue reads n's
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
