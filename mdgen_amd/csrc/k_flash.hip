// k_flash.hip -- streaming-softmax multi-head attention with RoPE'd operands, the learned bias key
// and key-padding mask, for one attention axis (temporal: 1001 keys; residue: 257 keys).
//
// Replaces mha.py:359-396: bmm(q,k^T) -> masked_fill(-inf) -> fp32 softmax -> bmm(p,v), without
// ever materialising the (bsz*16, len, len+1) score tensor (4.1 GB per layer at cfg-2 in the
// reference).  The head-averaged attention weights the reference computes and discards
// (mha.py:399-405) are not reproduced.
//
// Operand fragments are produced by k_ln_qkv in exactly the register layout the MFMAs want
// (DESIGN.md "fragment layout"), so this kernel issues only fully-coalesced 16-byte loads:
//   S^T[key][q] = K-frag (A) x Q-frag (B)      2 x mfma_32x32x16 (d = 24 padded to 32)
//   O^T[d][q]  += V^T-frag (A) x P^T (B)       2 x mfma_32x32x16, P^T = exp2(S^T - M) packed in place
// A lane owns one query column (q = lane&31) and half of the tile's keys.  One wave = one head x 64 queries (two
// 32-query tiles A, B); workgroup = 4 heads of the same queries; three workgroups per CU (<= 168 VGPRs).
// The loop is bound by the vector ALU (16 v_exp_f32 per (32-key, 32-query) pair cost about as much as its four
// MFMAs, profiles/r02_issue_rate.txt), so everything else a pair does on the VALU has been removed:
//   * the running shift -M rides in two spare K-dim slots of Q (a bf16 pair hi + lo against 1.0 in K), so the score
//     MFMA returns s - M;
//   * an all-ones V^T row (d = 24, a physical row of the fragment) makes the PV MFMA accumulate the denominator;
//   * NO row max and no range test in the common path.  P is bf16 -- an 8-bit exponent, the same range as fp32 --
//     so the shift only has to keep 2^(s - M) inside that range, not below 1.  M is anchored 63 above the row max of
//     the FIRST key tile (P <= 2^-63 there) and never moves: scores may rise 190 (log2 units) above that tile's max
//     before a P overflows, and an overflow cannot go unnoticed -- the denominator is the plain sum of all P (ones
//     row of V^T), so exponent-bit tests on the accumulators when the loop is done (inf / NaN anywhere, or a
//     denominator that underflowed) cover every P of the row.  If they fire
//     -- or if the first tile of the sequence is fully masked, so that there is nothing to anchor to -- the wave
//     discards its result and redoes the whole (head, 64 queries) with the ROBUST loop: true row max per tile, shift
//     re-anchored whenever the max moves up by more than 2^8, O rescaled.  Terms more than 63 below the anchor flush
//     to zero: they are < 2^-63 of the largest term of their row;
//   * software pipeline over (key tile, query tile) pairs A0 B0 A1 B1 ...: the block of a pair issues the score
//     MFMAs of the NEXT pair and the PV MFMAs of the PREVIOUS pair beside its own 16 exps -- every MFMA of a block
//     has its operands ready when the block starts, and the two neighbours belong to the other query tile;
//   * K / V^T are streamed with buffer loads (SGPR descriptor + constant per-lane offset + scalar tile offset: no
//     address arithmetic on the VALU), one tile ahead; V^T rows d > 24 are out-of-range lanes, which read zeros;
//   * per-tile key-validity words live in one VGPR (lane i = tile i of a 64-tile window), fetched with v_readlane;
//   * the learned bias key is an ordinary entry of the fragments (key slot len, written by k_ln_qkv).
#include "kernels.h"
#include "panel.h"
#include <type_traits>

namespace mdg {

__device__ __forceinline__ bf16x8 frag16(const unsigned char* p) { return *reinterpret_cast<const bf16x8*>(p); }

__device__ __forceinline__ bf16x8 frag8(const unsigned char* p) {  // 4 real bf16 + 4 zeros
    const u32x2 v = *reinterpret_cast<const u32x2*>(p);
    const u32x4 w = u32x4{v[0], v[1], 0u, 0u};
    return __builtin_bit_cast(bf16x8, w);
}

// K / V^T fragments of one 32-key tile.  They are fetched ahead with ORDINARY (compiler-counted) buffer loads, so
// hipcc inserts exact vmcnt waits, and pinned in place by sched_barrier(0) fences: without the fence the scheduler
// sinks each load to its first use and every tile pays a full L2 round trip.
struct KTile {
    u32x4 k0;   // K k-step 0: 8 bf16
    u32x4 k1;   // K k-step 1: 4 bf16, the constant 1.0 twice (slots 4, 5: they pick up -M = -(hi + lo) from Q), 2 zeros
};
struct VTile {
    u32x4 v0;   // V^T k-step 0
    u32x4 v1;   // V^T k-step 1
};
struct QTile {
    bf16x8 q0, q1;   // k-steps 0 / 1 of one 32-query tile; k-step-1 slots 4, 5 of the lanes hh == 0 carry -M
};
struct PTile {
    u32x4 p0, p1;   // P^T = exp2(S^T - M) of one (32-key, 32-query) tile as packed bf16, k-steps 0 / 1 of the PV MFMAs
};

// Running softmax state of ONE 32-query tile.  The row sum is NOT kept here: V^T row 24 is all ones, so the PV MFMA
// accumulates l = sum_k P[k][q] into O^T[24][q] (register 12 of the lanes with hh == 0) and every rescale of O
// rescales l with it.
struct Half {
    f32x16 o;
    float m;                     // applied shift M (exactly representable in bf16; rides in a spare K-dim slot of Q)
    unsigned long long unanch;   // wave-level lane mask (SGPR pair): lanes whose shift has NOT yet been anchored to a
                                 // finite score.  All zero after the first tile (or the first unmasked one)
};

__device__ __forceinline__ float half_max(float x) {   // max over the two half-waves (lane, lane^32)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

__device__ __forceinline__ float round_bf16(float x) { return __uint_as_float(pack_bf16(x, 0.f) << 16); }

// The shift M rides in two spare K-dim slots of Q as a bf16 pair hi + lo (16 significant bits: exact to < 1 log2 unit up
// to |M| = 2^16, where a single bf16 would already be off by hundreds); K holds 1.0 in both slots, so the score MFMA
// returns s - hi - lo.  Returns the shift actually applied.
__device__ __forceinline__ float set_shift(QTile& q, float m, int hh) {
    const float hi = round_bf16(m), lo = round_bf16(m - hi);
    if (hh == 0) {
        q.q1[4] = (__bf16)(-hi);
        q.q1[5] = (__bf16)(-lo);
    }
    return hi + lo;
}

// The anchor sits kAnchor (log2 units) above the row max it was taken from: P = 2^(s - M) <= 2^-kAnchor there, scores
// up to 127 + kAnchor above that max still give a finite P, terms more than 126 - kAnchor below it flush to zero.
constexpr float kAnchor = 63.f;

// Key-padding mask of one tile: `vm` = validity bits of its 32 keys (wave-uniform, in an SGPR).  Only ever runs for
// the last tile of a sequence and for padded residues; the shift is asm volatile so that none of it is hoisted.
__device__ __forceinline__ void mask_scores(f32x16& s, uint32_t vm, int hh) {
    uint32_t vmh;
    asm volatile("v_lshrrev_b32 %0, %1, %2" : "=v"(vmh) : "v"(4 * hh), "s"(vm));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const bool ok = (vmh >> ((r & 3) + 8 * (r >> 2))) & 1u;
        s[r] = ok ? s[r] : -1e30f;
    }
}

__device__ __forceinline__ f32x16 scores(const KTile& k, const QTile& q) {
    const bf16x8 k0 = __builtin_bit_cast(bf16x8, k.k0), k1 = __builtin_bit_cast(bf16x8, k.k1);
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q.q0, zero, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q.q1, s, 0, 0, 0);
    asm volatile("" :: "v"(k0), "v"(k1), "v"(q.q0), "v"(q.q1));   // sources outlive the MFMAs (see block())
    return s;
}

// The MFMA-dense block of one (key tile, query tile) pair, one straight-line region in which every MFMA has its
// operands ready at entry:
//   od += V^T P^T of the PREVIOUS pair            2 MFMAs (that pair belongs to the OTHER query tile, so a shift
//                                                 moved by this pair's slow path never concerns it)
//   n  = S^T - M of the FOLLOWING pair            K-frag (A) x Q-frag (B), 2 MFMAs
//   pc = exp2(s) of THIS pair, packed to bf16     16 v_exp + 8 v_cvt_pk
// i.e. 4 MFMAs (128 matrix-pipe cycles) beside 24 VALU operations, none waiting for another.  The PV MFMAs go
// first: once they have been issued the previous pair's P registers are dead, and the packed results of this pair
// can take their place (one P tuple live instead of two).
// The score MFMAs take the INLINE CONSTANT 0 as C (no 16-register zero tuple to keep alive).  With that operand form
// hipcc does not mark the destination early-clobber and may allocate it on top of a source that dies at the MFMA --
// which corrupts results.  Both sources are therefore kept alive across the MFMAs: Q is loop
// invariant, K gets a no-op use after the block.  (build.py check_isa still rejects any overlap in the final ISA.)
__device__ __forceinline__ void block(const KTile& kn, const QTile& qn, f32x16& n,
                                      const VTile& vp, const PTile& pp, f32x16& od,
                                      const f32x16& s, PTile& pc) {
    const bf16x8 k0 = __builtin_bit_cast(bf16x8, kn.k0), k1 = __builtin_bit_cast(bf16x8, kn.k1);
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    od = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vp.v0), __builtin_bit_cast(bf16x8, pp.p0), od, 0, 0, 0);
    n = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qn.q0, zero, 0, 0, 0);
    od = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vp.v1), __builtin_bit_cast(bf16x8, pp.p1), od, 0, 0, 0);
    n = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qn.q1, n, 0, 0, 0);
    float e[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) e[j] = __builtin_amdgcn_exp2f(s[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#ifdef MDGEN_DEV_FLASH_TRUNC   // (experiment) P truncated to bf16 by one v_perm_b32 per pair instead of v_cvt_pk_bf16_f32: the softmax
                               // denominator is the sum of the SAME truncated P (ones row of V^T), so the -1/2 ulp bias cancels to first order
        pc.p0[j] = __builtin_amdgcn_perm(__float_as_uint(e[2 * j + 1]), __float_as_uint(e[2 * j]), 0x07060302u);
        pc.p1[j] = __builtin_amdgcn_perm(__float_as_uint(e[9 + 2 * j]), __float_as_uint(e[8 + 2 * j]), 0x07060302u);
#else
        pc.p0[j] = pack_bf16(e[2 * j], e[2 * j + 1]);
        pc.p1[j] = pack_bf16(e[8 + 2 * j], e[9 + 2 * j]);
#endif
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x402, 6, 0);   // six VALU / transcendental
    }
    // sources outlive the MFMAs (see above); the packed P is pinned HERE, inside the block: left alone hipcc sinks the
    // exps below the block (next to their consumer, the next block's PV MFMA) and the overlap is lost
    asm volatile("" : "+v"(pc.p0), "+v"(pc.p1) : "v"(k0), "v"(k1), "v"(qn.q0), "v"(qn.q1));
}

// How far the running shift may lag the true row max before O is rescaled (log2 units: P <= 2^kDefer).
constexpr float kDefer = 8.f;

// ROBUST loop only.  Softmax bookkeeping of one (key tile, query tile) pair, everything except exp: key-padding mask,
// row max, and -- rarely -- moving the shift (which rewrites Q's -m slot and rescales O).  `vm`: validity bits of the tile's 32
// keys (wave-uniform, in an SGPR).
__device__ __forceinline__ void softmax_stats(Half& st, f32x16& s, QTile& q, uint32_t vm, int hh) {
    if (vm != 0xffffffffu) mask_scores(s, vm, hh);
    float t = fmaxf(s[0], s[1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) t = fmaxf(fmaxf(t, s[r]), s[r + 1]);   // v_max3_f32
    t = half_max(t);
    // one scalar test per pair: some lane's max ran ahead of its shift, or some lane is still unanchored
    if ((__builtin_amdgcn_ballot_w64(t > kDefer) | st.unanch) != 0) {   // wave-uniform and rare after the first tiles
        // Re-anchor when the shift lags the tile max by more than kDefer, or (first finite tile of a query only)
        // leads it by more than kDefer.  Both half-waves of a query see the same t -> same decision.
        const bool fin = t > -1e29f;
        const bool un = (st.unanch >> lane_id()) & 1ull;
        const bool mv = (t > kDefer) | (un & fin & (t < -kDefer));
        st.unanch &= ~__builtin_amdgcn_ballot_w64(fin);
        // new shift = m + tile max (as a bf16 pair); e = what this tile's already-shifted scores still have to lose
        const float r = set_shift(q, mv ? st.m + t : st.m, hh);
        const float e = r - st.m;
        st.m = r;
        // e < 0 only while nothing has been accumulated yet (first anchoring): O is still zero, keep it so
        const float a = __builtin_amdgcn_exp2f(fminf(-e, 0.f));
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            st.o[i] *= a;
            s[i] -= e;
        }
    }
}

#ifdef MDGEN_DEV_FLASH_STAMPS   // (experiments only) per-wave s_memtime / s_memrealtime stamps: [wave][16]
__device__ unsigned long long g_flash_stamps[32768 * 16];
extern "C" int mdgen_dev_flash_stamps(void* host, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_flash_stamps), bytes);
}
#define FLASH_STAMP(slot, v)                                                                          \
    if (lane == 0 && (long)blockIdx.x * 4 + w < 32768) g_flash_stamps[((long)blockIdx.x * 4 + w) * 16 + (slot)] = (v)
// k_flash_proj: [wave][16]: 0 start, 1..4 after head group 0..3, 5 after the barrier, 6 after the GEMM, 7 end (s_memtime);
// 8 / 9 s_memrealtime at start / end; 10 HW_ID
__device__ unsigned long long g_fproj_stamps[8192 * 16];
extern "C" int mdgen_dev_fproj_stamps(void* host, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fproj_stamps), bytes);
}
#define FPROJ_STAMP(slot, v)                                                                          \
    if (lane_id() == 0 && (long)blockIdx.x * 4 + w < 8192) g_fproj_stamps[((long)blockIdx.x * 4 + w) * 16 + (slot)] = (v)
#define FLASH_STAMP_ROW(w, hg) ((w) + 4096 * (hg))   // k_flash_proj: flash_job's own stamps of head group hg go to rows 4096 hg + ...
#else
#define FLASH_STAMP(slot, v)
#define FPROJ_STAMP(slot, v)
#define FLASH_STAMP_ROW(w, hg) (w)
#endif

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., so that h[j], q[j], s[j & 1] are constant-indexed and stay
// in registers (a runtime-looking index inside the lambdas sent the whole state to scratch)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// Output policies of flash_job.  Global: bf16 rows [token][384] (k_proj<0>'s A operand).  Only wave-uniform members: a per-lane
// address held across the loop would cost two VGPRs of a budget (168) that has none to spare.
struct FlashStoreGlobal {
    __bf16* obuf;
    AxisMap ax;
    int seq, head;
    __device__ __forceinline__ void row(int, int pos, u32x2 a, u32x2 b, u32x2 c) const {
        u32x2* d = reinterpret_cast<u32x2*>(obuf + ax.token(seq, pos) * kC + head * kDH + (lane_id() >> 5) * 12);
        d[0] = a;
        d[1] = b;
        d[2] = c;
    }
    __device__ __forceinline__ void pad(int, int) const {}
};

// NQ = 32-query tiles per wave.  NQ = 2 (three waves per SIMD) or NQ = 4 (two waves per SIMD, <= 256 VGPRs): with four
// tiles a K / V^T fragment load and a wave's prologue serve twice as many (key tile, query tile) pairs.
//
// flash_job: ONE wave's whole job -- head `head` of sequence `seq`, queries 32 NQ qc .. 32 NQ (qc + 1) - 1 -- shared by
// k_flash (one job per wave, output rows to HBM) and k_flash_proj (four jobs per wave, output rows into the LDS panel of
// the out-projection).  `store.row(j, pos, a, b, c)`: the 12 bf16 output features 12 hh .. 12 hh + 11 of this head (three
// 8-byte pieces) for query tile j, sequence position pos; `store.pad(j, pos)`: a row past the end of the sequence.
// What a job reads before its loop: requested by flash_prefetch, consumed by flash_job.  k_flash_proj requests the NEXT head's set
// before it starts the current head's loop, so that a job's two serial memory round trips (Q / first K tile, then the loop's
// first tiles) shrink to one (ATLAS: 8-tile sequences, where the prologue was ~40 % of a job).
template <int NQ>
struct FlashPre {
    QTile q[NQ];     // Q fragments of q-tiles NQ qc .. NQ qc + NQ - 1 (tiles past the end of the sequence reuse the first one)
    u32x4 kb0;       // the learned bias key's K slot (len a multiple of 32: it sits alone in the last tile, see flash_job)
    u32x2 kb1;
    f32x4 bvl[3];    // this lane half's 12 features of the learned bias value
    KTile k0t;       // K tile 0 (the anchor)
    uint32_t vmw;    // key-validity words of tiles 0..63 (lane i = tile i)
};

__device__ __forceinline__ void* flash_uniform_ptr(const unsigned char* q_) {
    const unsigned long long a = (unsigned long long)q_;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (void*)(((unsigned long long)hi << 32) | lo);
}

template <int NQ>
__device__ __forceinline__ void flash_prefetch(const FlashParams p, const int seq, const int head, const int qc, FlashPre<NQ>& pre) {
    const int lane = lane_id(), hh = lane >> 5;
    const int len = p.ax.len, nt = p.ax.ntile();
    const long ftile = (long)(seq * kH + head) * nt;   // first fragment tile of this (sequence, head)
    const unsigned char* qb = p.qf + ftile * kFragQ;
    const int qt0 = NQ * qc;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int qt = (qt0 + j) * 32 < len ? qt0 + j : qt0;
        pre.q[j].q0 = frag16(qb + (long)qt * kFragQ + lane * 16);
        pre.q[j].q1 = frag8(qb + (long)qt * kFragQ + 1024 + lane * 8);
    }
    // len a multiple of 32: the learned bias key sits ALONE in the last tile and is handled as a rank-1 term; its K slot and
    // value row are requested here, with Q, so that their round trip is hidden (requested at their use they cost more than
    // the tile they save)
    const bool bias_alone = (len & 31) == 0 && len >= 32;
    const unsigned char* kb = p.kf + (ftile + (bias_alone ? (len >> 5) : 0)) * kFragK + hh * 32 * 16;   // key slot 0 of this lane half
    pre.kb0 = *reinterpret_cast<const u32x4*>(kb);
    pre.kb1 = *reinterpret_cast<const u32x2*>(kb + 1024);
    const f32x4* bv = reinterpret_cast<const f32x4*>(p.bias_v + head * kDH + hh * 12);
    pre.bvl[0] = bv[0];
    pre.bvl[1] = bv[1];
    pre.bvl[2] = bv[2];
    // K tile 0 is wanted first (the anchor): requested together with Q, so that its round trip overlaps the validity-word read
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(flash_uniform_ptr(p.kf + ftile * kFragK), 0, 0x7fffffff, 0x00020000);
    pre.k0t.k0 = __builtin_amdgcn_raw_buffer_load_b128(krs, lane * 16, 0, 0);
    pre.k0t.k1 = __builtin_amdgcn_raw_buffer_load_b128(krs, lane * 16 + 1024, 0, 0);
    // per-tile key-validity words (key-padding mask; the bias key at position len is always valid; positions > len are
    // invalid): written by k_ln_qkv, read 64 tiles at a time into ONE register (lane i = tile i of the window) and picked out
    // with v_readlane.  No LDS, no barrier: the four waves of a workgroup never meet in the loop.
    pre.vmw = (p.vmask + (long)seq * p.vmask_stride)[lane];
}

template <int NQ, class Store>
__device__ __forceinline__ void flash_job(const FlashParams p, const int seq, const int head, const int qc, const int w,
                                          const FlashPre<NQ> pre, const Store store) {   // (by value: through a reference hipcc spills 15 registers)
    const int lane = lane_id(), hh = lane >> 5, ql = lane & 31;
    FLASH_STAMP(0, __builtin_amdgcn_s_memtime());
    FLASH_STAMP(4, __builtin_amdgcn_s_memrealtime());
    const int len = p.ax.len, nt = p.ax.ntile();   // tiles per (seq, head): they cover the len keys + the bias key
    const long ftile = (long)(seq * kH + head) * nt;   // first fragment tile of this (sequence, head)
    const int qt0 = NQ * qc;
    QTile q[NQ];
    bool qvalid[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        qvalid[j] = (qt0 + j) * 32 < len;
        q[j] = pre.q[j];
    }
    const bool bias_alone = (len & 31) == 0 && len >= 32;
    const u32x4 kb0 = pre.kb0;
    const u32x2 kb1 = pre.kb1;
    const f32x4 bvl[3] = {pre.bvl[0], pre.bvl[1], pre.bvl[2]};

    // ---- K / V^T streams: one buffer descriptor each (wave-uniform base = this (sequence, head)'s first tile), a
    // constant per-lane byte offset, and the tile offset as the scalar offset of the load.  V^T: rows d <= 24 are
    // lanes of the fragment (row 24 = ones); the lanes of rows d > 24 point far out of range and read zeros.
    // Loads are UNCONDITIONAL: tiles past the end of this (sequence, head) read whatever follows in the fragment
    // buffer (finite bf16 of the next head, or the tail) and their results are never used.
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(flash_uniform_ptr(p.kf + ftile * kFragK), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(flash_uniform_ptr(p.vf + ftile * kFragV), 0, 0x7fffffff, 0x00020000);
    const int koff = lane * 16;
    const int voff = ql <= kDH ? hh * 400 + ql * 16 : (int)0x80000000;
    auto issue_k = [&](KTile& t, int kt) __attribute__((always_inline)) {
        t.k0 = __builtin_amdgcn_raw_buffer_load_b128(krs, koff, kt * kFragK, 0);
        t.k1 = __builtin_amdgcn_raw_buffer_load_b128(krs, koff + 1024, kt * kFragK, 0);
    };
    auto issue_v = [&](VTile& t, int kt) __attribute__((always_inline)) {
        t.v0 = __builtin_amdgcn_raw_buffer_load_b128(vrs, voff, kt * kFragV, 0);
        t.v1 = __builtin_amdgcn_raw_buffer_load_b128(vrs, voff + 800, kt * kFragV, 0);
    };
    const KTile k0t = pre.k0t;
    const uint32_t* vmrow = p.vmask + (long)seq * p.vmask_stride;
    uint32_t vmw = pre.vmw;

    // One key tile = NQ (key tile, query tile) pairs, query tiles 0 .. NQ-1 in turn.  Pipeline: the block of pair i
    // issues the score MFMAs of pair i+1 and the PV MFMAs of pair i-1 beside its own exps.  Two score tuples and two P
    // tuples alternate by pair parity (NQ is even, so a query tile always meets the same ones); K / V tiles alternate
    // between two slots by tile parity (kx, vx: even tiles; ky, vy: odd tiles).
    Half h[NQ];
    f32x16 s[2];
    PTile pt[2];
    KTile kx, ky;
    VTile vx, vy;
    // (always_inline: a lambda that is NOT inlined keeps everything it captures by reference in scratch memory)
    // Walk order: step i of a run over n tiles visits tile phys(i) = (i + rot) mod n.  The fixed-anchor loop starts every 64-query
    // chunk of a sequence at a different tile (rot = qc n / nqc): the q-chunks of a (sequence, head) start together and stream the
    // same K / V^T fragments, and in step they all sit behind ONE chain of HBM misses (the leader's, one per tile: the stamps of
    // k_flash_proj showed the first job of a launch's first round at 1.7x the time of a warm one); staggered, the 16 chunks pull
    // the whole fragment set into L2 at once and each misses on a sixteenth of it.  A sum over all keys does not care about the
    // order (the shift is fixed before the loop); the robust loop keeps the natural order (rot = 0), as do sequences of more than
    // 64 tiles (the validity words arrive in 64-tile windows).
    int rot = 0, nrun = 1;
    auto phys = [&](int i) __attribute__((always_inline)) {
        const int t = i + rot;
        return t >= nrun ? t - nrun : t;
    };
    auto tile = [&](auto robust, int ti, KTile& kc, KTile& kn, VTile& vc, VTile& vprev) __attribute__((always_inline)) {
        constexpr bool kRobust = decltype(robust)::value;
        const int t = kRobust ? ti : phys(ti);
        const uint32_t vm = __builtin_amdgcn_readlane(vmw, t & 63);
        static_for<NQ>([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            // pair (t, j): scores in s[j & 1].  Block: s[(j+1) & 1] <- scores of the next pair; O of the previous
            // pair's query tile += its V P; pt[j & 1] <- exp(s[j & 1])
            if constexpr (kRobust) softmax_stats(h[j], s[j & 1], q[j], vm, hh);
            else if (vm != 0xffffffffu) mask_scores(s[j & 1], vm, hh);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (j == 0) block(kc, q[1], s[1], vprev, pt[(NQ - 1) & 1], h[NQ - 1].o, s[0], pt[0]);
            else if constexpr (j < NQ - 1) block(kc, q[j + 1], s[(j + 1) & 1], vc, pt[(j - 1) & 1], h[j - 1].o, s[j & 1], pt[j & 1]);
            else block(kn, q[0], s[0], vc, pt[(j - 1) & 1], h[j - 1].o, s[j & 1], pt[j & 1]);
            __builtin_amdgcn_sched_barrier(0);
#ifndef MDGEN_DEV_FLASH_NOLOAD   // (timing experiments only: scripts/micro/flash_variants.sh)
            if constexpr (j == 0) issue_v(vprev, kRobust ? ti + 1 : phys(ti + 1));      // V(t-1) is consumed: refill its slot
            if constexpr (j == NQ - 2) issue_k(kc, kRobust ? ti + 2 : phys(ti + 2));    // K(t) has served its last query tile
            if constexpr (j == 0 || j == NQ - 2) __builtin_amdgcn_sched_barrier(0);
#endif
        });
    };
    // The whole (head, 32 NQ queries) job over key tiles 0 .. nt - 1; returns with the last pair's PV MFMAs issued.
    auto run = [&](auto robust, const int nt, const int rot_) __attribute__((always_inline)) {
        rot = rot_;
        nrun = nt;
#pragma unroll
        for (int j = 0; j < NQ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[j].o[r] = opaque_zero();
        pt[(NQ - 1) & 1].p0 = pt[(NQ - 1) & 1].p1 = u32x4{0u, 0u, 0u, 0u};   // "previous pair" of the very first block: P = 0
        if (nt > 64) vmw = vmrow[lane];          // (a re-run after a long fast loop: back to the first window)
        issue_k(kx, phys(0));
        issue_v(vy, phys(0));    // stands in for "V(-1)": any finite values (its P is zero)
        issue_k(ky, phys(1));
        issue_v(vx, phys(0));
        __builtin_amdgcn_sched_barrier(0);
        s[0] = scores(kx, q[0]);   // pair (0, 0)
        // The loop body is a PAIR of tiles with a single exit (an exit between the two makes hipcc keep O in
        // different registers in the two halves and copy it mid-chain); an odd tile count gets a tail tile.
        int t = 0;
        for (; t + 1 < nt; t += 2) {
            if ((t & 63) == 0 && t) vmw = vmrow[t + lane];
            tile(robust, t, kx, ky, vx, vy);
            tile(robust, t + 1, ky, kx, vy, vx);
        }
        const bf16x8 lp0 = __builtin_bit_cast(bf16x8, pt[(NQ - 1) & 1].p0), lp1 = __builtin_bit_cast(bf16x8, pt[(NQ - 1) & 1].p1);
        if (t < nt) {
            if ((t & 63) == 0 && t) vmw = vmrow[t + lane];
            tile(robust, t, kx, ky, vx, vy);
            // pending: P of the last tile's last pair with V(last) = vx
            h[NQ - 1].o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vx.v0), __builtin_bit_cast(bf16x8, pt[(NQ - 1) & 1].p0), h[NQ - 1].o, 0, 0, 0);
            h[NQ - 1].o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vx.v1), __builtin_bit_cast(bf16x8, pt[(NQ - 1) & 1].p1), h[NQ - 1].o, 0, 0, 0);
        } else {
            h[NQ - 1].o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vy.v0), lp0, h[NQ - 1].o, 0, 0, 0);
            h[NQ - 1].o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vy.v1), lp1, h[NQ - 1].o, 0, 0, 0);
        }
    };

    // ---- a sequence with NO valid key but the learned bias key (a padded residue's temporal sequence: 6 % of the waves of
    //      the ATLAS config, which used to take the robust loop to compute a softmax over one key): the output of every
    //      query is exactly bias_v (rounded to bf16 as the V^T fragment holds it).  Written directly.
    if (nt <= 64 && !p.force_robust) {
        const uint32_t want = lane == (len >> 5) ? 1u << (len & 31) : 0u;
        const bool only_bias = __builtin_amdgcn_ballot_w64(lane < nt && vmw != want) == 0;
        if (only_bias) {
            const f32x4 b0 = bvl[0], b1 = bvl[1], b2 = bvl[2];
            const u32x2 d0 = {pack_bf16(b0[0], b0[1]), pack_bf16(b0[2], b0[3])}, d1 = {pack_bf16(b1[0], b1[1]), pack_bf16(b1[2], b1[3])},
                        d2 = {pack_bf16(b2[0], b2[1]), pack_bf16(b2[2], b2[3])};
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const int pos = (qt0 + j) * 32 + ql;
                if (qvalid[j] && pos < len) store.row(j, pos, d0, d1, d2);
                else store.pad(j, pos);
            }
            return;
        }
    }
    // ---- len a multiple of 32: the bias key sits ALONE in the last tile (key slot 0 of tile len / 32).  The fixed-anchor
    //      loop then skips that tile (one of nine at L = 256) and adds the key as a rank-1 term afterwards; its score is the
    //      plain dot product of this lane's q values with the key's (both bf16, as the MFMA sees them).  The robust loop
    //      keeps walking all nt tiles.
    const int nt_fast = bias_alone ? nt - 1 : nt;
    float sbias[NQ], vbias[12];
    if (bias_alone) {
        const f32x4 b0 = bvl[0], b1 = bvl[1], b2 = bvl[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            vbias[i] = round_bf16(b0[i]);
            vbias[4 + i] = round_bf16(b1[i]);
            vbias[8 + i] = round_bf16(b2[i]);
        }
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const u32x4 qa = __builtin_bit_cast(u32x4, q[j].q0);
            const u32x4 qb_ = __builtin_bit_cast(u32x4, q[j].q1);   // elements 0..3 real, 4..7 the -M slots / zeros: not used
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) d += bf16_lo(qa[i]) * bf16_lo(kb0[i]) + bf16_hi(qa[i]) * bf16_hi(kb0[i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) d += bf16_lo(qb_[i]) * bf16_lo(kb1[i]) + bf16_hi(qb_[i]) * bf16_hi(kb1[i]);
            sbias[j] = d + __shfl_xor(d, 32, 64);
        }
    }
    // ---- FAST loop: anchor every query tile kAnchor above the row max of key tile 0 and never move the shift.
    bool robust_needed = p.force_robust != 0;
    {
        const uint32_t vm0 = __builtin_amdgcn_readlane(vmw, 0);
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            f32x16 s0 = scores(k0t, q[j]);   // the -M slots are still zero: plain scores
            if (vm0 != 0xffffffffu) mask_scores(s0, vm0, hh);
            float t0 = fmaxf(s0[0], s0[1]);
#pragma unroll
            for (int r = 2; r < 16; r += 2) t0 = fmaxf(fmaxf(t0, s0[r]), s0[r + 1]);
            t0 = half_max(t0);
            // nothing to anchor to (all of tile 0 masked: a padded residue's temporal sequence) -> robust loop
            robust_needed |= __builtin_amdgcn_ballot_w64(!(t0 > -1e29f)) != 0;
            h[j].m = set_shift(q[j], t0 + kAnchor, hh);
        }
    }
    FLASH_STAMP(1, __builtin_amdgcn_s_memtime());
    FLASH_STAMP(6, (unsigned long long)robust_needed);
    FLASH_STAMP(8, (unsigned long long)__float_as_uint(h[0].m));   // lane 0's fixed anchor (first-tile row max + kAnchor)
    if (!robust_needed) {
        run(std::false_type{}, nt_fast, (p.rotate && nt_fast > 2 && nt <= 64) ? (qc * nt_fast) / ((len + 32 * NQ - 1) / (32 * NQ)) : 0);
        if (bias_alone) {
            // the learned bias key as a rank-1 term: P_b = 2^(q . k_b - M), O += P_b v_b, denominator += P_b
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const float pb = __builtin_amdgcn_exp2f(sbias[j] - h[j].m);
#pragma unroll
                for (int r = 0; r < 12; ++r) h[j].o[r] += pb * vbias[r];
                if (hh == 0) h[j].o[12] += pb;
            }
        }
        FLASH_STAMP(2, __builtin_amdgcn_s_memtime());
        // every P of a query row is summed into its denominator (register 12 of the lanes hh == 0): finite and
        // positive <=> no P overflowed.  Bit tests, not float compares: this file is built with -fno-honor-nans.
        bool bad = false;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const uint32_t lj = __float_as_uint(h[j].o[12]);
            bad |= (lj & 0x7f800000u) == 0x7f800000u;   // inf or NaN
            // a denominator that underflowed to nothing (first-tile max far above everything the bf16-pair shift
            // resolves); lanes hh == 1 hold row 28 there, a zero padding row: not a denominator
            bad |= hh == 0 && (lj & 0x7f800000u) == 0u;
            // ... and a P below the overflow threshold can still push a sum of P v over it: every accumulator is checked
#pragma unroll
            for (int i = 0; i < 16; ++i) bad |= (__float_as_uint(h[j].o[i]) & 0x7f800000u) == 0x7f800000u;
        }
        FLASH_STAMP(9, (unsigned long long)__builtin_amdgcn_ballot_w64(bad));
        robust_needed = __builtin_amdgcn_ballot_w64(bad) != 0;
#ifdef MDGEN_DEV_FLASH_NOFALLBACK   // (experiments only: shows what the fixed anchor alone does to overflowing scores)
        robust_needed = false;
#endif
    }
    if (robust_needed) {   // wave-uniform; start over with a moving shift
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            h[j].m = 0.f;
            h[j].unanch = ~0ull;
            set_shift(q[j], 0.f, hh);
        }
        run(std::true_type{}, nt, 0);
    }
    FLASH_STAMP(3, __builtin_amdgcn_s_memtime());
    FLASH_STAMP(11, (unsigned long long)__float_as_uint(h[0].m));   // lane 0's final shift (robust loop: ~ the true row max)
#ifdef MDGEN_DEV_FLASH_STAMPS
    if (lane == 0 && (long)blockIdx.x * 4 + w < 32768) g_flash_stamps[((long)blockIdx.x * 4 + w) * 16 + 6] |= (unsigned long long)robust_needed << 1;
#endif
    // ---- epilogue: O^T row 24 = softmax denominator (held by lanes hh == 0); registers 0..11 of lane-half hh are
    //      features 12 hh .. 12 hh + 11 of this head
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const float l = __shfl(h[j].o[12], ql, 64);
        const int pos = (qt0 + j) * 32 + ql;
        if (qvalid[j] && pos < len) {
            const float inv = 1.0f / l;
            const f32x16 o = h[j].o;
            store.row(j, pos, u32x2{pack_bf16(o[0] * inv, o[1] * inv), pack_bf16(o[2] * inv, o[3] * inv)},
                      u32x2{pack_bf16(o[4] * inv, o[5] * inv), pack_bf16(o[6] * inv, o[7] * inv)},
                      u32x2{pack_bf16(o[8] * inv, o[9] * inv), pack_bf16(o[10] * inv, o[11] * inv)});
        } else {
            store.pad(j, pos);
        }
    }
    FLASH_STAMP(5, __builtin_amdgcn_s_memrealtime());
}

template <int NQ>
__global__ __launch_bounds__(256, NQ == 2 ? 3 : 2) void k_flash(const FlashParams p) {
    const int w = __builtin_amdgcn_readfirstlane(wave_id());   // wave-uniform, and the compiler knows it
    const int nqc = (p.ax.len + 32 * NQ - 1) / (32 * NQ);
    // (sequence, head group) pairs are dealt to the 8 XCDs (block b runs on XCD b % 8) so that ALL q-chunks of a
    // pair -- which stream the same K/V -- share one L2: K/V leave HBM once.
    const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int qc = rest % nqc, pair = (rest / nqc) * 8 + xcd;
    const int seq = pair >> 2, hg = pair & 3;
    if (seq >= p.ax.nseq) return;
    const int head = hg * 4 + w;
    FlashPre<NQ> pre;
    flash_prefetch<NQ>(p, seq, head, qc, pre);
    flash_job<NQ>(p, seq, head, qc, w, pre, FlashStoreGlobal{p.obuf, p.ax, seq, head});
}

// =================================================================================================
// k_flash_proj: the attention of ALL 16 heads for 64 queries of one sequence, then the sub-layer's out-projection + gated
// residual (mha.py:389-397, latent_model.py:462,476) -- k_flash + k_proj<0> in one launch (round 5).
//
// A workgroup owns 64 consecutive positions of one sequence.  Its four waves run flash_job four times (head groups 0..3, wave w
// = head 4 hg + w) and drop the normalised output rows into a bf16 LDS panel [64][384] in the swizzled A-operand layout of the
// resident-panel GEMM family (panel.h) instead of HBM; after one barrier the panel is multiplied with W_o (wave w: features
// 96 w .. 96 w + 95, 144 MFMAs) and the result is added to the residual stream through the staged 16-byte read-modify-write
// epilogue k_proj<0> uses.  What it removes per launch at cfg-2: the 49 MB write and read of the attention output, 98 MB of
// the residual stream's traffic moved out of a kernel that did nothing else (k_proj<0> ran at its HBM roof, 55 us) into the
// shadow of other workgroups' VALU-bound attention loops, and a launch boundary.  Workgroups of a sequence go to one XCD (they
// stream the same K / V^T fragments).  Two workgroups per CU (227 registers: the next head's Q / first K tile / bias slots are
// requested before the current head's loop starts).  Measured and dropped (profiles/r05_experiments.txt): a 168-register build for
// three workgroups per CU (1024 jobs on 768 slots: the second, one-third-full round costs more than the denser loop gains), the
// residual rows of the whole panel requested up front (epilogue 16.8k -> 23.2k cycles: the HBM queue, not the round trips, paces it).
// =================================================================================================
typedef __attribute__((address_space(3))) unsigned char lds_byte;   // (an LDS pointer inside a struct must keep its address
                                                                    // space, or every store becomes a flat_store: DESIGN 6.25)
struct FlashStorePanel {
    lds_byte* panel;
    int head;
    __device__ __forceinline__ void put(int j, u32x2 a, u32x2 b, u32x2 c) const {
        const int lane = lane_id(), row = 32 * j + (lane & 31), byte = head * 48 + (lane >> 5) * 24;
        typedef __attribute__((address_space(3))) u32x2 lds_u32x2;
        *reinterpret_cast<lds_u32x2*>(panel + panel_off(row, byte, kC * 2)) = a;
        *reinterpret_cast<lds_u32x2*>(panel + panel_off(row, byte + 8, kC * 2)) = b;
        *reinterpret_cast<lds_u32x2*>(panel + panel_off(row, byte + 16, kC * 2)) = c;
    }
    __device__ __forceinline__ void row(int j, int, u32x2 a, u32x2 b, u32x2 c) const { put(j, a, b, c); }
    // rows past the end of the sequence enter the GEMM as zeros (their results are never stored)
    __device__ __forceinline__ void pad(int j, int) const { put(j, u32x2{0u, 0u}, u32x2{0u, 0u}, u32x2{0u, 0u}); }
};

__global__ __launch_bounds__(256, 2) void k_flash_proj(const FlashProjParams p) {
    constexpr int NQ = 2;
    static_assert(32 * NQ == kPanel, "one workgroup = one 64-row panel");
    __shared__ __attribute__((aligned(16))) unsigned char smem[sizeof(PanelRows) + kPanel * kC * 2];
    PanelRows* pr = reinterpret_cast<PanelRows*>(smem);
    unsigned char* panel = smem + sizeof(PanelRows);
    const int w = __builtin_amdgcn_readfirstlane(wave_id());
    const int nqc = (p.f.ax.len + kPanel - 1) / kPanel;
    const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int qc = rest % nqc, seq = (rest / nqc) * 8 + xcd;
    if (seq >= p.f.ax.nseq) return;
    FPROJ_STAMP(0, __builtin_amdgcn_s_memtime());
    FPROJ_STAMP(8, __builtin_amdgcn_s_memrealtime());
    FPROJ_STAMP(10, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4));   // HW_REG_HW_ID
    setup_rows_axis(pr, p.f.ax, seq, qc * kPanel, p.mm);   // read after the barrier below
    FlashPre<NQ> cur;
    flash_prefetch<NQ>(p.f, seq, w, qc, cur);
#pragma unroll 1
    for (int hg = 0; hg < 4; ++hg) {
        FlashPre<NQ> nxt;
        flash_prefetch<NQ>(p.f, seq, 4 * (hg < 3 ? hg + 1 : 3) + w, qc, nxt);   // in flight while this head's job runs
        __builtin_amdgcn_sched_barrier(0);
        flash_job<NQ>(p.f, seq, 4 * hg + w, qc, FLASH_STAMP_ROW(w, hg), cur, FlashStorePanel{(lds_byte*)panel, 4 * hg + w});
        cur = nxt;
        FPROJ_STAMP(1 + hg, __builtin_amdgcn_s_memtime());
    }
    __syncthreads();   // the attention output of all 16 heads is in the panel
    FPROJ_STAMP(5, __builtin_amdgcn_s_memtime());
    const int lane = lane_id();
    f32x16 acc[6];
#ifndef MDGEN_DEV_FLASH_NOEARLY   // as k_flash_proj8: the first batch of residual rows requested ahead of the GEMM (ATLAS +0.8 % end to end,
                                  // profiles/r06_experiments.txt #3)
    EpiPre<8> ep;
    epi_rmw_request<8>(0, 0, pr, 96 * w, p.h, ep);
    __builtin_amdgcn_sched_barrier(0);
#endif
    zero_acc<6>(acc);
    wave_gemm<2, 3, 24, false>(panel, kC * 2, 0, 0, p.wo + (size_t)(3 * w) * 24 * 64 + lane, 24 * 64, acc);
    FPROJ_STAMP(6, __builtin_amdgcn_s_memtime());
    __syncthreads();   // every wave is done reading the panel: reuse it as four 12 KiB staging slabs
#ifndef MDGEN_DEV_FLASH_NOEARLY
    epilogue_gate_residual_lds_pre<3>(acc, pr, reinterpret_cast<float*>(panel) + w * (32 * 96), 96 * w, p.bo, p.mm, p.gate_chunk, true,
                                      p.h, ep);
#else
    epilogue_gate_residual_lds<3>(acc, pr, reinterpret_cast<float*>(panel) + w * (32 * 96), 96 * w, p.bo, p.mm, p.gate_chunk, true,
                                  p.h);
#endif
    FPROJ_STAMP(7, __builtin_amdgcn_s_memtime());
    FPROJ_STAMP(9, __builtin_amdgcn_s_memrealtime());
}

// =================================================================================================
// k_flash_proj8: the same fusion with four query tiles per wave.  The loop of either form runs at 260-290 cycles per (32-key,
// 32-query) pair per SIMD -- its bound is the VALU issue port (16 quarter-rate exponentials + 8 conversions per pair: ~200 cycles
// per SIMD at any occupancy, profiles/r05_experiments.txt 17), not the number of waves -- so what this form buys is elsewhere: the
// K / V^T fragment loads and the prologue of a job serve twice the pairs, and the launch has no second, partly filled round.
// Here ONE workgroup of EIGHT waves owns 128 consecutive positions of a sequence: every wave runs flash_job<4> (128 queries of one
// head: the K / V^T fragment loads and the prologue serve twice the pairs) for head 8 pass + wave, two passes; the output rows go
// into a bf16 LDS panel [128][384] (96 KB: one workgroup per CU, which the 256-register budget of two waves per SIMD implies anyway);
// then the eight waves run the out-projection as a 2 x 4 grid of k_proj<0>'s wave tiles (wave (g, w): rows 64 g .. 64 g + 63,
// features 96 w .. 96 w + 95) and the staged residual epilogue over eight 12 KiB slabs.  cfg-2: 64 sequences x 8 chunks = 512
// workgroups = exactly two rounds of 256 (one round per sub-batch stream); ATLAS 512 / 500.
// Measured and dropped (profiles/r05_experiments.txt 17): the same kernel at THREE waves per SIMD -- six-wave workgroups (the
// dispatcher keeps only one per CU at 168 registers: scripts/micro/occ_probe.hip) and twelve-wave workgroups with 32 (head, half)
// flash_job<2> jobs (a tie: the loop is bound by the VALU issue port, 16 quarter-rate exponentials per pair, at any occupancy).
// =================================================================================================
__global__ __launch_bounds__(512, 1) void k_flash_proj8(const FlashProjParams p) {
    constexpr int NQ = 4, kRows = 32 * NQ;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * sizeof(PanelRows) + kRows * kC * 2];
    PanelRows* pr = reinterpret_cast<PanelRows*>(smem);   // [2]: one row table per 64-row half
    unsigned char* panel = smem + 2 * sizeof(PanelRows);
    const int w8 = __builtin_amdgcn_readfirstlane(wave_id()), g = w8 >> 2, w = w8 & 3;
    const int nqc = (p.f.ax.len + kRows - 1) / kRows;
    const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int qc = rest % nqc, seq = (rest / nqc) * 8 + xcd;
    if (seq >= p.f.ax.nseq) return;
    if (threadIdx.x < kRows) {   // waves 0 and 1 fill the two row tables (read after the barrier below)
        const int hf = threadIdx.x >> 6, i = threadIdx.x & 63;
        const int pos = qc * kRows + hf * kPanel + i;
        long tk = -1, mo = 0;
        if (pos < p.f.ax.len) {
            tk = p.f.ax.token(seq, pos);
            mo = p.mm.row_off(tk);
        }
        pr[hf].tok[i] = (int)tk;
        pr[hf].moff[i] = (int)mo;
        set_uniform(&pr[hf], i, tk, mo);
    }
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
#ifndef MDGEN_DEV_FLASH_NOPRIO
        // The two waves of a SIMD (w8, w8 + 4) take turns at static priority, the second-dispatched one first: the arbiter otherwise serves
        // the older wave on every conflict (MI355X_MICROARCH "two waves per SIMD"), both waves walk their jobs in step and each one's
        // prologue / anchor / store phases meet the other's.  One flip per job, none inside the loop.  k_flash_proj8 204.3 / 205.8 ->
        // 195.4 / 196.3 us per launch at cfg-2 (profiles/r06_experiments.txt #2); a fixed priority for waves 4..7 alone: no change.
        if ((w8 >= 4) == (pass == 0)) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
#endif
        const int head = 8 * pass + w8;
        FlashPre<NQ> pre;
        flash_prefetch<NQ>(p.f, seq, head, qc, pre);
        flash_job<NQ>(p.f, seq, head, qc, w8, pre, FlashStorePanel{(lds_byte*)panel, head});
    }
#ifndef MDGEN_DEV_FLASH_NOPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    __syncthreads();   // the attention output of all 16 heads is in the panel
    const int lane = lane_id();
    f32x16 acc[6];
#ifndef MDGEN_DEV_FLASH_NOEARLY
    // the residual rows of the epilogue's first batch are requested BEFORE the out-projection GEMM (they do not depend on it), and every
    // later batch before the stores of the one ahead of it: 204 -> 200-201 us per launch at cfg-2 (profiles/r06_experiments.txt #2)
    EpiPre<8> ep;
    epi_rmw_request<8>(0, 0, &pr[g], 96 * w, p.h, ep);
    __builtin_amdgcn_sched_barrier(0);
#endif
    zero_acc<6>(acc);
    wave_gemm<2, 3, 24, false>(panel, kC * 2, 2 * g, 0, p.wo + (size_t)(3 * w) * 24 * 64 + lane, 24 * 64, acc);
    __syncthreads();   // every wave is done reading the panel: reuse it as eight 12 KiB staging slabs
#ifndef MDGEN_DEV_FLASH_NOEARLY
    epilogue_gate_residual_lds_pre<3>(acc, &pr[g], reinterpret_cast<float*>(panel) + w8 * (32 * 96), 96 * w, p.bo, p.mm, p.gate_chunk, true,
                                      p.h, ep);
#else
    epilogue_gate_residual_lds<3>(acc, &pr[g], reinterpret_cast<float*>(panel) + w8 * (32 * 96), 96 * w, p.bo, p.mm, p.gate_chunk, true,
                                  p.h);
#endif
}

// Measured (cfg-2 / cfg-4, same box, back to back): NQ = 4 runs 160-162 / 76 us, NQ = 2 160-164 / 76 us -- a tie; 2 keeps
// three waves per SIMD and is faster on short sequences (IPA stack: 10 vs 14 us).
#ifndef MDGEN_FLASH_NQ
#define MDGEN_FLASH_NQ 2
#endif
void launch_flash(const FlashParams& p, hipStream_t s) {
    constexpr int NQ = MDGEN_FLASH_NQ;
    const int nqc = (p.ax.len + 32 * NQ - 1) / (32 * NQ);
    const int npair8 = (p.ax.nseq * 4 + 7) / 8;   // (sequence, head group) pairs, in groups of 8 (one per XCD)
    hipLaunchKernelGGL(k_flash<NQ>, dim3(npair8 * nqc * 8), dim3(256), 0, s, p);
}

// jobs of a fused launch: one workgroup per (sequence, 64-query chunk)
long flash_proj_jobs(const AxisMap& ax) { return (long)ax.nseq * ((ax.len + kPanel - 1) / kPanel); }

void launch_flash_proj(const FlashProjParams& p, int form, hipStream_t s) {
    const int nseq8 = (p.f.ax.nseq + 7) / 8;   // sequences, in groups of 8 (one per XCD)
    if (form == 8) {
        const int nqc = (p.f.ax.len + 2 * kPanel - 1) / (2 * kPanel);
        hipLaunchKernelGGL(k_flash_proj8, dim3(nseq8 * nqc * 8), dim3(512), 0, s, p);
        return;
    }
    const int nqc = (p.f.ax.len + kPanel - 1) / kPanel;
    hipLaunchKernelGGL(k_flash_proj, dim3(nseq8 * nqc * 8), dim3(256), 0, s, p);
}

}  // namespace mdg
