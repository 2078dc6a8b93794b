// k_flash.hip -- streaming-softmax multi-head attention with RoPE'd operands, the learned bias key
// and key-padding mask, for one attention axis (temporal: 1001 keys; residue: 257 keys).
//
// Replaces mha.py:359-396: bmm(q,k^T) -> masked_fill(-inf) -> fp32 softmax -> bmm(p,v), without
// ever materialising the (bsz*16, len, len+1) score tensor (4.1 GB per layer at cfg-2 in the
// reference).  The head-averaged attention weights the reference computes and discards
// (mha.py:399-405) are not reproduced.
//
// Operand fragments are produced by k_ln_qkv in exactly the register layout the MFMAs want
// (DESIGN.md "fragment layout"), so this kernel issues only fully-coalesced 16/8-byte loads:
//   S^T[key][q] = K-frag (A) x Q-frag (B)      2 x mfma_32x32x16 (d = 24 padded to 32)
//   O^T[d][q]  += V^T-frag (A) x P^T (B)       2 x mfma_32x32x16, P^T = exp2(S^T - m) packed in place
// A lane owns one query column (q = lane&31) and half of the tile's keys, so the row max is lane-local plus
// one exchange with lane^32.  One wave = one head x 64 queries (two 32-query tiles A, B); workgroup = 4 heads of
// the same queries; three workgroups per CU (<= 168 VGPRs): per-wave issue (one instruction per ~4-5 cycles,
// profiles/r02_issue_rate.txt) is what bounds a softmax loop, so waves per SIMD matter more than anything else.
// Per score: max3 + exp + cvt, nothing more:
//   * the running shift -m rides in a spare K-dim slot of Q (bf16-exact), so the score MFMA returns s - m;
//   * an all-ones V^T row (d = 24) makes the PV MFMA accumulate the softmax denominator;
//   * the shift is re-anchored only when a tile's max exceeds it by 2^kDefer (one scalar test per pair);
//   * software pipeline over (key tile, query tile) pairs A0 B0 A1 B1 ...: the block of a pair issues the score
//     MFMAs of the NEXT pair and the PV MFMAs of the PREVIOUS pair beside its own 16 exps -- every MFMA of a block
//     has its operands ready when the block starts, and the two neighbours belong to the other query tile;
//   * K/V are prefetched one to two tiles ahead with unconditional loads, tiles in linear order (K: SGPR base);
//   * the learned bias key is an ordinary entry of the fragments (key slot len, written by k_ln_qkv).
#include "kernels.h"

namespace mdg {

__device__ __forceinline__ bf16x8 frag16(const unsigned char* p) { return *reinterpret_cast<const bf16x8*>(p); }

__device__ __forceinline__ bf16x8 frag8(const unsigned char* p) {  // 4 real bf16 + 4 zeros
    const u32x2 v = *reinterpret_cast<const u32x2*>(p);
    const u32x4 w = u32x4{v[0], v[1], 0u, 0u};
    return __builtin_bit_cast(bf16x8, w);
}

// K / V^T fragments of one 32-key tile.  They are fetched ahead with ORDINARY loads (so hipcc counts them and
// inserts exact vmcnt waits) and pinned in place by sched_barrier(0) fences: without the fence the scheduler
// sinks each load to its first use and every tile pays a full L2 round trip.  (Inline-asm loads are not an
// option here: across the loop back-edge hipcc copies the asm outputs before the data lands.)
struct KTile {
    u32x4 k0;   // K k-step 0: 8 bf16
    u32x2 k1;   // K k-step 1: 4 bf16 (+ the constant-one slot and 3 implicit zeros)
};
struct VTile {
    u32x4 v0;   // V^T k-step 0
    u32x4 v1;   // V^T k-step 1
};
struct QTile {
    bf16x8 q0, q1;   // k-steps 0 / 1 of one 32-query tile; k-step-1 slot 4 of the lanes hh == 0 carries -m
};
struct PTile {
    bf16x8 p0, p1;   // P^T = exp2(S^T - m) of one (32-key, 32-query) tile, k-steps 0 / 1 of the PV MFMAs
};

// Running softmax state of ONE 32-query tile.  The row sum is NOT kept here: V^T row 24 (a padding row, d >= 24)
// is all ones, so the PV MFMA accumulates l = sum_k P[k][q] into O^T[24][q] (register 12 of the lanes with
// hh == 0) and every rescale of O rescales l with it.
struct Half {
    f32x16 o;
    float m;                     // applied shift (exactly representable in bf16; rides in a spare K-dim slot of Q)
    unsigned long long unanch;   // wave-level lane mask (SGPR pair): lanes whose shift has NOT yet been anchored to a
                                 // finite score.  All zero after the first tile or two: then only "max moved up" matters
};

__device__ __forceinline__ float half_max(float x) {   // max over the two half-waves (lane, lane^32)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

__device__ __forceinline__ float round_bf16(float x) { return __uint_as_float(pack_bf16(x, 0.f) << 16); }

// How far the running shift may lag the true row max before O is rescaled (log2 units: P <= 2^kDefer).
constexpr float kDefer = 8.f;

// Softmax bookkeeping of one (key tile, query tile) pair, everything except exp: key-padding mask, row max, and
// -- rarely -- moving the shift (which rewrites Q's -m slot and rescales O).  `vm`: validity bits of the tile's 32
// keys (wave-uniform, in an SGPR).
__device__ __forceinline__ void softmax_stats(Half& st, f32x16& s, QTile& q, uint32_t vm, int hh) {
    if (vm != 0xffffffffu) {   // rare (the last tile of a sequence, padded residues): keep ALL of it inside the branch
        uint32_t vmh = vm >> (4 * hh);
        asm volatile("" : "+v"(vmh));   // opaque: else the 16 bit tests are hoisted above the branch and always run
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = (vmh >> ((r & 3) + 8 * (r >> 2))) & 1u;
            s[r] = ok ? s[r] : -1e30f;
        }
    }
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
        // new shift = bf16(m + tile max); e = what this tile's already-shifted scores still have to lose
        const float r = mv ? round_bf16(st.m + t) : st.m;
        const float e = r - st.m;
        st.m = r;
        // e < 0 only while nothing has been accumulated yet (first anchoring): O is still zero, keep it so
        const float a = __builtin_amdgcn_exp2f(fminf(-e, 0.f));
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            st.o[i] *= a;
            s[i] -= e;
        }
        if (hh == 0) q.q1[4] = (__bf16)(-r);   // -m into Q's spare slot (k-step 1, slot 4)
    }
}

// The MFMA-dense block of one (key tile, query tile) pair, one straight-line region in which every MFMA has its
// operands ready at entry:
//   od += V^T P^T of the PREVIOUS pair            2 MFMAs (that pair belongs to the OTHER query tile, so a shift
//                                                 moved by this pair's softmax_stats never concerns it)
//   n  = S^T - m of the FOLLOWING pair            K-frag (A) x Q-frag (B), 2 MFMAs
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
    const bf16x8 k0 = __builtin_bit_cast(bf16x8, kn.k0);
    const u32x4 k1w = u32x4{kn.k1[0], kn.k1[1], 0x00003f80u, 0u};   // slot 4 = 1.0: picks up -m from Q
    const bf16x8 k1 = __builtin_bit_cast(bf16x8, k1w);
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    od = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vp.v0), pp.p0, od, 0, 0, 0);
    n = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qn.q0, zero, 0, 0, 0);
    od = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vp.v1), pp.p1, od, 0, 0, 0);
    n = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qn.q1, n, 0, 0, 0);
    float e[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) e[j] = __builtin_amdgcn_exp2f(s[j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        pc.p0[j] = (__bf16)e[j];
        pc.p1[j] = (__bf16)e[8 + j];
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

__global__ __launch_bounds__(256, 3) void k_flash(const FlashParams p) {
    const int lane = lane_id(), hh = lane >> 5, ql = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(wave_id());   // wave-uniform, and the compiler knows it: the K stream
                                                               // is then addressed as SGPR base + lane offset
    const int len = p.ax.len, nt = p.ax.ntile();   // tiles per (seq, head): they cover the len keys + the bias key
    const int nqc = (len + 63) / 64;
    // (sequence, head group) pairs are dealt to the 8 XCDs (block b runs on XCD b % 8) so that ALL q-chunks of a
    // pair -- which stream the same K/V -- share one L2: K/V leave HBM once.
    const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int qc = rest % nqc, pair = (rest / nqc) * 8 + xcd;
    const int seq = pair >> 2, hg = pair & 3;
    if (seq >= p.ax.nseq) return;
    const int head = hg * 4 + w;
    const long fbase = (long)(seq * kH + head) * nt * kFragBytes;
    const unsigned char* qb = p.qf + fbase;
    const long seq_base = p.ax.token(seq, 0);
    const int pstride = p.ax.pos_stride;

    // ---- per-tile key validity bitmasks (key-padding mask; the bias key at position len is always valid;
    //      positions > len are invalid), shared by the 4 waves (same sequence): wave w fills tiles w, w+4, ...
    __shared__ uint32_t vmask[260];
    for (int kt = w; kt < nt + 1; kt += 4) {
        const int pos = kt * 32 + ql;
        const bool ok = kt < nt && (pos == len || (pos < len && p.mk.at(seq_base + (long)(pos < len ? pos : 0) * pstride) != 0.f));
        const uint32_t vm = (uint32_t)__ballot(ok && hh == 0);
        if (lane == 0) vmask[kt] = vm;
    }
    __syncthreads();

    // ---- Q fragments of q-tiles 2qc, 2qc+1 (the second may not exist: reuse the first, never stored)
    const int qt0 = 2 * qc;
    const bool has2 = (qt0 + 1) * 32 < len;
    const int qt1 = has2 ? qt0 + 1 : qt0;
    QTile qa, qb_;
    qa.q0 = frag16(qb + (long)qt0 * kFragBytes + lane * 16);
    qa.q1 = frag8(qb + (long)qt0 * kFragBytes + 1024 + lane * 8);
    qb_.q0 = frag16(qb + (long)qt1 * kFragBytes + lane * 16);
    qb_.q1 = frag8(qb + (long)qt1 * kFragBytes + 1024 + lane * 8);

    // per-lane fragment streams.  K: uniform base + lane offset.  V^T: rows d < 24 are lanes of the fragment; rows
    // d > 24 read a zero page, row 24 a ones page (stride 0): the all-ones row accumulates the softmax denominator.
    const bool vreal = ql < kDH;
    const unsigned char* vptr = vreal ? p.vf + fbase + hh * 384 + ql * 16 : p.zero_page + (ql == kDH ? 128 : 0);
    const unsigned vstep = vreal ? kFragBytes : 0;
    const unsigned char* kbase = p.kf + fbase;   // wave-uniform
    const unsigned ko0 = lane * 16, ko1 = 1024 + lane * 8;

    Half ha, hb;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        ha.o[r] = opaque_zero();
        hb.o[r] = opaque_zero();
    }
    ha.m = hb.m = 0.f;
    ha.unanch = hb.unanch = ~0ull;

    // Loads are UNCONDITIONAL: tiles past the end of this (sequence, head) read whatever follows in the fragment
    // buffer (finite bf16 of the next head, or the zeroed tail) and their results are dropped -- a load under `if`
    // makes hipcc wait for the data right behind it.
    auto issue_k = [&](KTile& t, int kt) {
        const unsigned char* b = kbase + (long)kt * kFragBytes;
        t.k0 = *reinterpret_cast<const u32x4*>(b + ko0);
        t.k1 = *reinterpret_cast<const u32x2*>(b + ko1);
    };
    auto issue_v = [&](VTile& t, int kt) {
        const unsigned char* b = vptr + (unsigned long)((unsigned)kt * vstep);
        t.v0 = *reinterpret_cast<const u32x4*>(b);
        t.v1 = *reinterpret_cast<const u32x4*>(b + 768);   // the constant page repeats itself at +768
    };

    // One key tile = two (key tile, query tile) pairs, A then B.  Pipeline: the block of pair i issues the score
    // MFMAs of pair i+1 and the PV MFMAs of pair i-1 beside its own exps.  Score tuples sa / sb and P tuples pa / pb
    // belong to query tile A / B for good, so nothing is ever copied; K / V tiles alternate between two slots by
    // tile parity (kx, vx: even tiles; ky, vy: odd tiles).
    f32x16 sa, sb;
    PTile pa, pb;
    KTile kx, ky;
    VTile vx, vy;
    pb.p0 = pb.p1 = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});   // "previous pair" of the very first block: P = 0
    uint32_t vm_next = __builtin_amdgcn_readfirstlane(vmask[0]);
    auto tile = [&](int t, KTile& kc, KTile& kn, VTile& vc, VTile& vprev) {
        const uint32_t vm = vm_next;
        const uint32_t vmv = vmask[t + 1];   // the next tile's mask: requested now, moved to an SGPR at the end
        // pair (t, A): scores in sa.  Block: sb <- scores (t, B);  hb.o += V(t-1) P_B(t-1);  pa <- exp(sa)
        softmax_stats(ha, sa, qa, vm, hh);
        __builtin_amdgcn_sched_barrier(0);
        block(kc, qb_, sb, vprev, pb, hb.o, sa, pa);
        __builtin_amdgcn_sched_barrier(0);
        issue_v(vprev, t + 1);   // V(t-1) and K(t) (for query tile B: just issued) are consumed: refill both slots
        issue_k(kc, t + 2);
        __builtin_amdgcn_sched_barrier(0);
        // pair (t, B): scores in sb.  Block: sa <- scores (t+1, A);  ha.o += V(t) P_A(t);  pb <- exp(sb)
        softmax_stats(hb, sb, qb_, vm, hh);
        __builtin_amdgcn_sched_barrier(0);
        block(kn, qa, sa, vc, pa, ha.o, sb, pb);
        __builtin_amdgcn_sched_barrier(0);
        vm_next = __builtin_amdgcn_readfirstlane(vmv);
    };

    issue_k(kx, 0);
    issue_v(vy, 0);    // stands in for "V(-1)": any finite values (its P is zero)
    issue_k(ky, 1);
    issue_v(vx, 0);
    __builtin_amdgcn_sched_barrier(0);
    {   // scores of pair (0, A)
        const bf16x8 k0 = __builtin_bit_cast(bf16x8, kx.k0);
        const u32x4 k1w = u32x4{kx.k1[0], kx.k1[1], 0x00003f80u, 0u};
        const bf16x8 k1 = __builtin_bit_cast(bf16x8, k1w);
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qa.q0, zero, 0, 0, 0);
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qa.q1, sa, 0, 0, 0);
        asm volatile("" :: "v"(k0), "v"(k1), "v"(qa.q0), "v"(qa.q1));
    }
    // The loop body is a PAIR of tiles with a single exit (an exit between the two makes hipcc keep O in
    // different registers in the two halves and copy it mid-chain); an odd tile count gets a tail tile.
    int t = 0;
    for (; t + 1 < nt; t += 2) {
        tile(t, kx, ky, vx, vy);
        tile(t + 1, ky, kx, vy, vx);
    }
    if (t < nt) {
        tile(t, kx, ky, vx, vy);
        // pending: P_B of the last tile (in pb) with V(last) = vx
        hb.o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vx.v0), pb.p0, hb.o, 0, 0, 0);
        hb.o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vx.v1), pb.p1, hb.o, 0, 0, 0);
    } else {
        hb.o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vy.v0), pb.p0, hb.o, 0, 0, 0);
        hb.o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vy.v1), pb.p1, hb.o, 0, 0, 0);
    }
    const float l0 = __shfl(ha.o[12], ql, 64);   // O^T row 24 = softmax denominator, held by lanes hh == 0
    const float l1 = __shfl(hb.o[12], ql, 64);
    const f32x16 o0 = ha.o, o1 = hb.o;
    // ---- epilogue: registers 0..11 of lane-half hh are features 12*hh .. 12*hh+11 of this head
    {
        const int pos = qt0 * 32 + ql;
        if (pos < len) {
            const float inv = 1.0f / l0;
            u32x2* d = reinterpret_cast<u32x2*>(p.obuf + (seq_base + (long)pos * pstride) * kC + head * kDH + hh * 12);
            d[0] = u32x2{pack_bf16(o0[0] * inv, o0[1] * inv), pack_bf16(o0[2] * inv, o0[3] * inv)};
            d[1] = u32x2{pack_bf16(o0[4] * inv, o0[5] * inv), pack_bf16(o0[6] * inv, o0[7] * inv)};
            d[2] = u32x2{pack_bf16(o0[8] * inv, o0[9] * inv), pack_bf16(o0[10] * inv, o0[11] * inv)};
        }
    }
    if (has2) {
        const int pos = qt1 * 32 + ql;
        if (pos < len) {
            const float inv = 1.0f / l1;
            u32x2* d = reinterpret_cast<u32x2*>(p.obuf + (seq_base + (long)pos * pstride) * kC + head * kDH + hh * 12);
            d[0] = u32x2{pack_bf16(o1[0] * inv, o1[1] * inv), pack_bf16(o1[2] * inv, o1[3] * inv)};
            d[1] = u32x2{pack_bf16(o1[4] * inv, o1[5] * inv), pack_bf16(o1[6] * inv, o1[7] * inv)};
            d[2] = u32x2{pack_bf16(o1[8] * inv, o1[9] * inv), pack_bf16(o1[10] * inv, o1[11] * inv)};
        }
    }
}

void launch_flash(const FlashParams& p, hipStream_t s) {
    const int nqc = (p.ax.len + 63) / 64;
    const int npair8 = (p.ax.nseq * 4 + 7) / 8;   // (sequence, head group) pairs, in groups of 8 (one per XCD)
    hipLaunchKernelGGL(k_flash, dim3(npair8 * nqc * 8), dim3(256), 0, s, p);
}

}  // namespace mdg
