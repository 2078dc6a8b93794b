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
// one exchange with lane^32.  One wave = one head x 64 queries (2 q-tiles); workgroup = 4 heads of the same
// queries.  The kernel is VALU-bound at dh = 24 (one v_exp per score), so the softmax is reduced to
// max3 + exp + cvt per score:
//   * the running shift -m rides in a spare K-dim slot of Q (bf16-exact), so the score MFMA returns s - m;
//   * an all-ones V^T row (d = 24) makes the PV MFMA accumulate the softmax denominator;
//   * the shift is re-anchored only when a tile's max exceeds it by 2^kDefer (wave-uniform ballot);
//   * the NEXT tile's score MFMAs are issued inside the current tile's exp block;
//   * K/V are prefetched with unconditional, wrapped tile indices (a load under `if` is a serialised load);
//   * the learned bias key is a register-built virtual tile after the real ones.
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

// Running softmax state of the wave's two 32-query tiles.  The row sum is NOT kept here: V^T row 24 (a padding
// row, d >= 24) is all ones, so the PV MFMA accumulates l = sum_k P[k][q] into O^T[24][q] (register 12 of the
// lanes with hh == 0) and every rescale of O rescales l with it.
struct FlashState {
    f32x16 o0, o1;
    float m0, m1;        // applied shift (exactly representable in bf16; rides in a spare K-dim slot of Q)
    bool anch0, anch1;   // the shift has been anchored to a finite score at least once
    bool settled;        // wave-uniform: every lane is anchored -> only "max moved up" needs checking
};

__device__ __forceinline__ float half_max(float x) {   // max over the two half-waves (lane, lane^32)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

__device__ __forceinline__ float round_bf16(float x) { return __uint_as_float(pack_bf16(x, 0.f) << 16); }

// How far the running shift may lag the true row max before O is rescaled (log2 units: P <= 2^kDefer).
constexpr float kDefer = 8.f;

struct QFrags {
    bf16x8 q00, q01, q10, q11;   // [q-tile][k-step]; k-step-1 slot 4 of the lanes hh == 0 carries -m
    f32x16 zc;                   // all-zero accumulator input of the score MFMAs (opaque to the compiler)
};

// S^T - m for one 32-key tile against both q-tiles.  C is a LIVE all-zero register tuple (QFrags::zc), never the
// inline constant 0: with the constant hipcc lets the destination overlap A (caught by build.py check_isa).
__device__ __forceinline__ void scores(const KTile& t, const QFrags& q, f32x16& s0, f32x16& s1) {
    const f32x16& zc = q.zc;
    const bf16x8 k0 = __builtin_bit_cast(bf16x8, t.k0);
    const u32x4 k1w = u32x4{t.k1[0], t.k1[1], 0x00003f80u, 0u};   // slot 4 = 1.0: picks up -m from Q
    const bf16x8 k1 = __builtin_bit_cast(bf16x8, k1w);
    s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q.q00, zc, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q.q10, zc, 0, 0, 0);
    s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q.q01, s0, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q.q11, s1, 0, 0, 0);
}

// Softmax bookkeeping of the current tile (everything except exp): key-padding mask, row max, and -- rarely --
// moving the shift.  Must run BEFORE the next tile's scores are issued, because it may rewrite Q's -m slot.
__device__ __forceinline__ void softmax_stats(FlashState& st, f32x16& s0, f32x16& s1, QFrags& q, uint32_t vm, int hh) {
    if (vm != 0xffffffffu) {
        const uint32_t vmh = vm >> (4 * hh);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = (vmh >> ((r & 3) + 8 * (r >> 2))) & 1u;
            s0[r] = ok ? s0[r] : -1e30f;
            s1[r] = ok ? s1[r] : -1e30f;
        }
    }
    float t0 = fmaxf(s0[0], s0[1]), t1 = fmaxf(s1[0], s1[1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) {
        t0 = fmaxf(fmaxf(t0, s0[r]), s0[r + 1]);   // v_max3_f32
        t1 = fmaxf(fmaxf(t1, s1[r]), s1[r + 1]);
    }
    t0 = half_max(t0);
    t1 = half_max(t1);
    const bool up = fmaxf(t0, t1) > kDefer;
    if (!st.settled || __builtin_amdgcn_ballot_w64(up) != 0) {   // wave-uniform and rare after the first tiles
        // Re-anchor when the shift lags the tile max by more than kDefer, or (first finite tile of a query
        // only) leads it by more than kDefer.  Both half-waves of a query see the same t -> same decision.
        const bool fin0 = t0 > -1e29f, fin1 = t1 > -1e29f;
        const bool mv0 = (t0 > kDefer) | (!st.anch0 & fin0 & (t0 < -kDefer));
        const bool mv1 = (t1 > kDefer) | (!st.anch1 & fin1 & (t1 < -kDefer));
        st.anch0 |= fin0;
        st.anch1 |= fin1;
        st.settled = __builtin_amdgcn_ballot_w64(st.anch0 & st.anch1) == ~0ull;
        // new shift = bf16(m + tile max); e = what this tile's already-shifted scores still have to lose
        const float r0 = mv0 ? round_bf16(st.m0 + t0) : st.m0, r1 = mv1 ? round_bf16(st.m1 + t1) : st.m1;
        const float e0 = r0 - st.m0, e1 = r1 - st.m1;
        st.m0 = r0;
        st.m1 = r1;
        // e < 0 only while nothing has been accumulated yet (first anchoring): O is still zero, keep it so
        const float a0 = __builtin_amdgcn_exp2f(fminf(-e0, 0.f)), a1 = __builtin_amdgcn_exp2f(fminf(-e1, 0.f));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st.o0[r] *= a0;
            st.o1[r] *= a1;
            s0[r] -= e0;
            s1[r] -= e1;
        }
        if (hh == 0) {   // -m into Q's spare slot (k-step 1, slot 4)
            q.q01[4] = (__bf16)(-r0);
            q.q11[4] = (__bf16)(-r1);
        }
    }
}

// The MFMA-dense block of one tile: P^T = exp2(S^T - m) packed in place, O^T += V^T P^T, and the scores of the
// FOLLOWING tile issued into the same straight-line block so its MFMAs run under this tile's exps.  It is one
// unconditional block on purpose (after the last tile the "next" scores are computed from stale K registers
// and dropped): with a branch hipcc hoists the exps above it and the overlap is lost.
__device__ __forceinline__ void exp_pv(FlashState& st, const f32x16& s0, const f32x16& s1, const bf16x8 v0, const bf16x8 v1,
                                       const KTile& kn, const QFrags& q, f32x16& n0, f32x16& n1) {
    scores(kn, q, n0, n1);
    bf16x8 p00, p01, p10, p11;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        p00[j] = (__bf16)__builtin_amdgcn_exp2f(s0[j]);
        p10[j] = (__bf16)__builtin_amdgcn_exp2f(s1[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        p01[j] = (__bf16)__builtin_amdgcn_exp2f(s0[8 + j]);
        p11[j] = (__bf16)__builtin_amdgcn_exp2f(s1[8 + j]);
    }
    st.o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, p00, st.o0, 0, 0, 0);
    st.o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, p10, st.o1, 0, 0, 0);
    st.o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, p01, st.o0, 0, 0, 0);
    st.o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, p11, st.o1, 0, 0, 0);
    {   // interleave: 4 score MFMAs under the first 32 exp/cvt, then the PV MFMAs as their P arrives
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x402, 8, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x402, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x402, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x402, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
}

__global__ __launch_bounds__(256, 2) void k_flash(const FlashParams p) {
    const int lane = lane_id(), w = wave_id(), hh = lane >> 5, ql = lane & 31;
    const int len = p.ax.len, ntile = p.ax.ntile();   // ntile: tiles per (seq, head) in the fragment LAYOUT
    const int nreal = (len + 31) >> 5;                // tiles that hold real keys
    const int nqc = (len + 63) / 64;
    const int hg = blockIdx.x & 3;
    const int rest = blockIdx.x >> 2;
    const int qc = rest % nqc, seq = rest / nqc;
    const int head = hg * 4 + w;
    const long fbase = (long)(seq * kH + head) * ntile * kFragBytes;
    const unsigned char* qb = p.qf + fbase;
    const long seq_base = p.ax.token(seq, 0);
    const int pstride = p.ax.pos_stride;

    // ---- per-tile key validity bitmasks (key-padding mask; positions >= len are invalid), shared by the
    //      4 waves (same sequence): wave w fills tiles w, w+4, ...
    __shared__ uint32_t vmask[256];
    for (int kt = w; kt < nreal; kt += 4) {
        const int pos = kt * 32 + ql;
        const bool ok = pos < len && p.mk.at(seq_base + (long)pos * pstride) != 0.f;
        const uint32_t vm = (uint32_t)__ballot(ok && hh == 0);
        if (lane == 0) vmask[kt] = vm;
    }
    __syncthreads();

    // ---- Q fragments of q-tiles 2qc, 2qc+1 (the second may not exist: reuse the first, never stored)
    const int qt0 = 2 * qc;
    const bool has2 = (qt0 + 1) * 32 < len;
    const int qt1 = has2 ? qt0 + 1 : qt0;
    QFrags q;
    q.q00 = frag16(qb + (long)qt0 * kFragBytes + lane * 16);
    q.q01 = frag8(qb + (long)qt0 * kFragBytes + 1024 + lane * 8);
    q.q10 = frag16(qb + (long)qt1 * kFragBytes + lane * 16);
    q.q11 = frag8(qb + (long)qt1 * kFragBytes + 1024 + lane * 8);

    // per-lane fragment streams: K rows are lanes; V^T rows d > 24 read a zero page, row 24 a ones page (stride
    // 0): row 24 is the all-ones row that accumulates the softmax denominator
    const bool vreal = ql < kDH;
    const unsigned char* vp = vreal ? p.vf + fbase + hh * 384 + ql * 16 : p.zero_page + (ql == kDH ? 128 : 0);
    const long vstep = vreal ? kFragBytes : 0;
    const long v1off = vreal ? 768 : 0;
    // K fragment: k-step 0 at +lane*16, k-step 1 at +1024+lane*8 -> two base pointers
    const unsigned char* k0p = p.kf + fbase + lane * 16;
    const unsigned char* k1p = p.kf + fbase + 1024 + lane * 8;

    FlashState st;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        st.o0[r] = opaque_zero();
        st.o1[r] = opaque_zero();
        q.zc[r] = opaque_zero();
    }
    st.m0 = st.m1 = 0.f;
    st.anch0 = st.anch1 = false;
    st.settled = false;

    // Loads are UNCONDITIONAL (the tile index wraps instead): a load under `if` makes hipcc merge old and new
    // register values right behind it, i.e. wait for the data it has just requested.
    auto issue_k = [&](KTile& t, int kt) {
        t.k0 = *reinterpret_cast<const u32x4*>(k0p + (long)kt * kFragBytes);
        t.k1 = *reinterpret_cast<const u32x2*>(k1p + (long)kt * kFragBytes);
    };
    auto issue_v = [&](VTile& t, int kt) {
        t.v0 = *reinterpret_cast<const u32x4*>(vp + (long)kt * vstep);
        t.v1 = *reinterpret_cast<const u32x4*>(vp + (long)kt * vstep + v1off);
    };
    // One tile: `s` holds its shifted scores (issued during the previous tile), `vt` its V^T, `kn` the NEXT
    // tile's K.  Order: stats (may move the shift in Q) -> [next scores || exp || PV].
    // The two score tuples ping-pong between "current" and "next" so nothing is copied.
    f32x16 sa0, sa1, sb0, sb1;
    auto step = [&](const KTile& kn, const VTile& vt, uint32_t vm, f32x16& s0, f32x16& s1, f32x16& n0, f32x16& n1) {
        softmax_stats(st, s0, s1, q, vm, hh);
        exp_pv(st, s0, s1, __builtin_bit_cast(bf16x8, vt.v0), __builtin_bit_cast(bf16x8, vt.v1), kn, q, n0, n1);
    };

    // Key tiles are visited in ROTATED order, starting at a q-chunk dependent tile (softmax does not care), so
    // that the workgroups sharing one sequence's K/V stream are not all first-touching the same lines.
    const int start = (qc * nreal) / nqc;
    auto tile = [&](int i) {   // i <= nreal + 2
        int t = start + i;
        t = t >= nreal ? t - nreal : t;
        t = t >= nreal ? t - nreal : t;
        return t >= nreal ? 0 : t;
    };

    KTile ka, kb;
    VTile va, vb;
    issue_k(ka, tile(0));
    issue_v(va, tile(0));
    issue_k(kb, tile(1));
    __builtin_amdgcn_sched_barrier(0);
    scores(ka, q, sa0, sa1);
    // Positions >= len of the last real tile hold finite leftovers (the K/V fragment regions are zeroed once per
    // call and only ever receive finite bf16), masked to P = 0.  After the last real tile the "next" scores are
    // computed from wrapped-around K and dropped.
    // The loop body is a PAIR of steps with a single exit (an exit between the two steps makes hipcc keep O in
    // different registers in the two halves and copy it mid-chain); an odd tile count gets a tail step.
    int i = 0;
    for (; i + 1 < nreal; i += 2) {
        // even step: V in va, next K in kb; refill vb <- V(i+1), ka <- K(i+2)
        issue_v(vb, tile(i + 1));
        issue_k(ka, tile(i + 2));
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch AHEAD of the tile that is about to be computed
        step(kb, va, vmask[tile(i)], sa0, sa1, sb0, sb1);
        __builtin_amdgcn_sched_barrier(0);
        // odd step: V in vb, next K in ka; refill va <- V(i+2), kb <- K(i+3)
        issue_v(va, tile(i + 2));
        issue_k(kb, tile(i + 3));
        __builtin_amdgcn_sched_barrier(0);
        step(ka, vb, vmask[tile(i + 1)], sb0, sb1, sa0, sa1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (i < nreal) step(kb, va, vmask[tile(i)], sa0, sa1, sb0, sb1);

    // ---- the learned bias key/value (mha.py:265-268): one extra key at index `len`, rotated at position `len`,
    //      always valid.  It is processed as a virtual tile built in registers: every K row is the bias key, only
    //      key slot 0 is unmasked, V^T column 0 is the bias value.
    {
        KTile kbias;
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
        kbias.k0 = u32x4{pack_bf16(e[0], e[1]), pack_bf16(e[2], e[3]), pack_bf16(e[4], e[5]), pack_bf16(e[6], e[7])};
        kbias.k1 = u32x2{pack_bf16(e[8], e[9]), pack_bf16(e[10], e[11])};
        // V^T fragment row d = lane&31 (24 = ones row, 25.. = zero); feature of row d = psi(d)
        const int dpsi = 12 * ((ql >> 2) & 1) + 4 * (ql >> 3) + (ql & 3);
        const float bvf = (ql < kDH) ? p.bias_v[head * kDH + dpsi] : (ql == kDH ? 1.f : 0.f);
        VTile vbias;
        vbias.v0 = u32x4{hh == 0 ? pack_bf16(bvf, 0.f) : 0u, 0u, 0u, 0u};   // key slot 0 = k-step 0, hh 0, j 0
        vbias.v1 = u32x4{0u, 0u, 0u, 0u};
        f32x16 c0, c1, d0, d1;
        scores(kbias, q, c0, c1);
        step(kbias, vbias, 0x1u, c0, c1, d0, d1);
    }
    const float l0 = __shfl(st.o0[12], ql, 64);   // O^T row 24 = softmax denominator, held by lanes hh == 0
    const float l1 = __shfl(st.o1[12], ql, 64);
    const f32x16 o0 = st.o0, o1 = st.o1;
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
    hipLaunchKernelGGL(k_flash, dim3(p.ax.nseq * nqc * 4), dim3(256), 0, s, p);
}

}  // namespace mdg
