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
// A lane owns one query column (q = lane&31) and half of the tile's keys, so the row max / sum are
// lane-local plus one exchange with lane^32.
// One wave = one head x 64 queries (2 q-tiles); workgroup = 4 heads of the same queries.
#include "kernels.h"

namespace mdg {

__device__ __forceinline__ bf16x8 frag16(const unsigned char* p) { return *reinterpret_cast<const bf16x8*>(p); }

__device__ __forceinline__ bf16x8 frag8(const unsigned char* p) {  // 4 real bf16 + 4 zeros
    const u32x2 v = *reinterpret_cast<const u32x2*>(p);
    const u32x4 w = u32x4{v[0], v[1], 0u, 0u};
    return __builtin_bit_cast(bf16x8, w);
}

// K / V^T fragments of one 32-key tile.  They are fetched one tile ahead with ORDINARY loads (so hipcc counts
// them and inserts exact vmcnt waits) and pinned in place by sched_barrier(0) fences: without the fence the
// scheduler sinks each load to its first use and every tile pays a full L2 round trip.  (Inline-asm loads are
// not an option here: across the loop back-edge hipcc copies the asm outputs before the data lands.)
struct KVTile {
    u32x4 k0;   // K k-step 0: 8 bf16
    u32x2 k1;   // K k-step 1: 4 bf16 (+4 implicit zeros)
    u32x4 v0;   // V^T k-step 0
    u32x4 v1;   // V^T k-step 1
};

struct FlashState {
    f32x16 o0, o1;
    float m0, m1, l0, l1;
};

// one 32-key tile against the wave's two 32-query tiles
__device__ __forceinline__ void flash_tile(FlashState& st, const KVTile& t, const bf16x8 q00, const bf16x8 q01,
                                           const bf16x8 q10, const bf16x8 q11, const f32x16& zc, uint32_t vm, bool last,
                                           int kl_last, const bf16x8 kb0, const bf16x8 kb1, __bf16 bvd, int hh, int ql) {
    bf16x8 k0 = __builtin_bit_cast(bf16x8, t.k0);
    const u32x4 k1w = u32x4{t.k1[0], t.k1[1], 0u, 0u};
    bf16x8 k1 = __builtin_bit_cast(bf16x8, k1w);
    bf16x8 v0 = __builtin_bit_cast(bf16x8, t.v0);
    bf16x8 v1 = __builtin_bit_cast(bf16x8, t.v1);
    if (last) {  // wave-uniform: splice in the learned bias key/value, zero anything beyond it
        if (ql == kl_last) {
            k0 = kb0;
            k1 = kb1;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int key0 = (j & 3) + 8 * (j >> 2) + 4 * hh;        // k-step 0 slot j
            const int key1 = (j & 3) + 8 * (2 + (j >> 2)) + 4 * hh;  // k-step 1 slot j
            if (key0 == kl_last) v0[j] = bvd; else if (key0 > kl_last) v0[j] = (__bf16)0.f;
            if (key1 == kl_last) v1[j] = bvd; else if (key1 > kl_last) v1[j] = (__bf16)0.f;
        }
    }
    // C is a LIVE all-zero register tuple (never the inline constant, common.h): dst != C, and the build's
    // ISA check guarantees dst does not overlap A/B.
    f32x16 s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q00, zc, 0, 0, 0);
    f32x16 s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q10, zc, 0, 0, 0);
    s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q01, s0, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q11, s1, 0, 0, 0);
    if (vm != 0xffffffffu) {
        const uint32_t vmh = vm >> (4 * hh);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = (vmh >> ((r & 3) + 8 * (r >> 2))) & 1u;
            s0[r] = ok ? s0[r] : -1e30f;
            s1[r] = ok ? s1[r] : -1e30f;
        }
    }
    // ---- online softmax (log2 domain: q carries dh^-1/2 * log2(e))
    float t0 = s0[0], t1 = s1[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) {
        t0 = fmaxf(t0, s0[r]);
        t1 = fmaxf(t1, s1[r]);
    }
    t0 = fmaxf(t0, __shfl_xor(t0, 32, 64));
    t1 = fmaxf(t1, __shfl_xor(t1, 32, 64));
    const bool grow = (t0 > st.m0) | (t1 > st.m1);
    if (__builtin_amdgcn_ballot_w64(grow) != 0) {   // wave-uniform: some query's running max moved -> rescale
        const float n0 = fmaxf(st.m0, t0), n1 = fmaxf(st.m1, t1);
        const float a0 = __builtin_amdgcn_exp2f(st.m0 - n0), a1 = __builtin_amdgcn_exp2f(st.m1 - n1);
        st.m0 = n0;
        st.m1 = n1;
        st.l0 *= a0;
        st.l1 *= a1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st.o0[r] *= a0;
            st.o1[r] *= a1;
        }
    }
    float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        s0[r] = __builtin_amdgcn_exp2f(s0[r] - st.m0);
        s1[r] = __builtin_amdgcn_exp2f(s1[r] - st.m1);
        ps0 += s0[r];
        ps1 += s1[r];
    }
    st.l0 += ps0;
    st.l1 += ps1;
    bf16x8 p00, p01, p10, p11;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        p00[j] = (__bf16)s0[j];
        p01[j] = (__bf16)s0[8 + j];
        p10[j] = (__bf16)s1[j];
        p11[j] = (__bf16)s1[8 + j];
    }
    st.o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, p00, st.o0, 0, 0, 0);
    st.o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, p10, st.o1, 0, 0, 0);
    st.o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, p01, st.o0, 0, 0, 0);
    st.o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, p11, st.o1, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void k_flash(const FlashParams p) {
    const int lane = lane_id(), w = wave_id(), hh = lane >> 5, ql = lane & 31;
    const int len = p.ax.len, ntile = p.ax.ntile();
    const int nqc = (len + 63) / 64;
    const int hg = blockIdx.x & 3;
    const int rest = blockIdx.x >> 2;
    const int qc = rest % nqc, seq = rest / nqc;
    const int head = hg * 4 + w;
    const long fbase = (long)(seq * kH + head) * ntile * kFragBytes;
    const unsigned char* qb = p.qf + fbase;
    const long seq_base = p.ax.token(seq, 0);
    const int pstride = p.ax.pos_stride;

    // ---- per-tile key validity bitmasks (key-padding mask; the bias key is always valid), shared
    //      by the 4 waves (same sequence): wave w fills tiles w, w+4, ...
    __shared__ uint32_t vmask[256];
    for (int kt = w; kt < ntile; kt += 4) {
        const int pos = kt * 32 + ql;
        bool ok = false;
        if (pos < len) ok = p.mk.at(seq_base + (long)pos * pstride) != 0.f;
        else if (pos == len) ok = true;
        const uint32_t vm = (uint32_t)__ballot(ok && hh == 0);
        if (lane == 0) vmask[kt] = vm;
    }
    __syncthreads();

    // ---- Q fragments of q-tiles 2qc, 2qc+1 (the second may not exist: reuse the first, never stored)
    const int qt0 = 2 * qc;
    const bool has2 = (qt0 + 1) * 32 < len;
    const int qt1 = has2 ? qt0 + 1 : qt0;
    const bf16x8 q00 = frag16(qb + (long)qt0 * kFragBytes + lane * 16);
    const bf16x8 q01 = frag8(qb + (long)qt0 * kFragBytes + 1024 + lane * 8);
    const bf16x8 q10 = frag16(qb + (long)qt1 * kFragBytes + lane * 16);
    const bf16x8 q11 = frag8(qb + (long)qt1 * kFragBytes + 1024 + lane * 8);

    // ---- learned bias key/value (mha.py:265-268): key index `len`, rotated at position `len`
    const int kt_last = len >> 5, kl_last = len & 31;
    bf16x8 kb0, kb1;
    {
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
        const u32x4 a = u32x4{pack_bf16(e[0], e[1]), pack_bf16(e[2], e[3]), pack_bf16(e[4], e[5]), pack_bf16(e[6], e[7])};
        const u32x4 b = u32x4{pack_bf16(e[8], e[9]), pack_bf16(e[10], e[11]), 0u, 0u};
        kb0 = __builtin_bit_cast(bf16x8, a);
        kb1 = __builtin_bit_cast(bf16x8, b);
    }
    // V^T fragment row d = lane&31 (rows >= 24 are zero padding); feature of row d = psi(d)
    const int dpsi = 12 * ((ql >> 2) & 1) + 4 * (ql >> 3) + (ql & 3);
    const __bf16 bvd = (ql < kDH) ? (__bf16)p.bias_v[head * kDH + dpsi] : (__bf16)0.f;

    // per-lane fragment streams: K rows are lanes; V^T rows d >= 24 read a zero page with stride 0
    const bool vreal = ql < kDH;
    const unsigned char* vp = vreal ? p.vf + fbase + hh * 384 + ql * 16 : p.zero_page;
    const long vstep = vreal ? kFragBytes : 0;

    FlashState st;
    f32x16 zc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        st.o0[r] = opaque_zero();
        st.o1[r] = opaque_zero();
        zc[r] = opaque_zero();
    }
    st.m0 = st.m1 = -1e30f;
    st.l0 = st.l1 = 0.f;

    // K fragment: k-step 0 at +lane*16, k-step 1 at +1024+lane*8 -> two base pointers
    const unsigned char* k0p = p.kf + fbase + lane * 16;
    KVTile ta, tb;
    const unsigned char* k1p = p.kf + fbase + 1024 + lane * 8;
    const long v1off = vreal ? 768 : 0;
    auto issue = [&](KVTile& t, int kt) {
        t.k0 = *reinterpret_cast<const u32x4*>(k0p + (long)kt * kFragBytes);
        t.k1 = *reinterpret_cast<const u32x2*>(k1p + (long)kt * kFragBytes);
        t.v0 = *reinterpret_cast<const u32x4*>(vp + (long)kt * vstep);
        t.v1 = *reinterpret_cast<const u32x4*>(vp + (long)kt * vstep + v1off);
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch AHEAD of the tile that is about to be computed
    };
    issue(ta, 0);
    int kt = 0;
    for (; kt + 1 < ntile; kt += 2) {
        issue(tb, kt + 1);
        flash_tile(st, ta, q00, q01, q10, q11, zc, vmask[kt], kt == kt_last, kl_last, kb0, kb1, bvd, hh, ql);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < ntile) issue(ta, kt + 2);
        flash_tile(st, tb, q00, q01, q10, q11, zc, vmask[kt + 1], kt + 1 == kt_last, kl_last, kb0, kb1, bvd, hh, ql);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (kt < ntile)   // odd tile count: the last tile was requested in the final loop iteration (or before the loop)
        flash_tile(st, ta, q00, q01, q10, q11, zc, vmask[kt], kt == kt_last, kl_last, kb0, kb1, bvd, hh, ql);
    float l0 = st.l0 + __shfl_xor(st.l0, 32, 64);
    float l1 = st.l1 + __shfl_xor(st.l1, 32, 64);
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
