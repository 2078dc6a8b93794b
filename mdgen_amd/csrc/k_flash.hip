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
    const unsigned char* kb = p.kf + fbase;
    const unsigned char* vb = p.vf + fbase;
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

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        o0[r] = opaque_zero();
        o1[r] = opaque_zero();
    }
    float m0 = -1e30f, m1 = -1e30f, l0 = 0.f, l1 = 0.f;

    for (int kt = 0; kt < ntile; ++kt) {
        const unsigned char* kp = kb + (long)kt * kFragBytes;
        const unsigned char* vp = vb + (long)kt * kFragBytes;
        bf16x8 k0 = frag16(kp + lane * 16);
        bf16x8 k1 = frag8(kp + 1024 + lane * 8);
        bf16x8 v0, v1;
        if (ql < kDH) {
            v0 = frag16(vp + hh * 384 + ql * 16);
            v1 = frag16(vp + 768 + hh * 384 + ql * 16);
        } else {
            const u32x4 z = u32x4{0u, 0u, 0u, 0u};
            v0 = __builtin_bit_cast(bf16x8, z);
            v1 = v0;
        }
        const uint32_t vm = vmask[kt];
        if (kt == kt_last) {  // wave-uniform: splice in the bias key/value, zero anything beyond it
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
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = opaque_zero();   // see common.h: never let C fold to the inline constant
            s1[r] = opaque_zero();
        }
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q00, s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q10, s1, 0, 0, 0);
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
        const float n0 = fmaxf(m0, t0), n1 = fmaxf(m1, t1);
        const float a0 = __builtin_amdgcn_exp2f(m0 - n0), a1 = __builtin_amdgcn_exp2f(m1 - n1);
        m0 = n0;
        m1 = n1;
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(s0[r] - n0);
            s1[r] = __builtin_amdgcn_exp2f(s1[r] - n1);
            ps0 += s0[r];
            ps1 += s1[r];
            o0[r] *= a0;
            o1[r] *= a1;
        }
        l0 = l0 * a0 + ps0;
        l1 = l1 * a1 + ps1;
        bf16x8 p00, p01, p10, p11;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            p00[j] = (__bf16)s0[j];
            p01[j] = (__bf16)s0[8 + j];
            p10[j] = (__bf16)s1[j];
            p11[j] = (__bf16)s1[8 + j];
        }
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, p00, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, p10, o1, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, p01, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, p11, o1, 0, 0, 0);
    }
    l0 += __shfl_xor(l0, 32, 64);
    l1 += __shfl_xor(l1, 32, 64);
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
