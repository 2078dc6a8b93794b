// Attention of the training step with bf16 matrix operands (option train_precision = 16): forward with the
// log-sum-exp tape, and the two backward passes that start from that tape.  Same contract, same buffers and the same
// fp32 inputs / outputs as k32_attn / k32_attn_bwd_q / k32_attn_bwd_kv (k_fp32.hip, k_fp32_bwd.hip), which remain the
// exact-fp32 path; these kernels put the four products of the attention and the five of its backward on the matrix
// cores (reference: mha.py:265-268, 359-396 forward; the backward is what autograd derives from it).
//
// Shape of all three: a workgroup of four waves OWNS 128 rows of one (sequence, head) -- queries in the forward and in
// the query pass, keys in the key pass -- one 32-row tile per wave, held in registers as MFMA B operands (lane =
// row).  The OTHER side streams through LDS in chunks of 64 rows, converted to bf16 on the way in, the next chunk's
// global loads in flight while the current one is multiplied.  Every product is computed transposed, D[other][own],
// so that everything that is per own row (running max, log-sum-exp, delta = dO . O, key validity) is per LANE, and so
// that the accumulator registers of one product, packed to bf16 pairs, ARE the B operand of the next one (its
// contraction runs over the `other` rows in accumulator order; the LDS copy that supplies the matching A operand is
// written transposed in that order, perm_pos()).
//
//   forward      S^T = K Q^T            P^T = exp(S^T - m)        O^T += V^T P^T                 (online softmax over chunks)
//   query pass   S^T = K Q^T            dP^T = V dO^T - delta     dS^T = P^T o dP^T              dQ^T += K^T dS^T
//   key pass     S = Q K^T - lse        dP = dO V^T - delta       dV^T += dO^T P    dK^T += Q^T dS
//
// The learned bias key / value (rotated at position len, mha.py:359-366) is row `len` of the key side; key padding is
// an additive -3e38 that the score accumulators start from (forward, query pass) or a per-lane flag (key pass).
//
// Round 6: axes of 129 .. 256 positions (the ATLAS training shapes) take the SEQUENCE-RESIDENT forms further down --
// k16_attn_seq / k16_attn_bwd_seq: eight waves own all rows of one (sequence, head), the other side is converted into LDS once,
// scores in log2 units, per-row addends in the padding features of the operands, RoPE applied while converting (option
// train_attn_form; launch16_attn / launch16_attn_bwd choose).  The kernels right below remain for every other length.
#include "common.h"
#include "kernels.h"

namespace mdg {
namespace {

constexpr int kRowB = 80;      // bytes of one row of a row-major tile: 32 bf16 features (24 + 8 zeros) + 16 of padding
constexpr int kTrB = 144;      // bytes of one feature row of a transposed tile: 64 bf16 rows + 16 of padding
constexpr int kChunk = 64;     // streamed rows per LDS fill
constexpr float kMasked = -3.0e38f;

// experiment switches (csrc/dev.h; product builds define none): what one ingredient of the chunk loop costs
#ifdef MDGEN_DEV_ATTN16_NOEXP
#define A16_EXP2(x) (x)
#else
#define A16_EXP2(x) __builtin_amdgcn_exp2f(x)
#endif
#ifdef MDGEN_DEV_ATTN16_NOMMA
#define A16_MMA(a, b, c) (c)
#else
#define A16_MMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif
#ifdef MDGEN_DEV_ATTN16_NOBAR
#define A16_BARRIER() asm volatile("" ::: "memory")
#else
#define A16_BARRIER() __syncthreads()
#endif
#ifdef MDGEN_DEV_ATTN16_SEQNOLOAD   // k16_attn_bwd_seq: the fill's global loads / all but one tile of each pass / the result stores left out
#define A16_SEQ_NOLOAD true
#else
#define A16_SEQ_NOLOAD false
#endif
#ifdef MDGEN_DEV_ATTN16_SEQNOLOOP
#define A16_SEQ_NOLOOP true
#else
#define A16_SEQ_NOLOOP false
#endif
#ifdef MDGEN_DEV_ATTN16_SEQNOSTORE
#define A16_SEQ_NOSTORE true
#else
#define A16_SEQ_NOSTORE false
#endif
#ifdef MDGEN_DEV_ATTN16_NOSTAGE
#define A16_STAGE(c0) ((c0) == 0)
#else
#define A16_STAGE(c0) true
#endif

// Position, in the contraction order of a chained B operand, of row rho (0..31) of a tile: accumulator register
// 8 s + j of lane-half hh holds row mfma_row(8 s + j, hh) and becomes element (k-step s, 8 hh + j).
__device__ __forceinline__ int perm_pos(int rho) {
    return 16 * (rho >> 4) + 8 * ((rho >> 2) & 1) + 4 * ((rho >> 3) & 1) + (rho & 3);
}

__device__ __forceinline__ float half_sum(float x) {   // x(lane) + x(lane ^ 32)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float half_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// rotated bias key (24) | bias value (24) of one head, computed once per workgroup
__device__ __forceinline__ void fill_bias(float* sb, const float* bias_k, const float* bias_v, const float* inv_freq, int hd,
                                          int len) {
    const int d = threadIdx.x;
    if (d < kDH) {
        const int i = d % 12;
        const float ang = (float)len * inv_freq[i];
        const float c = cosf(ang), s = sinf(ang);
        const float x1 = bias_k[hd * kDH + i], x2 = bias_k[hd * kDH + i + 12];
        sb[d] = d < 12 ? x1 * c - x2 * s : x2 * c + x1 * s;
        sb[kDH + d] = bias_v[hd * kDH + d];
    }
}

struct Own {   // one owned row as an MFMA B operand: features 8 hh .. 8 hh + 7 | 16 + 8 hh .. (zeros past feature 23)
    bf16x8 f0, f1;
};
// `row` points at the row's 24 floats (global or LDS); null: a zero row
template <typename P>
__device__ __forceinline__ Own own_row(P row, int hh, float* part_dot = nullptr, const float* other = nullptr, float scale = 1.0f) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a, c = a, d = a;
    if (row) {
        a = *reinterpret_cast<const f32x4*>(row + 8 * hh);
        b = *reinterpret_cast<const f32x4*>(row + 8 * hh + 4);
        if (hh == 0) {
            c = *reinterpret_cast<const f32x4*>(row + 16);
            d = *reinterpret_cast<const f32x4*>(row + 20);
        }
    }
    if (part_dot) {   // this lane's share of row . other (the partner lane holds the rest)
        const f32x4 oa = *reinterpret_cast<const f32x4*>(other + 8 * hh), ob = *reinterpret_cast<const f32x4*>(other + 8 * hh + 4);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += a[i] * oa[i] + b[i] * ob[i];
        if (hh == 0) {
            const f32x4 oc = *reinterpret_cast<const f32x4*>(other + 16), od = *reinterpret_cast<const f32x4*>(other + 20);
#pragma unroll
            for (int i = 0; i < 4; ++i) s += c[i] * oc[i] + d[i] * od[i];
        }
        *part_dot = s;
    }
    if (scale != 1.0f) {   // (compile-time constant at every call site)
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] *= scale; b[i] *= scale; c[i] *= scale; d[i] *= scale; }
    }
    Own o;
    o.f0 = __builtin_bit_cast(bf16x8, u32x4{pack_bf16(a[0], a[1]), pack_bf16(a[2], a[3]), pack_bf16(b[0], b[1]), pack_bf16(b[2], b[3])});
    o.f1 = __builtin_bit_cast(bf16x8, u32x4{pack_bf16(c[0], c[1]), pack_bf16(c[2], c[3]), pack_bf16(d[0], d[1]), pack_bf16(d[2], d[3])});
    return o;
}

// Accumulator registers 0..7 / 8..15 packed to the two B operands of the next product.
__device__ __forceinline__ void chain(const f32x16& v, bf16x8& b0, bf16x8& b1) {
    b0 = __builtin_bit_cast(bf16x8, u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])});
    b1 = __builtin_bit_cast(bf16x8, u32x4{pack_bf16(v[8], v[9]), pack_bf16(v[10], v[11]), pack_bf16(v[12], v[13]), pack_bf16(v[14], v[15])});
}

// One value per accumulator register, taken from a per-row LDS table of the tile (row of register r = mfma_row(r, hh)).
__device__ __forceinline__ f32x16 rows_from(const float* tab, int hh, float sign) {
    f32x16 v;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(tab + 8 * g + 4 * hh);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[4 * g + i] = sign * t[i];
    }
    return v;
}

__device__ __forceinline__ bf16x8 lds_frag(const unsigned char* p) { return *reinterpret_cast<const bf16x8*>(p); }

// The streamed side: thread `tid` < 192 carries rows (2 pair, 2 pair + 1) x features (4 quad .. 4 quad + 3) of the chunk for
// up to two source matrices; threads 192..255 carry one per-row scalar (or two) each.
struct Item {
    int pair, quad;
    bool on;      // threads 0 .. 191: a (row pair, feature quad) of the chunk
    bool scalar;  // threads 192 .. 255: one per-row scalar (or two); threads past 255 (eight-wave workgroups) stage nothing
};
__device__ __forceinline__ Item item_of(int tid) { return Item{tid / 6, tid % 6, tid < 192, tid >= 192 && tid < 256}; }

__device__ __forceinline__ void put_rowmajor(unsigned char* tile, const Item& it, const f32x4 (&v)[2]) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
        *reinterpret_cast<u32x2*>(tile + (2 * it.pair + e) * kRowB + it.quad * 8) =
            u32x2{pack_bf16(v[e][0], v[e][1]), pack_bf16(v[e][2], v[e][3])};
}
__device__ __forceinline__ void put_transposed(unsigned char* tile, const Item& it, const f32x4 (&v)[2]) {
    const int rho = 2 * it.pair;
    const int col = (rho >> 5) * 64 + perm_pos(rho & 31) * 2;   // rows rho, rho + 1 are neighbours in the permuted order
#pragma unroll
    for (int i = 0; i < 4; ++i)
        *reinterpret_cast<uint32_t*>(tile + (4 * it.quad + i) * kTrB + col) = pack_bf16(v[0][i], v[1][i]);
}
// zero what the fills never write: features 24..31 of the row-major tiles, feature rows 24..31 of the transposed ones
__device__ __forceinline__ void clear_rowmajor(unsigned char* tile) {
    for (int e = threadIdx.x; e < kChunk; e += blockDim.x) *reinterpret_cast<u32x4*>(tile + e * kRowB + 48) = u32x4{0, 0, 0, 0};
}
__device__ __forceinline__ void clear_transposed(unsigned char* tile) {
    for (int e = threadIdx.x; e < 8 * (kTrB / 16); e += blockDim.x) *reinterpret_cast<u32x4*>(tile + 24 * kTrB + e * 16) = u32x4{0, 0, 0, 0};
}

// Inverse RoPE (and a scale) of a gradient held as D[feature][own row] -- what k32_rope_bwd does in a pass of its own
// (mha.py:356-357 backward; the q gradient also takes the q scale).  Lane (row, hh) holds features 8 gq + 4 hh + i in
// v[4 gq + i] (gq < 3); the rotation pairs feature f < 12 with f + 12, which sits in lane ^ 32: three exchanges of four
// registers (hh 0 sends group e, hh 1 group (e + 1) % 3), then y1 c + y2 s for the lower member, y2 c - y1 s for the upper.
// `pos` = the row's position in its sequence (what k32_rope rotates by: the axis index of the token).
__device__ __forceinline__ void unrope(f32x16& v, int pos, const float* __restrict__ inv_freq, int hh, float scale) {
    float recv[3][4];
#pragma unroll
    for (int e = 0; e < 3; ++e)
#pragma unroll
        for (int i = 0; i < 4; ++i) recv[e][i] = __shfl_xor(hh ? v[4 * ((e + 1) % 3) + i] : v[4 * e + i], 32, 64);
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const int base = e == 0 ? 0 : e == 1 ? 8 : 4;           // pair indices base .. base + 3
        const int grp = hh ? (e + 1) % 3 : e;                   // this lane's register group of the exchange
        const bool lower = hh ? e == 2 : e < 2;                 // it holds the pair's lower member (feature < 12)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float ang = (float)pos * inv_freq[base + i];
            const float c = cosf(ang), sn = sinf(ang);
            const float own = v[4 * grp + i], oth = recv[e][i];
            v[4 * grp + i] = (lower ? own * c + oth * sn : own * c - oth * sn) * scale;
        }
    }
}

struct Wg {   // which rows this workgroup / wave owns
    int seq, hd, blk;
};
// Workgroup b runs on XCD b % 8 (hardware round robin), and each XCD has its own L2: ALL workgroups of a sequence -- 16 heads x nblk
// row blocks, every one of which streams the whole sequence's rows of the [token][1152] buffer, 96 bytes per row and head -- are
// dealt to ONE XCD, so that a row leaves HBM once.  (With consecutive workgroups per sequence the 32 workgroups of an ATLAS
// sequence sat on all eight XCDs and the PMC counters showed 1.3 GiB read per launch for 0.5 GiB of operands.)
// The grid is padded to a multiple of eight sequences: seq >= nseq exits (wg_grid).
__device__ __forceinline__ Wg wg_of(int nblk) {
    const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3, per_seq = nblk * kH;
    const int within = rest % per_seq;
    return Wg{(rest / per_seq) * 8 + xcd, within / nblk, within % nblk};
}

}  // namespace

// ---- forward -------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void k16_attn(const float* __restrict__ qkv, int ld, AxisMap ax, MaskMap mk,
                                                   const float* __restrict__ bias_k, const float* __restrict__ bias_v,
                                                   const float* __restrict__ inv_freq, float* __restrict__ out,
                                                   float* __restrict__ lse_out) {
    __shared__ __attribute__((aligned(16))) unsigned char sK[kChunk * kRowB];
    __shared__ __attribute__((aligned(16))) unsigned char sVt[32 * kTrB];
    __shared__ __attribute__((aligned(16))) float sM[kChunk];
    __shared__ __attribute__((aligned(16))) float sB[2 * kDH];
    const int len = ax.len, tid = threadIdx.x, lane = lane_id(), l31 = lane & 31, hh = lane >> 5;
    const Wg g = wg_of((len + NW * 32 - 1) / (NW * 32));
    if (g.seq >= ax.nseq) return;
    fill_bias(sB, bias_k, bias_v, inv_freq, g.hd, len);
    clear_rowmajor(sK);
    clear_transposed(sVt);
    const int qi = g.blk * NW * 32 + wave_id() * 32 + l31;
    const long qtok = ax.token(g.seq, qi < len ? qi : len - 1);
    const Own q = own_row(qkv + qtok * ld + g.hd * kDH, hh);
    const Item it = item_of(tid);
    f32x4 kf[2], vf[2];
    float mval = 0.f;
    auto fetch = [&](int c0) {
        if (it.on) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int j = c0 + 2 * it.pair + e;
                kf[e] = vf[e] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (j < len) {
                    const float* row = qkv + ax.token(g.seq, j) * ld + g.hd * kDH + 4 * it.quad;
                    kf[e] = *reinterpret_cast<const f32x4*>(row + kC);
                    vf[e] = *reinterpret_cast<const f32x4*>(row + 2 * kC);
                }
            }
        } else if (it.scalar) {
            const int j = c0 + tid - 192;
            mval = j < len ? (mk.at(ax.token(g.seq, j)) != 0.f ? 0.f : kMasked) : (j == len ? 0.f : kMasked);
        }
    };
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = opaque_zero();
    float mrun = kMasked, den = 0.f;
    fetch(0);
    for (int c0 = 0; c0 < len + 1; c0 += kChunk) {
        A16_BARRIER();
        if (!A16_STAGE(c0)) {
        } else if (it.on) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (c0 + 2 * it.pair + e == len) {   // the bias key / value (LDS table, ready after the first barrier)
                    kf[e] = *reinterpret_cast<const f32x4*>(sB + 4 * it.quad);
                    vf[e] = *reinterpret_cast<const f32x4*>(sB + kDH + 4 * it.quad);
                }
            put_rowmajor(sK, it, kf);
            put_transposed(sVt, it, vf);
        } else if (it.scalar) {
            sM[tid - 192] = mval;
        }
        A16_BARRIER();
        if (A16_STAGE(c0 + kChunk) && c0 + kChunk < len + 1) fetch(c0 + kChunk);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (c0 + 32 * t > len) break;
            f32x16 s = rows_from(sM + 32 * t, hh, 1.0f);
            const unsigned char* kr = sK + (32 * t + l31) * kRowB + hh * 16;
            s = A16_MMA(lds_frag(kr), q.f0, s);
            s = A16_MMA(lds_frag(kr + 32), q.f1, s);
            float tmax = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
            const float mnew = fmaxf(mrun, half_max(tmax));
            const float alpha = A16_EXP2((mrun - mnew) * kLog2e);
            mrun = mnew;
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = A16_EXP2((s[r] - mnew) * kLog2e);
                sum += s[r];
                o[r] *= alpha;
            }
            den = den * alpha + sum;
            bf16x8 p0, p1;
            chain(s, p0, p1);
            const unsigned char* vr = sVt + l31 * kTrB + 64 * t + hh * 16;
            o = A16_MMA(lds_frag(vr), p0, o);
            o = A16_MMA(lds_frag(vr + 32), p1, o);
        }
    }
    den = half_sum(den);
    if (qi >= len) return;
    const float inv = 1.0f / den;
    float* dst = out + qtok * kC + g.hd * kDH + 4 * hh;
#pragma unroll
    for (int gq = 0; gq < 3; ++gq)
        *reinterpret_cast<f32x4*>(dst + 8 * gq) = f32x4{o[4 * gq] * inv, o[4 * gq + 1] * inv, o[4 * gq + 2] * inv, o[4 * gq + 3] * inv};
    if (lse_out && hh == 0) lse_out[qtok * kH + g.hd] = mrun + logf(den);
}

// ---- backward, query pass ------------------------------------------------------------------------------------------
// dq -- already taken back through RoPE and the q scale (no k32_rope_bwd pass after these kernels) -- into dqkv[:, 0:384];
// (lse, delta) into stats[token][head][2] for the key pass.
template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void k16_attn_bwd_q(const float* __restrict__ qkv, int ld, AxisMap ax, MaskMap mk,
                                                         const float* __restrict__ bias_k, const float* __restrict__ bias_v,
                                                         const float* __restrict__ inv_freq, const float* __restrict__ o,
                                                         const float* __restrict__ dout, float* __restrict__ dqkv,
                                                         float* __restrict__ stats, const float* __restrict__ lse_in) {
    __shared__ __attribute__((aligned(16))) unsigned char sK[kChunk * kRowB];
    __shared__ __attribute__((aligned(16))) unsigned char sV[kChunk * kRowB];
    __shared__ __attribute__((aligned(16))) unsigned char sKt[32 * kTrB];
    __shared__ __attribute__((aligned(16))) float sM[kChunk];
    __shared__ __attribute__((aligned(16))) float sB[2 * kDH];
    const int len = ax.len, tid = threadIdx.x, lane = lane_id(), l31 = lane & 31, hh = lane >> 5;
    const Wg g = wg_of((len + NW * 32 - 1) / (NW * 32));
    if (g.seq >= ax.nseq) return;
    fill_bias(sB, bias_k, bias_v, inv_freq, g.hd, len);
    clear_rowmajor(sK);
    clear_rowmajor(sV);
    clear_transposed(sKt);
    const int qi = g.blk * NW * 32 + wave_id() * 32 + l31;
    const long qtok = ax.token(g.seq, qi < len ? qi : len - 1);
    const Own q = own_row(qkv + qtok * ld + g.hd * kDH, hh);
    float delta;
    const Own dO = own_row(dout + qtok * kC + g.hd * kDH, hh, &delta, o + qtok * kC + g.hd * kDH);
    delta = half_sum(delta);
    const float lse = lse_in[qtok * kH + g.hd];
    const Item it = item_of(tid);
    f32x4 kf[2], vf[2];
    float mval = 0.f;
    auto fetch = [&](int c0) {
        if (it.on) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int j = c0 + 2 * it.pair + e;
                kf[e] = vf[e] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (j < len) {
                    const float* row = qkv + ax.token(g.seq, j) * ld + g.hd * kDH + 4 * it.quad;
                    kf[e] = *reinterpret_cast<const f32x4*>(row + kC);
                    vf[e] = *reinterpret_cast<const f32x4*>(row + 2 * kC);
                }
            }
        } else if (it.scalar) {
            const int j = c0 + tid - 192;
            mval = j < len ? (mk.at(ax.token(g.seq, j)) != 0.f ? 0.f : kMasked) : (j == len ? 0.f : kMasked);
        }
    };
    f32x16 dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[r] = opaque_zero();
    fetch(0);
    for (int c0 = 0; c0 < len + 1; c0 += kChunk) {
        A16_BARRIER();
        if (!A16_STAGE(c0)) {
        } else if (it.on) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (c0 + 2 * it.pair + e == len) {
                    kf[e] = *reinterpret_cast<const f32x4*>(sB + 4 * it.quad);
                    vf[e] = *reinterpret_cast<const f32x4*>(sB + kDH + 4 * it.quad);
                }
            put_rowmajor(sK, it, kf);
            put_rowmajor(sV, it, vf);
            put_transposed(sKt, it, kf);
        } else if (it.scalar) {
            sM[tid - 192] = mval;
        }
        A16_BARRIER();
        if (A16_STAGE(c0 + kChunk) && c0 + kChunk < len + 1) fetch(c0 + kChunk);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (c0 + 32 * t > len) break;
            f32x16 s = rows_from(sM + 32 * t, hh, 1.0f);
            f32x16 dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[r] = -delta;
            const unsigned char* kr = sK + (32 * t + l31) * kRowB + hh * 16;
            const unsigned char* vr = sV + (32 * t + l31) * kRowB + hh * 16;
            s = A16_MMA(lds_frag(kr), q.f0, s);
            s = A16_MMA(lds_frag(kr + 32), q.f1, s);
            dp = A16_MMA(lds_frag(vr), dO.f0, dp);
            dp = A16_MMA(lds_frag(vr + 32), dO.f1, dp);
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = A16_EXP2((s[r] - lse) * kLog2e) * dp[r];
            bf16x8 d0, d1;
            chain(s, d0, d1);
            const unsigned char* tr = sKt + l31 * kTrB + 64 * t + hh * 16;
            dq = A16_MMA(lds_frag(tr), d0, dq);
            dq = A16_MMA(lds_frag(tr + 32), d1, dq);
        }
    }
    unrope(dq, qi < len ? qi : len - 1, inv_freq, hh, 0.20412414523193151f);   // back through RoPE and the q scale 24^-1/2
    if (qi >= len) return;
    float* dst = dqkv + qtok * ld + g.hd * kDH + 4 * hh;
#pragma unroll
    for (int gq = 0; gq < 3; ++gq)
        *reinterpret_cast<f32x4*>(dst + 8 * gq) = f32x4{dq[4 * gq], dq[4 * gq + 1], dq[4 * gq + 2], dq[4 * gq + 3]};
    if (hh == 0) {
        stats[(qtok * kH + g.hd) * 2] = lse;
        stats[(qtok * kH + g.hd) * 2 + 1] = delta;
    }
}

// ---- backward, key pass --------------------------------------------------------------------------------------------
// Real keys write dqkv[:, 384:1152] (dk taken back through RoPE); the bias key writes dbias[seq][dk rotated back: head x 24 | dv: head x 24].
template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void k16_attn_bwd_kv(const float* __restrict__ qkv, int ld, AxisMap ax, MaskMap mk,
                                                          const float* __restrict__ bias_k, const float* __restrict__ bias_v,
                                                          const float* __restrict__ inv_freq, const float* __restrict__ dout,
                                                          const float* __restrict__ stats, float* __restrict__ dqkv,
                                                          float* __restrict__ dbias) {
    __shared__ __attribute__((aligned(16))) unsigned char sQ[kChunk * kRowB];
    __shared__ __attribute__((aligned(16))) unsigned char sdO[kChunk * kRowB];
    __shared__ __attribute__((aligned(16))) unsigned char sQt[32 * kTrB];
    __shared__ __attribute__((aligned(16))) unsigned char sdOt[32 * kTrB];
    __shared__ __attribute__((aligned(16))) float sLse[kChunk];
    __shared__ __attribute__((aligned(16))) float sDel[kChunk];
    __shared__ __attribute__((aligned(16))) float sB[2 * kDH];
    const int len = ax.len, tid = threadIdx.x, lane = lane_id(), l31 = lane & 31, hh = lane >> 5;
    const Wg g = wg_of((len + NW * 32) / (NW * 32));
    if (g.seq >= ax.nseq) return;
    fill_bias(sB, bias_k, bias_v, inv_freq, g.hd, len);
    clear_rowmajor(sQ);
    clear_rowmajor(sdO);
    clear_transposed(sQt);
    clear_transposed(sdOt);
    const int j = g.blk * NW * 32 + wave_id() * 32 + l31;
    const long ktok = ax.token(g.seq, j < len ? j : len - 1);
    const bool valid = j < len ? mk.at(ktok) != 0.f : j == len;
    const bool has_bias = g.blk * NW * 32 <= len && len < (g.blk + 1) * NW * 32;   // workgroup-uniform
    const Item it = item_of(tid);
    f32x4 qf[2], df[2];
    float lval = 0.f, dval = 0.f;
    auto fetch = [&](int c0) {
        if (it.on) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int i = c0 + 2 * it.pair + e;
                qf[e] = df[e] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (i < len) {
                    const long t = ax.token(g.seq, i);
                    qf[e] = *reinterpret_cast<const f32x4*>(qkv + t * ld + g.hd * kDH + 4 * it.quad);
                    df[e] = *reinterpret_cast<const f32x4*>(dout + t * kC + g.hd * kDH + 4 * it.quad);
                }
            }
        } else if (it.scalar) {
            const int i = c0 + tid - 192;
            const long t = ax.token(g.seq, i < len ? i : len - 1);
            lval = i < len ? stats[(t * kH + g.hd) * 2] : 3.0e38f;     // beyond len: p = exp(-inf) = 0
            dval = stats[(t * kH + g.hd) * 2 + 1];
        }
    };
    fetch(0);
    __syncthreads();   // sB
    Own k, v;
    if (j < len) {
        k = own_row(qkv + ktok * ld + kC + g.hd * kDH, hh);
        v = own_row(qkv + ktok * ld + 2 * kC + g.hd * kDH, hh);
    } else if (j == len) {
        k = own_row((const float*)sB, hh);
        v = own_row((const float*)(sB + kDH), hh);
    } else {
        k = own_row((const float*)nullptr, hh);
        v = k;
    }
    f32x16 dk, dv;
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[r] = dv[r] = opaque_zero();
    for (int c0 = 0; c0 < len; c0 += kChunk) {
        A16_BARRIER();
        if (!A16_STAGE(c0)) {
        } else if (it.on) {
            put_rowmajor(sQ, it, qf);
            put_rowmajor(sdO, it, df);
            put_transposed(sQt, it, qf);
            put_transposed(sdOt, it, df);
        } else if (it.scalar) {
            sLse[tid - 192] = lval;
            sDel[tid - 192] = dval;
        }
        A16_BARRIER();
        if (A16_STAGE(c0 + kChunk) && c0 + kChunk < len) fetch(c0 + kChunk);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (c0 + 32 * t >= len) break;
            f32x16 s = rows_from(sLse + 32 * t, hh, -1.0f);
            f32x16 dp = rows_from(sDel + 32 * t, hh, -1.0f);
            const unsigned char* qr = sQ + (32 * t + l31) * kRowB + hh * 16;
            const unsigned char* dr = sdO + (32 * t + l31) * kRowB + hh * 16;
            s = A16_MMA(lds_frag(qr), k.f0, s);
            s = A16_MMA(lds_frag(qr + 32), k.f1, s);
            dp = A16_MMA(lds_frag(dr), v.f0, dp);
            dp = A16_MMA(lds_frag(dr + 32), v.f1, dp);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = valid ? A16_EXP2(s[r] * kLog2e) : 0.f;
                dp[r] *= s[r];
            }
            bf16x8 p0, p1, d0, d1;
            chain(s, p0, p1);
            chain(dp, d0, d1);
            const unsigned char* dt = sdOt + l31 * kTrB + 64 * t + hh * 16;
            const unsigned char* qt = sQt + l31 * kTrB + 64 * t + hh * 16;
            dv = A16_MMA(lds_frag(dt), p0, dv);
            dv = A16_MMA(lds_frag(dt + 32), p1, dv);
            dk = A16_MMA(lds_frag(qt), d0, dk);
            dk = A16_MMA(lds_frag(qt + 32), d1, dk);
        }
    }
    f32x16 dku = dk;    // real keys: back through RoPE at their position (the bias key's own path below keeps dk)
    unrope(dku, j < len ? j : 0, inv_freq, hh, 1.0f);
    if (j < len) {
        float* dst = dqkv + ktok * ld + kC + g.hd * kDH + 4 * hh;
#pragma unroll
        for (int gq = 0; gq < 3; ++gq) {
            *reinterpret_cast<f32x4*>(dst + 8 * gq) = f32x4{dku[4 * gq], dku[4 * gq + 1], dku[4 * gq + 2], dku[4 * gq + 3]};
            *reinterpret_cast<f32x4*>(dst + kC + 8 * gq) = f32x4{dv[4 * gq], dv[4 * gq + 1], dv[4 * gq + 2], dv[4 * gq + 3]};
        }
    }
    if (!has_bias) return;
    // bias key: its two lanes drop (dk | dv) into the table, 24 threads undo the rotation (position len) and write the
    // per-sequence row that is summed over sequences afterwards
    __syncthreads();
    if (j == len) {
#pragma unroll
        for (int gq = 0; gq < 3; ++gq)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sB[8 * gq + 4 * hh + i] = dk[4 * gq + i];
                sB[kDH + 8 * gq + 4 * hh + i] = dv[4 * gq + i];
            }
    }
    __syncthreads();
    float* dst = dbias + (long)g.seq * 2 * kC + g.hd * kDH;
    if (tid < 12) {
        const float ang = (float)len * inv_freq[tid];
        const float c = cosf(ang), s = sinf(ang);
        dst[tid] = sB[tid] * c + sB[tid + 12] * s;            // R(-theta): d x1 = d y1 c + d y2 s
        dst[tid + 12] = sB[tid + 12] * c - sB[tid] * s;       //            d x2 = d y2 c - d y1 s
    } else if (tid >= 32 && tid < 32 + kDH) {
        dst[kC + tid - 32] = sB[kDH + tid - 32];
    }
}

// ---- backward, one workgroup per (sequence, head) and pass --------------------------------------------------------------
// Axes of 129 .. 256 positions (the ATLAS training shapes: 250 frames x 256 residues): the chunked kernels above stage the
// other side of every 128-row block through LDS again (five fills of the sequence per (sequence, head) over the two passes,
// two barriers per 64-row chunk, global loads inside the loop).  Here a workgroup of eight waves owns ALL rows of one
// (sequence, head) -- queries in k16_attn_bwd_seq_q, keys in k16_attn_bwd_seq_kv, one 32-row tile per wave -- and converts the
// whole other side ONCE: k, v row-major + k transposed (288 rows: the bias key may open a ninth tile; 66 KB), or q, dO
// row-major + transposed (256 rows; 78 KB); after that one fill the pass runs out of LDS alone (no barrier, no global load in the
// loop, the next tile's score MFMAs issued beside the current tile's exponentials), two workgroups per CU so that one's
// fill / result stores overlap the other's pass.  (A first form kept both sides in one 147 KB workgroup: its fill, its passes
// and its stores each took a third of the launch and nothing overlapped them: profiles/r06_experiments.txt #18.)
// Same products, the same chained-accumulator layout as the kernels above; what differs:
//   * scores in log2 units: q is multiplied by log2(e) BEFORE it is rounded to bf16 (forward k16_attn_seq rounds the same
//     product), so p = 2^s needs no multiply; dk is multiplied by ln 2 in its store stage;
//   * per-row addends ride in the padding features 24 .. 27 of the operands (d = 24 of 32): -lse2 and -delta as three bf16
//     each in the own q / dO operand against 1.0 in every K / V row (query pass), key validity 0 / -3e38 in the own k operand
//     against 1.0 in every Q row (key pass): the exponent's argument comes out of the MFMA as it is;
//   * a bias key that opens a tile of its own (len = 256) is shared out: wave w takes query tile w against it and the eight
//     partial (dk | dv) rows are summed in wave order (deterministic).
namespace {
constexpr int kSeqQ = 256, kSeqK = 288;
constexpr int kTrQB = kSeqQ * 2 + 16, kTrKB = kSeqK * 2 + 16;   // bytes of one feature row of a transposed (Q | K) image
constexpr float kLn2 = 0.6931471805599453f;
constexpr int kSeqLds = 2 * kSeqQ * kRowB + 2 * 32 * kTrQB + (2 * kSeqQ + 2 * kDH + 8 * 2 * kDH + 12) * 4;   // the key pass's images: 78 576 B, two workgroups per CU

__device__ __forceinline__ void put_tr(unsigned char* img, int stride, int rho, int quad, const f32x4& a, const f32x4& b) {
    const int col = (rho >> 5) * 64 + perm_pos(rho & 31) * 2;   // rows rho, rho + 1 (rho even) are neighbours in the permuted order
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint32_t*>(img + (4 * quad + i) * stride + col) = pack_bf16(a[i], b[i]);
}
__device__ __forceinline__ void put_rm(unsigned char* img, int row, int quad, const f32x4& v) {
    *reinterpret_cast<u32x2*>(img + row * kRowB + quad * 8) = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
}
// x as three bf16 (24 significant bits) in features 24, 25, 26 of an own operand's k-step 1 (lanes hh == 1): against 1.0 in the
// other operand's rows the MFMA adds x to every product of the lane's column
__device__ __forceinline__ float round_bf16_(float x) { return __uint_as_float(pack_bf16(x, 0.f) << 16); }
__device__ __forceinline__ void ride3(bf16x8& f, float x) {
    const float hi = round_bf16_(x), mid = round_bf16_(x - hi), lo = round_bf16_(x - hi - mid);
    f[0] = (__bf16)hi;
    f[1] = (__bf16)mid;
    f[2] = (__bf16)lo;
}
// unrope() with the hardware sine / cosine (input in revolutions): 3e-5 rad at position 256, far inside the bf16 operands' error
__device__ __forceinline__ void unrope_fast(f32x16& v, int pos, const float* __restrict__ inv_freq, int hh, float scale) {
    float recv[3][4];
#pragma unroll
    for (int e = 0; e < 3; ++e)
#pragma unroll
        for (int i = 0; i < 4; ++i) recv[e][i] = __shfl_xor(hh ? v[4 * ((e + 1) % 3) + i] : v[4 * e + i], 32, 64);
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const int base = e == 0 ? 0 : e == 1 ? 8 : 4;
        const int grp = hh ? (e + 1) % 3 : e;
        const bool lower = hh ? e == 2 : e < 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float rev = __builtin_amdgcn_fractf((float)pos * (inv_freq[base + i] * 0.15915494309189535f));
            const float c = __builtin_amdgcn_cosf(rev), sn = __builtin_amdgcn_sinf(rev);
            const float own = v[4 * grp + i], oth = recv[e][i];
            v[4 * grp + i] = (lower ? own * c + oth * sn : own * c - oth * sn) * scale;
        }
    }
}
// This lane's twelve values of a gradient row (features 8 gq + 4 hh + i in v[4 gq + i]) at element offset e0 of the q | k | v gradient
// buffer: fp32, or rounded to bf16 (out16: the buffer is bf16 rows -- it is only ever the token operand of the dX product and dY of
// the weight gradient, which round it to bf16 anyway)
__device__ __forceinline__ void store_grad(float* base, long e0, const f32x16& v, bool out16) {
    if (out16) {
        unsigned short* b16 = reinterpret_cast<unsigned short*>(base);
#pragma unroll
        for (int gq = 0; gq < 3; ++gq)
            *reinterpret_cast<u32x2*>(b16 + e0 + 8 * gq) = u32x2{pack_bf16(v[4 * gq], v[4 * gq + 1]), pack_bf16(v[4 * gq + 2], v[4 * gq + 3])};
    } else {
#pragma unroll
        for (int gq = 0; gq < 3; ++gq)
            *reinterpret_cast<f32x4*>(base + e0 + 8 * gq) = f32x4{v[4 * gq], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]};
    }
}
// the 16 bytes of padding features 24 .. 31 of every row of a row-major image
__device__ __forceinline__ void pad_rows(unsigned char* img, int nrows, u32x4 v) {
    for (int e = threadIdx.x; e < nrows; e += 512) *reinterpret_cast<u32x4*>(img + e * kRowB + 48) = v;
}
// RoPE inside the kernels (launch argument `rope`: q, k arrive unrotated and k32_rope is not launched; mha.py:356-357).  sF = the
// twelve frequencies in revolutions per position; hardware sine / cosine as in unrope_fast.
// One row's features 4 qq .. 4 qq + 3 (lo) and 12 + 4 qq .. (hi): the two halves of four rotation pairs.
__device__ __forceinline__ void rope_pair(f32x4& lo, f32x4& hi, int pos, int qq, const float* sF) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float rev = __builtin_amdgcn_fractf((float)pos * sF[4 * qq + c]);
        const float cs = __builtin_amdgcn_cosf(rev), sn = __builtin_amdgcn_sinf(rev);
        const float x1 = lo[c], x2 = hi[c];
        lo[c] = x1 * cs - x2 * sn;
        hi[c] = x2 * cs + x1 * sn;
    }
}
// An own row (24 floats in global memory) rotated at `pos`, scaled, as an MFMA B operand.
__device__ __forceinline__ Own own_row_rope(const float* row, int pos, int hh, const float* sF, float scale) {
    float x[24], y[24];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * c);
#pragma unroll
        for (int i = 0; i < 4; ++i) x[4 * c + i] = v[i];
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const float rev = __builtin_amdgcn_fractf((float)pos * sF[i]);
        const float cs = __builtin_amdgcn_cosf(rev), sn = __builtin_amdgcn_sinf(rev);
        y[i] = (x[i] * cs - x[i + 12] * sn) * scale;
        y[i + 12] = (x[i + 12] * cs + x[i] * sn) * scale;
    }
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = hh ? y[8 + i] : y[i];
        b[i] = hh ? 0.f : y[16 + i];
    }
    Own o;
    o.f0 = __builtin_bit_cast(bf16x8, u32x4{pack_bf16(a[0], a[1]), pack_bf16(a[2], a[3]), pack_bf16(a[4], a[5]), pack_bf16(a[6], a[7])});
    o.f1 = __builtin_bit_cast(bf16x8, u32x4{pack_bf16(b[0], b[1]), pack_bf16(b[2], b[3]), pack_bf16(b[4], b[5]), pack_bf16(b[6], b[7])});
    return o;
}
__device__ __forceinline__ void fill_freq(float* sF, const float* __restrict__ inv_freq) {
    if (threadIdx.x >= 64 && threadIdx.x < 76) sF[threadIdx.x - 64] = inv_freq[threadIdx.x - 64] * 0.15915494309189535f;
}
__device__ __forceinline__ void zero_tr_tail(unsigned char* img, int stride) {   // feature rows 24 .. 31 of a transposed image
    for (int e = threadIdx.x; e < 8 * (stride / 16); e += 512) *reinterpret_cast<u32x4*>(img + 24 * stride + e * 16) = u32x4{0, 0, 0, 0};
}
}  // namespace

// query pass: dq (taken back through RoPE and the q scale) into dqkv[:, 0:384]
__device__ __forceinline__ void attn_bwd_seq_q(unsigned char* lds, const Wg g, const float* __restrict__ qkv, int ld, const AxisMap& ax,
                                               const MaskMap& mk, const float* __restrict__ bias_k, const float* __restrict__ bias_v,
                                               const float* __restrict__ inv_freq, const float* __restrict__ o,
                                               const float* __restrict__ dout, float* __restrict__ dqkv,
                                               const float* __restrict__ lse_in, bool rope, bool out16) {
    unsigned char *sK = lds, *sV = sK + kSeqK * kRowB, *sKt = sV + kSeqK * kRowB;
    float *sM = reinterpret_cast<float*>(sKt + 32 * kTrKB), *sB = sM + kSeqK, *sF = sB + 2 * kDH;
    static_assert(2 * kSeqK * kRowB + 32 * kTrKB + (kSeqK + 2 * kDH + 12) * 4 <= kSeqLds, "query pass: LDS");
    const int len = ax.len, tid = threadIdx.x, lane = lane_id(), l31 = lane & 31, hh = lane >> 5, w = wave_id();
    const int nkt = len / 32 + 1, nqt = (len + 31) / 32;   // key tiles incl. the bias key's, query tiles
    const long tok0 = ax.token(g.seq, 0);
    const long pstr = ax.pos_stride;
    fill_bias(sB, bias_k, bias_v, inv_freq, g.hd, len);
    fill_freq(sF, inv_freq);
    pad_rows(sK, 32 * nkt, u32x4{0x3F803F80u, 0x00003F80u, 0, 0});   // 1.0 in features 24, 25, 26: -lse2 rides against them
    pad_rows(sV, 32 * nkt, u32x4{0x3F803F80u, 0x00003F80u, 0, 0});   //                             -delta
    zero_tr_tail(sKt, kTrKB);
    for (int r = tid; r < 32 * nkt; r += 512)
        sM[r] = r < len ? (mk.at(tok0 + (long)r * pstr) != 0.f ? 0.f : kMasked) : (r == len ? 0.f : kMasked);
    // own rows straight from global memory (fp32 -> bf16 in registers), delta = dO . O on the way
    const int own = 32 * w + l31;
    const long qtok = tok0 + (long)(own < len ? own : len - 1) * pstr;
    Own q, dO;
    float delta = 0.f, lse2 = -kMasked;
    if (w < nqt) {
        if (!rope) q = own_row(qkv + qtok * ld + g.hd * kDH, hh, nullptr, nullptr, kLog2e);   // scores in log2 units
        dO = own_row(dout + qtok * kC + g.hd * kDH, hh, &delta, o + qtok * kC + g.hd * kDH);
        delta = half_sum(delta);
        if (own < len) lse2 = kLog2e * lse_in[qtok * kH + g.hd];   // beyond len: p = 2^-inf = 0
    }
    __syncthreads();   // sB, sF
    if (w < nqt) {
        if (rope) q = own_row_rope(qkv + qtok * ld + g.hd * kDH, own < len ? own : len - 1, hh, sF, kLog2e);
        if (hh) {   // features 24 .. 31 of the own operands: -lse2 and -delta ride against the 1.0s of the K / V rows
            ride3(q.f1, -lse2);
            ride3(dO.f1, -delta);
        }
    }
    // the fill: item = (row pair, feature quads qq and qq + 3) -- both halves of its rotation pairs in one thread, two neighbouring
    // rows for the packed transposed stores; 16 nkt * 3 <= 432 items
    if (tid < 16 * nkt * 3) {
        const int pair = tid / 3, qq = tid - 3 * pair, r0 = 2 * pair;
        f32x4 kf[2][2], vf[2][2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int r = r0 + x;
            kf[x][0] = kf[x][1] = vf[x][0] = vf[x][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r < len && !A16_SEQ_NOLOAD) {
                const float* row = qkv + (tok0 + (long)r * pstr) * ld + g.hd * kDH + 4 * qq;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    kf[x][h] = *reinterpret_cast<const f32x4*>(row + kC + 12 * h);
                    vf[x][h] = *reinterpret_cast<const f32x4*>(row + 2 * kC + 12 * h);
                }
                if (rope) rope_pair(kf[x][0], kf[x][1], r, qq, sF);
            } else if (r == len) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    kf[x][h] = *reinterpret_cast<const f32x4*>(sB + 4 * qq + 12 * h);
                    vf[x][h] = *reinterpret_cast<const f32x4*>(sB + kDH + 4 * qq + 12 * h);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                put_rm(sK, r0 + x, qq + 3 * h, kf[x][h]);
                put_rm(sV, r0 + x, qq + 3 * h, vf[x][h]);
            }
            put_tr(sKt, kTrKB, r0, qq + 3 * h, kf[0][h], kf[1][h]);
        }
    }
    __syncthreads();
    if (w >= nqt) return;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[r] = opaque_zero();
    // software pipeline over the key tiles: the block of tile t issues the score / dP MFMAs of tile t + 1 beside its own
    // exponentials, then its dq MFMAs (the last block recomputes its own scores: discarded)
    auto scores = [&](int t, f32x16& s, f32x16& dp) {
        s = rows_from(sM + 32 * t, hh, 1.0f);
        const unsigned char* kr = sK + (32 * t + l31) * kRowB + hh * 16;
        const unsigned char* vr = sV + (32 * t + l31) * kRowB + hh * 16;
        const bf16x8 k0 = lds_frag(kr), k1 = lds_frag(kr + 32), v0 = lds_frag(vr), v1 = lds_frag(vr + 32);
        s = A16_MMA(k0, q.f0, s);
        dp = A16_MMA(v0, dO.f0, zero);
        s = A16_MMA(k1, q.f1, s);
        dp = A16_MMA(v1, dO.f1, dp);
        asm volatile("" ::"v"(v0), "v"(v1), "v"(dO.f0), "v"(dO.f1));   // sources outlive the inline-constant-C MFMA (common.h opaque_zero)
    };
    f32x16 s, dp;
    scores(0, s, dp);
    for (int t = 0; t < (A16_SEQ_NOLOOP ? 1 : nkt); ++t) {
        const unsigned char* tr = sKt + l31 * kTrKB + 64 * t + hh * 16;
        const bf16x8 t0 = lds_frag(tr), t1 = lds_frag(tr + 32);
        f32x16 sn, dpn;
        scores(t + 1 < nkt ? t + 1 : t, sn, dpn);
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = A16_EXP2(s[r]) * dp[r];   // p = 2^(s2 - lse2),  d s = p (d p - delta)
        bf16x8 d0, d1;
        chain(s, d0, d1);
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x402, 10, 0);  // ten VALU / transcendental
        }
        dq = A16_MMA(t0, d0, dq);
        dq = A16_MMA(t1, d1, dq);
        s = sn;
        dp = dpn;
    }
    unrope_fast(dq, own < len ? own : len - 1, inv_freq, hh, 0.20412414523193151f);   // back through RoPE and the q scale 24^-1/2
    if (own < len && !A16_SEQ_NOSTORE) store_grad(dqkv, qtok * ld + g.hd * kDH + 4 * hh, dq, out16);
}

// key pass: real keys write dqkv[:, 384:1152] (dk taken back through RoPE); the bias key writes dbias[seq][dk rotated back | dv]
__device__ __forceinline__ void attn_bwd_seq_kv(unsigned char* lds, const Wg g, const float* __restrict__ qkv, int ld, const AxisMap& ax,
                                                const MaskMap& mk, const float* __restrict__ bias_k, const float* __restrict__ bias_v,
                                                const float* __restrict__ inv_freq, const float* __restrict__ o,
                                                const float* __restrict__ dout, float* __restrict__ dqkv,
                                                float* __restrict__ dbias, const float* __restrict__ lse_in, bool rope, bool out16) {
    unsigned char *sQ = lds, *sdO = sQ + kSeqQ * kRowB, *sQt = sdO + kSeqQ * kRowB, *sdOt = sQt + 32 * kTrQB;
    float *sLse = reinterpret_cast<float*>(sdOt + 32 * kTrQB), *sDel = sLse + kSeqQ, *sB = sDel + kSeqQ, *sRed = sB + 2 * kDH,
          *sF = sRed + 8 * 2 * kDH;
    const int len = ax.len, tid = threadIdx.x, lane = lane_id(), l31 = lane & 31, hh = lane >> 5, w = wave_id();
    const int nkt = len / 32 + 1, nqt = (len + 31) / 32;
    const long tok0 = ax.token(g.seq, 0);
    const long pstr = ax.pos_stride;
    fill_bias(sB, bias_k, bias_v, inv_freq, g.hd, len);
    fill_freq(sF, inv_freq);
    pad_rows(sQ, 32 * nqt, u32x4{0, 0x3F800000u, 0, 0});   // 1.0 in feature 27: the own key's validity rides against it
    pad_rows(sdO, 32 * nqt, u32x4{0, 0, 0, 0});
    zero_tr_tail(sQt, kTrQB);
    zero_tr_tail(sdOt, kTrQB);
    if (tid < 8 * 2 * kDH) sRed[tid] = 0.f;
    // per query row: -lse2 and -delta = -(dO . O)  (one thread per row; its 24 + 24 floats are lines the fill below reads too)
    if (tid < 32 * nqt) {
        float nl = kMasked, nd = 0.f;   // beyond len: p = 2^-inf = 0
        if (tid < len && !A16_SEQ_NOLOAD) {
            const long t = tok0 + (long)tid * pstr;
            nl = -kLog2e * lse_in[t * kH + g.hd];
            const float* dr = dout + t * kC + g.hd * kDH;
            const float* orow = o + t * kC + g.hd * kDH;
            float acc[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(dr + 4 * c), b = *reinterpret_cast<const f32x4*>(orow + 4 * c);
                acc[c] = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
            }
            nd = -(((acc[0] + acc[1]) + (acc[2] + acc[3])) + (acc[4] + acc[5]));
        }
        sLse[tid] = nl;
        sDel[tid] = nd;
    }
    if (rope) __syncthreads();   // sF
    if (tid < 16 * nqt * 3) {   // the fill: (row pair, feature quads qq and qq + 3) items of the query side (see the query pass)
        const int pair = tid / 3, qq = tid - 3 * pair, r0 = 2 * pair;
        f32x4 qf[2][2], df[2][2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int r = r0 + x;
            qf[x][0] = qf[x][1] = df[x][0] = df[x][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r < len && !A16_SEQ_NOLOAD) {
                const long t = tok0 + (long)r * pstr;
                const float* row = qkv + t * ld + g.hd * kDH + 4 * qq;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    qf[x][h] = *reinterpret_cast<const f32x4*>(row + 12 * h);
                    df[x][h] = *reinterpret_cast<const f32x4*>(dout + t * kC + g.hd * kDH + 4 * qq + 12 * h);
                }
                if (rope) rope_pair(qf[x][0], qf[x][1], r, qq, sF);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) qf[x][h][i] *= kLog2e;   // scores in log2 units (k16_attn_seq rounds the same product)
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                put_rm(sQ, r0 + x, qq + 3 * h, qf[x][h]);
                put_rm(sdO, r0 + x, qq + 3 * h, df[x][h]);
            }
            put_tr(sQt, kTrQB, r0, qq + 3 * h, qf[0][h], qf[1][h]);
            put_tr(sdOt, kTrQB, r0, qq + 3 * h, df[0][h], df[1][h]);
        }
    }
    __syncthreads();
    // (dk, dv) of key tile jt over the query tiles t0 .. t1 - 1
    auto key_pass = [&](int jt, int t0, int t1, f32x16& dk, f32x16& dv) {
        const int j = 32 * jt + l31;
        Own k, v;
        bool valid;
        if (j < len) {
            const long ktok = tok0 + (long)j * pstr;
            k = rope ? own_row_rope(qkv + ktok * ld + kC + g.hd * kDH, j, hh, sF, 1.0f) : own_row(qkv + ktok * ld + kC + g.hd * kDH, hh);
            v = own_row(qkv + ktok * ld + 2 * kC + g.hd * kDH, hh);
            valid = mk.at(ktok) != 0.f;
        } else if (j == len) {
            k = own_row((const float*)sB, hh);
            v = own_row((const float*)(sB + kDH), hh);
            valid = true;
        } else {
            k = own_row((const float*)nullptr, hh);
            v = k;
            valid = false;
        }
        if (hh) k.f1[3] = (__bf16)(valid ? 0.f : kMasked);   // feature 27, against the 1.0 of the Q rows: a padded key's scores are -3e38
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[r] = dv[r] = opaque_zero();
        auto scores = [&](int t, f32x16& s, f32x16& dp) {
            s = rows_from(sLse + 32 * t, hh, 1.0f);    // -lse2 of the tile's queries
            dp = rows_from(sDel + 32 * t, hh, 1.0f);   // -delta
            const unsigned char* qr = sQ + (32 * t + l31) * kRowB + hh * 16;
            const unsigned char* dr = sdO + (32 * t + l31) * kRowB + hh * 16;
            const bf16x8 q0 = lds_frag(qr), q1 = lds_frag(qr + 32), e0 = lds_frag(dr), e1 = lds_frag(dr + 32);
            s = A16_MMA(q0, k.f0, s);
            dp = A16_MMA(e0, v.f0, dp);
            s = A16_MMA(q1, k.f1, s);
            dp = A16_MMA(e1, v.f1, dp);
        };
        // (not software-pipelined: with the next tile's scores in flight the kernel needs 132 registers, one workgroup per CU; at
        // 128 the second workgroup's waves fill this wave's waits)
        for (int t = t0; t < (A16_SEQ_NOLOOP ? t0 + 1 : t1); ++t) {
            f32x16 s, dp;
            scores(t, s, dp);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = A16_EXP2(s[r]);
                dp[r] *= s[r];
            }
            bf16x8 p0, p1, d0, d1;
            chain(s, p0, p1);
            chain(dp, d0, d1);
            const unsigned char* dt = sdOt + l31 * kTrQB + 64 * t + hh * 16;
            const unsigned char* qt = sQt + l31 * kTrQB + 64 * t + hh * 16;
            const bf16x8 a0 = lds_frag(dt), a1 = lds_frag(dt + 32), b0 = lds_frag(qt), b1 = lds_frag(qt + 32);
            dv = A16_MMA(a0, p0, dv);
            dk = A16_MMA(b0, d0, dk);
            dv = A16_MMA(a1, p1, dv);
            dk = A16_MMA(b1, d1, dk);
        }
    };
    auto bias_row = [&](float* slot, const f32x16& dk, const f32x16& dv) {   // the bias key's two lanes: raw (dk | dv) into a slot
#pragma unroll
        for (int gq = 0; gq < 3; ++gq)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                slot[8 * gq + 4 * hh + i] = dk[4 * gq + i] * kLn2;   // (the Q images carry q log2(e))
                slot[kDH + 8 * gq + 4 * hh + i] = dv[4 * gq + i];
            }
    };
    const int own = 32 * w + l31;
    if (w < nkt && w < 8) {
        f32x16 dk, dv;
        key_pass(w, 0, nqt, dk, dv);
        if (own == len) bias_row(sRed, dk, dv);
        unrope_fast(dk, own < len ? own : 0, inv_freq, hh, kLn2);   // the Q images carry q log2(e)
        if (own < len && !A16_SEQ_NOSTORE) {
            const long e0 = (tok0 + (long)own * pstr) * ld + kC + g.hd * kDH + 4 * hh;
            store_grad(dqkv, e0, dk, out16);
            store_grad(dqkv, e0 + kC, dv, out16);
        }
    }
    if (nkt == 9 && w < nqt) {   // len = 256: the bias key's tile, one query tile per wave
        f32x16 dk, dv;
        key_pass(8, w, w + 1, dk, dv);
        if (l31 == 0) bias_row(sRed + w * 2 * kDH, dk, dv);
    }
    __syncthreads();
    // the bias key: partial rows summed in wave order, dk back through its rotation (position len), one row per sequence
    float* dst = dbias + (long)g.seq * 2 * kC + g.hd * kDH;
    auto red = [&](int i) {
        float a = 0.f;
#pragma unroll
        for (int x = 0; x < 8; ++x) a += sRed[x * 2 * kDH + i];
        return a;
    };
    if (tid < 12) {
        const float ang = (float)len * inv_freq[tid];
        const float c = cosf(ang), sn = sinf(ang);
        const float y1 = red(tid), y2 = red(tid + 12);
        dst[tid] = y1 * c + y2 * sn;            // R(-theta): d x1 = d y1 c + d y2 s
        dst[tid + 12] = y2 * c - y1 * sn;       //            d x2 = d y2 c - d y1 s
    } else if (tid >= 32 && tid < 32 + kDH) {
        dst[kC + tid - 32] = red(kDH + tid - 32);
    }
}

// Workgroup b: sequence (b / 32 / 8) * 8 + b % 8 (all 32 workgroups of a sequence on ONE XCD, see wg_of), pass (b / 8 / 16) % 2,
// head (b / 8) % 16: the two passes of a sequence are dispatched back to back, so the key pass finds the rows the query pass read
// in its XCD's L2 (64 resident workgroups = two sequences = 3.8 MB of rows).
__global__ __launch_bounds__(512, 2) void k16_attn_bwd_seq(const float* __restrict__ qkv, int ld, AxisMap ax, MaskMap mk,
                                                           const float* __restrict__ bias_k, const float* __restrict__ bias_v,
                                                           const float* __restrict__ inv_freq, const float* __restrict__ o,
                                                           const float* __restrict__ dout, float* __restrict__ dqkv,
                                                           float* __restrict__ dbias, const float* __restrict__ lse_in, bool rope, bool out16) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[kSeqLds];
    const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3, within = rest % (2 * kH);
    const Wg g{(rest / (2 * kH)) * 8 + xcd, within % kH, 0};
    if (g.seq >= ax.nseq) return;
    if (within < kH) attn_bwd_seq_q(lds, g, qkv, ld, ax, mk, bias_k, bias_v, inv_freq, o, dout, dqkv, lse_in, rope, out16);
    else attn_bwd_seq_kv(lds, g, qkv, ld, ax, mk, bias_k, bias_v, inv_freq, o, dout, dqkv, dbias, lse_in, rope, out16);
}

// ---- forward, one workgroup per (sequence, head): the same idea for the forward pass (k: 288 row-major rows, v transposed, the
// key-validity addends: 43 KB, filled once; no barrier and no global load inside the key loop).  Tile order and arithmetic are
// those of k16_attn, except that q is multiplied by log2(e) BEFORE it is rounded to bf16 (scores in log2 units: no multiply in
// front of the exponentials, here and in k16_attn_bwd_seq, which rounds the same product): the two forms agree to bf16 rounding.
__global__ __launch_bounds__(512, 2) void k16_attn_seq(const float* __restrict__ qkv, int ld, AxisMap ax, MaskMap mk,
                                                       const float* __restrict__ bias_k, const float* __restrict__ bias_v,
                                                       const float* __restrict__ inv_freq, float* __restrict__ out,
                                                       float* __restrict__ lse_out, bool rope) {
    __shared__ __attribute__((aligned(16))) unsigned char sK[kSeqK * kRowB];
    __shared__ __attribute__((aligned(16))) unsigned char sVt[32 * kTrKB];
    __shared__ __attribute__((aligned(16))) float sM[kSeqK];
    __shared__ __attribute__((aligned(16))) float sB[2 * kDH];
    __shared__ __attribute__((aligned(16))) float sF[12];
    const int len = ax.len, tid = threadIdx.x, lane = lane_id(), l31 = lane & 31, hh = lane >> 5, w = wave_id();
    const Wg g = wg_of(1);
    if (g.seq >= ax.nseq) return;
    const int nkt = len / 32 + 1, nqt = (len + 31) / 32;
    const long tok0 = ax.token(g.seq, 0);
    const long pstr = ax.pos_stride;
    fill_bias(sB, bias_k, bias_v, inv_freq, g.hd, len);
    for (int e = tid; e < kSeqK; e += 512) *reinterpret_cast<u32x4*>(sK + e * kRowB + 48) = u32x4{0, 0, 0, 0};
    for (int e = tid; e < 8 * (kTrKB / 16); e += 512) *reinterpret_cast<u32x4*>(sVt + 24 * kTrKB + e * 16) = u32x4{0, 0, 0, 0};
    for (int r = tid; r < 32 * nkt; r += 512)
        sM[r] = r < len ? (mk.at(tok0 + (long)r * pstr) != 0.f ? 0.f : kMasked) : (r == len ? 0.f : kMasked);
    const int qi = 32 * w + l31;
    const long qtok = tok0 + (long)(qi < len ? qi : len - 1) * pstr;
    fill_freq(sF, inv_freq);
    Own q;
    if (!rope) q = own_row(qkv + qtok * ld + g.hd * kDH, hh, nullptr, nullptr, kLog2e);   // scores in log2 units
    __syncthreads();   // sB, sF
    if (rope) q = own_row_rope(qkv + qtok * ld + g.hd * kDH, qi < len ? qi : len - 1, hh, sF, kLog2e);
    if (tid < 16 * nkt * 3) {   // the fill: (row pair, feature quads qq and qq + 3) items (see k16_attn_bwd_seq's query pass)
        const int pair = tid / 3, qq = tid - 3 * pair, r0 = 2 * pair;
        f32x4 kf[2][2], vf[2][2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int r = r0 + x;
            kf[x][0] = kf[x][1] = vf[x][0] = vf[x][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r < len) {
                const float* row = qkv + (tok0 + (long)r * pstr) * ld + g.hd * kDH + 4 * qq;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    kf[x][h] = *reinterpret_cast<const f32x4*>(row + kC + 12 * h);
                    vf[x][h] = *reinterpret_cast<const f32x4*>(row + 2 * kC + 12 * h);
                }
                if (rope) rope_pair(kf[x][0], kf[x][1], r, qq, sF);
            } else if (r == len) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    kf[x][h] = *reinterpret_cast<const f32x4*>(sB + 4 * qq + 12 * h);
                    vf[x][h] = *reinterpret_cast<const f32x4*>(sB + kDH + 4 * qq + 12 * h);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int x = 0; x < 2; ++x) put_rm(sK, r0 + x, qq + 3 * h, kf[x][h]);
            put_tr(sVt, kTrKB, r0, qq + 3 * h, vf[0][h], vf[1][h]);
        }
    }
    __syncthreads();
    if (w >= nqt) return;
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = opaque_zero();
    float mrun = kMasked, den = 0.f;
#pragma unroll 3
    for (int t = 0; t < nkt; ++t) {
        f32x16 s = rows_from(sM + 32 * t, hh, 1.0f);
        const unsigned char* kr = sK + (32 * t + l31) * kRowB + hh * 16;
        s = A16_MMA(lds_frag(kr), q.f0, s);
        s = A16_MMA(lds_frag(kr + 32), q.f1, s);
        float tmax = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
        const float mnew = fmaxf(mrun, half_max(tmax));
        const float alpha = A16_EXP2(mrun - mnew);
        mrun = mnew;
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = A16_EXP2(s[r] - mnew);
            sum += s[r];
            o[r] *= alpha;
        }
        den = den * alpha + sum;
        bf16x8 p0, p1;
        chain(s, p0, p1);
        const unsigned char* vr = sVt + l31 * kTrKB + 64 * t + hh * 16;
        o = A16_MMA(lds_frag(vr), p0, o);
        o = A16_MMA(lds_frag(vr + 32), p1, o);
    }
    den = half_sum(den);
    if (qi >= len) return;
    const float inv = 1.0f / den;
    float* dst = out + qtok * kC + g.hd * kDH + 4 * hh;
#pragma unroll
    for (int gq = 0; gq < 3; ++gq)
        *reinterpret_cast<f32x4*>(dst + 8 * gq) = f32x4{o[4 * gq] * inv, o[4 * gq + 1] * inv, o[4 * gq + 2] * inv, o[4 * gq + 3] * inv};
    if (lse_out && hh == 0) lse_out[qtok * kH + g.hd] = mrun * 0.6931471805599453f + logf(den);
}

// 1 (default): axes of 129 .. 256 positions take the sequence-resident backward kernel; 0: the chunked pair for every length
// (context option "train_attn_form"; mdgen_debug_train_attention: precision 160)
thread_local int g_k16_attn_form = 1;

static unsigned wg_grid(int nseq, int nblk) { return (unsigned)((long)((nseq + 7) / 8) * 8 * kH * nblk); }   // see wg_of

// Forward: 256 queries per workgroup (eight waves) once an axis is longer than 128 -- K / V are then staged once per
// (sequence, head) at the ATLAS lengths instead of once per 128-query block (124 -> 104 us); four waves below that.
bool attn16_seq_form(const AxisMap& ax) { return g_k16_attn_form && ax.len > 128 && ax.len <= kSeqQ; }

void launch16_attn(const float* qkv, int ld, const AxisMap& ax, const MaskMap& mk, const float* bias_k, const float* bias_v,
                   const float* inv_freq, float* out, hipStream_t s, float* lse_out, bool rope_inside) {
    if (attn16_seq_form(ax)) {
        hipLaunchKernelGGL(k16_attn_seq, dim3(wg_grid(ax.nseq, 1)), dim3(512), 0, s, qkv, ld, ax, mk, bias_k, bias_v, inv_freq, out, lse_out,
                           rope_inside);
    } else if (ax.len > 128) {
        const int nqb = (ax.len + 255) / 256;
        hipLaunchKernelGGL(k16_attn<8>, dim3(wg_grid(ax.nseq, nqb)), dim3(512), 0, s, qkv, ld, ax, mk, bias_k, bias_v,
                           inv_freq, out, lse_out);
    } else {
        hipLaunchKernelGGL(k16_attn<4>, dim3(wg_grid(ax.nseq, 1)), dim3(256), 0, s, qkv, ld, ax, mk, bias_k, bias_v, inv_freq,
                           out, lse_out);
    }
}
// (the backward passes stay at four waves: with eight -- 148 registers, one workgroup per CU, staging done by 192 of 512
// threads -- the query pass went 168 -> 186 us and the key pass 172 -> 221 us at the ATLAS lengths)
void launch16_attn_bwd(const float* qkv, int ld, const AxisMap& ax, const MaskMap& mk, const float* bias_k,
                       const float* bias_v, const float* inv_freq, const float* o, const float* dout, float* dqkv,
                       float* stats, float* dbias, hipStream_t s, const float* lse_in, bool rope_inside, bool out_bf16) {
    if (attn16_seq_form(ax)) {   // one workgroup per (sequence, head) and pass, the other side resident in LDS
        hipLaunchKernelGGL(k16_attn_bwd_seq, dim3(2 * wg_grid(ax.nseq, 1)), dim3(512), 0, s, qkv, ld, ax, mk, bias_k, bias_v, inv_freq, o, dout,
                           dqkv, dbias, lse_in, rope_inside, out_bf16);
        return;
    }
    const int nqb = (ax.len + 127) / 128, nkb = (ax.len + 128) / 128;
    hipLaunchKernelGGL(k16_attn_bwd_q<4>, dim3(wg_grid(ax.nseq, nqb)), dim3(256), 0, s, qkv, ld, ax, mk, bias_k, bias_v,
                       inv_freq, o, dout, dqkv, stats, lse_in);
    hipLaunchKernelGGL(k16_attn_bwd_kv<4>, dim3(wg_grid(ax.nseq, nkb)), dim3(256), 0, s, qkv, ld, ax, mk, bias_k,
                       bias_v, inv_freq, dout, stats, dqkv, dbias);
}

}  // namespace mdg
