// k_small.hip -- step-invariant / token-embedding / IPA kernels (HBM- or latency-bound, fp32).
#include "kernels.h"

namespace mdg {

// -------------------------------------------------------------------------------------------------
// TimestepEmbedder (layers.py:17-55) followed by the SiLU that opens every adaLN_modulation
// (latent_model.py:349-352, 408-411; layers.py:66-69):  out[r] = SiLU(W2 SiLU(W0 emb(t_r*mult) + b0) + b2)
// emb = [cos(t f_i) | sin(t f_i)], f_i = exp(-ln(1e4) i / 128).  One block per time row.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(384) void k_temb(const float* __restrict__ t_rows, float tmul,
                                              const float* __restrict__ w0, const float* __restrict__ b0,
                                              const float* __restrict__ w2, const float* __restrict__ b2,
                                              float* __restrict__ out) {
    __shared__ float emb[256];
    __shared__ float h1[kC];
    const int r = blockIdx.x, c = threadIdx.x;
    const float t = t_rows[r] * tmul;
    if (c < 256) {
        const int i = c & 127;
        const float f = expf(-9.210340371976184f * (float)i / 128.0f);
        const float a = t * f;
        emb[c] = (c < 128) ? cosf(a) : sinf(a);
    }
    __syncthreads();
    float acc = b0[c];
    const float* wr = w0 + (long)c * 256;
    for (int i = 0; i < 256; ++i) acc += wr[i] * emb[i];
    h1[c] = acc / (1.0f + expf(-acc));
    __syncthreads();
    float acc2 = b2[c];
    const float* wr2 = w2 + (long)c * kC;
    for (int i = 0; i < kC; ++i) acc2 += wr2[i] * h1[i];
    out[(long)r * kC + c] = acc2 / (1.0f + expf(-acc2));
}

// adaLN table: mod[r][o] = b[o] + sum_c W[o][c] st[r][c]  (all adaLN_modulation Linears of the model
// concatenated along o).  One wave per output feature; lanes split K = 384.
__global__ __launch_bounds__(256) void k_adaln(const float* __restrict__ st, int nrows, const float* __restrict__ w,
                                               const float* __restrict__ b, int nout, float* __restrict__ mod) {
    const int o = blockIdx.x * 4 + wave_id();
    if (o >= nout) return;
    const int lane = lane_id();
    float wv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) wv[i] = w[(long)o * kC + lane + 64 * i];
    const float bo = b[o];
    for (int r = 0; r < nrows; ++r) {
        const float* s = st + (long)r * kC;
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) a += wv[i] * s[lane + 64 * i];
        a = wave_sum(a);
        if (lane == 0) mod[(long)r * nout + o] = a + bo;
    }
}

// rotary table, 128 B per position: rope[pos][h][16] = cos(pos f_i) i=6h..6h+5 | 2 pad | sin(...) | 2 pad,
// f = rot_emb.inv_freq (mha.py:130,356).  Each lane-half's six (cos, sin) pairs are 16-byte aligned and never
// straddle a cache line (DESIGN.md "misaligned line-crossing loads").
__global__ void k_rope_table(float* __restrict__ rope, const float* __restrict__ inv_freq, int npos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npos * 16) return;
    const int pos = i / 16, k = i % 16;
    const int h = k >> 3, j = k & 7;
    float c = 0.f, sn = 0.f;
    if (j < 6) {
        const float a = (float)pos * inv_freq[6 * h + j];
        c = cosf(a);
        sn = sinf(a);
    }
    rope[(long)pos * kRopeRow + h * 16 + j] = c;
    rope[(long)pos * kRopeRow + h * 16 + 8 + j] = sn;
}

__global__ void k_gather_f32(const float* __restrict__ src, const int* __restrict__ idx, float scale,
                             float* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = idx[i] >= 0 ? scale * src[idx[i]] : 0.f;
}

// Weight re-pack into MFMA fragment order: dst[(ft*ksteps + ks)*64 + lane] = 8 bf16 of
// W[rowmap[ft*32 + (lane&31)]][ks*16 + (lane>>5)*8 .. +7] * scale   (rowmap < 0 -> zeros)
// kappa == 1: the K slice of a k-step in rows.h's kappa order, k = 16 ks + 8 (j >> 2) + 4 hh + (j & 3) (operands whose B fragments are
// LayerNorm registers of the row-owner kernels); 0: natural, k = 16 ks + 8 hh + j
__global__ void k_pack_rows(const float* __restrict__ w, int ld, const int* __restrict__ rowmap, int nft, int ksteps,
                            float scale, bf16x8* __restrict__ dst, int kappa, int part) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)nft * ksteps * 64;
    if (i >= total) return;
    const int lane = (int)(i & 63);
    const long fk = i >> 6;
    const int ks = (int)(fk % ksteps), ft = (int)(fk / ksteps);
    const int row = rowmap[ft * 32 + (lane & 31)];
    const int hh = lane >> 5;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = kappa ? ks * 16 + 8 * (j >> 2) + 4 * hh + (j & 3) : ks * 16 + hh * 8 + j;
        const float x = row >= 0 && k < ld ? w[(long)row * ld + k] * scale : 0.f;
        const __bf16 hi = (__bf16)x;
        v[j] = part ? (__bf16)(x - (float)hi) : hi;
    }
    dst[i] = v;
}

// -------------------------------------------------------------------------------------------------
// Token embedding (latent_model.py:233-246):
//   h = latent_to_emb(x) [+ pos_embed[l]] + cond_to_emb(x_cond) + mask_to_emb[x_cond_mask] + ipa_out[b,l]
// A [N x D] . [D x 384] product with D = 21 / 28: far too thin for the bf16 path's panels, and as a per-channel dot
// product on the VALU (round 1) it ran at 0.9 TB/s of its 98 MB output.  Here it is an fp32 MFMA job:
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate -- the reference's arithmetic), D = tokens x channels.
// Workgroup = 128 tokens (4 row tiles) x 384 channels; wave w owns channels 96 w .. 96 w + 95 and keeps its slices
// of both weight matrices in registers (2 x 3 tiles x 14 k-steps); token rows are staged once in LDS (row stride 29
// floats: the per-k column reads are conflict-free).  cond_to_emb is skipped for row tiles whose x_cond rows are all
// zero (every non-conditioning frame).  The epilogue adds bias, mask embedding, pos_embed / ipa_out rows (requested
// four token rows ahead) and stores 128-byte row segments.
// -------------------------------------------------------------------------------------------------
constexpr int kEmbLd = 29, kEmbK = 14;
// POS / IPA: pos_embed / ipa_out present (compile-time so that their loads are unconditional).
// kEmbTok: tokens per workgroup -- 128 when that still makes a few hundred workgroups; 32 (one row tile per workgroup) for the
// small launches (B = 1, the TPS shard), whose time is one workgroup's latency: 40 -> ~17 us at 12 800 tokens.
template <bool POS, bool IPA, int kEmbTok>
__global__ __launch_bounds__(256) void k_embed(const EmbedParams p) {
    __shared__ float xs[kEmbTok * kEmbLd];
    __shared__ float cs[kEmbTok * kEmbLd];
    __shared__ int ms[kEmbTok];     // bit0: x_cond_mask, bit1: x_cond row has a non-zero
    __shared__ int tpos[kEmbTok];   // l * kC                 (row of pos_embed)
    __shared__ int tipa[kEmbTok];   // (b * L + l) * kC       (row of ipa_out)
    const int tid = threadIdx.x;
    const long tok0 = (long)blockIdx.x * kEmbTok;
    if (tid < kEmbTok) {
        long t = tok0 + tid;
        t = t < p.N ? t : p.N - 1;
        const int l = (int)(t % p.L), b = (int)(t / ((long)p.T * p.L));
        ms[tid] = 0;
        tpos[tid] = l * kC;
        tipa[tid] = (b * p.L + l) * kC;
    }
    __syncthreads();
    {   // stage x / x_cond rows: unconditional clamped loads, all in flight together
        constexpr int NIT = (kEmbTok * 28 + 255) / 256;
        float a[NIT], b[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i0 = tid + 256 * it;
            const int i = i0 < kEmbTok * 28 ? i0 : kEmbTok * 28 - 1;   // (32-token workgroups: 3.5 rounds)
            const int tk = i / 28, d = i % 28;
            long t = tok0 + tk;
            t = t < p.N ? t : p.N - 1;
            const int dc = d < p.D ? d : 0;
            a[it] = p.x[t * p.D + dc];
            b[it] = p.x_cond[t * p.D + dc];
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + 256 * it;
            if (i >= kEmbTok * 28) break;
            const int tk = i / 28, d = i % 28;
            const bool ok = tok0 + tk < p.N && d < p.D;
            xs[tk * kEmbLd + d] = ok ? a[it] : 0.f;
            cs[tk * kEmbLd + d] = ok ? b[it] : 0.f;
            if (ok && b[it] != 0.f) atomicOr(&ms[tk], 2);
        }
    }
    const int w = __builtin_amdgcn_readfirstlane(wave_id()), lane = lane_id(), hh = lane >> 5, j = lane & 31;
    const int nk = (p.D + 1) >> 1;
    // B operands (k = 2 ks + hh, column = channel): this wave's three channel tiles of both weight matrices
    // (from the packed copies, k_pack_embed: one 256-byte request per (tile, k-step) instead of 64 lines 84 bytes apart --
    // 168 such gathers per lane were the fixed cost of every workgroup)
    float wl[3][kEmbK], wc[3][kEmbK], b0[3], me0[3], me1[3];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
        const int c = 96 * w + 32 * ct + j;
#pragma unroll
        for (int ks = 0; ks < kEmbK; ++ks) {
            const int o = ((w * 3 + ct) * kEmbK + ks) * 64 + lane;
            wl[ct][ks] = p.wl_pack[o];
            wc[ct][ks] = p.wc_pack[o];
        }
        b0[ct] = p.bl[c] + p.bc[c];
        me0[ct] = p.mask_emb[c];
        me1[ct] = p.mask_emb[kC + c];
    }
    __syncthreads();
    if (tid < kEmbTok) {
        const long t = tok0 + tid;
        if (t < p.N && p.x_cond_mask[t]) ms[tid] |= 1;
    }
    __syncthreads();
#pragma unroll 1
    for (int rt = 0; rt < kEmbTok / 32; ++rt) {
        if (tok0 + rt * 32 >= p.N) break;
        f32x16 acc[3];
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
        const float* xr = xs + (rt * 32 + j) * kEmbLd + hh;
#pragma unroll
        for (int ks = 0; ks < kEmbK; ++ks) {
            if (ks < nk) {
                const float a = xr[2 * ks];
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wl[ct][ks], acc[ct], 0, 0, 0);
            }
        }
        if (__builtin_amdgcn_ballot_w64((ms[rt * 32 + j] & 2) != 0) != 0) {   // some token of this tile is conditioned
            const float* cr = cs + (rt * 32 + j) * kEmbLd + hh;
#pragma unroll
            for (int ks = 0; ks < kEmbK; ++ks) {
                if (ks < nk) {
                    const float a = cr[2 * ks];
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wc[ct][ks], acc[ct], 0, 0, 0);
                }
            }
        }
        // epilogue: register r of lane-half hh is token row mfma_row(r, hh) of the tile, lane j its channel
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 4) {
            float add[4][3];
            int mrow[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int tk = rt * 32 + mfma_row(r0 + u, hh);
                mrow[u] = ms[tk];
                const int op = tpos[tk], oi = tipa[tk];
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) {
                    const int c = 96 * w + 32 * ct + j;
                    float v = 0.f;
                    if (POS) v += p.pos_embed[op + c];
                    if (IPA) v += p.ipa_out[oi + c];
                    add[u][ct] = v;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long t = tok0 + rt * 32 + mfma_row(r0 + u, hh);
                if (t < p.N) {
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct)
                        p.h[t * kC + 96 * w + 32 * ct + j] = acc[ct][r0 + u] + b0[ct] + ((mrow[u] & 1) ? me1[ct] : me0[ct]) + add[u][ct];
                }
            }
        }
    }
}

// IPA-stack input (latent_model.py:184-187 / 193-201): h[g,l] = aatype_emb[aatype[b,l]]
// (+ latent_to_emb_{f,r}(rel7[b,l]) for the two-sided model); g = step*B + b.
__global__ __launch_bounds__(384) void k_ipa_init(const float* __restrict__ aa_emb, const int64_t* __restrict__ aatype,
                                                  const float* __restrict__ rel7, const float* __restrict__ w7,
                                                  const float* __restrict__ b7, float* __restrict__ h, int B, int L) {
    const long row = blockIdx.x;  // (g, l)
    const int c = threadIdx.x;
    const int l = (int)(row % L);
    const int b = (int)((row / L) % B);
    const long bl = (long)b * L + l;
    float a = aa_emb[(long)aatype[bl] * kC + c];
    if (rel7) {
        a += b7[c];
#pragma unroll
        for (int d = 0; d < 7; ++d) a += w7[c * 7 + d] * rel7[bl * 7 + d];
    }
    h[row * kC + c] = a;
}

// Host float values travel as kernel arguments (captured by value in a hipGraph).  Replaces a train of
// 4-byte hipMemsetD32Async calls, which intermittently lost writes on ROCm 7 (DESIGN.md "hardware findings").
__global__ void k_write_floats(FloatChunk c, float* __restrict__ dst) {
    const int i = threadIdx.x;
    if (i < c.n) dst[i] = c.v[i];
}

__global__ void k_add_inplace(float* __restrict__ dst, const float* __restrict__ src, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

// -------------------------------------------------------------------------------------------------
// Invariant point attention core, c_z = 0, H=4, c=32, Pq=Pv=8 (ipa.py:126-234), fp32.
// One thread per (group, query residue, head); keys streamed with an online softmax.
//   logit = q.k sqrt(1/96) - 0.5 softplus(w_h) sqrt(1/108) sum_p |R_i qp + t_i - R_j kp - t_j|^2
//           + 1e5 (m_i m_j - 1)
// out[token] = [ o(128) | o_pt.x(32) | o_pt.y(32) | o_pt.z(32) | |o_pt|(32) ] as bf16 (ipa.py:250-254)
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rot_apply(const float* R, const float* t, float x, float y, float z, float& ox,
                                          float& oy, float& oz) {
    ox = R[0] * x + R[1] * y + R[2] * z + t[0];
    oy = R[3] * x + R[4] * y + R[5] * z + t[1];
    oz = R[6] * x + R[7] * y + R[8] * z + t[2];
}

// concat features of one (token, head): [ o(32) | o_pt.x(8) | o_pt.y(8) | o_pt.z(8) | |o_pt|(8) ] at their places in the
// 256-wide row (ipa.py:226-254), bf16 for the MFMA path or fp32 for the fp32 path
__device__ __forceinline__ void ipa_store_features(const IpaAttnParams& p, long gi, int hd, const float (&o)[32],
                                                   const float (&op)[8][3], float inv, const float (&Ri)[9],
                                                   const float (&ti)[3]) {
    auto put = [&](int col, float v) {
        if (p.feat32) p.feat32[gi * kIpaFeat + col] = v;
        else p.feat[gi * kIpaFeat + col] = (__bf16)v;
    };
#pragma unroll
    for (int c = 0; c < 32; ++c) put(hd * 32 + c, o[c] * inv);
#pragma unroll
    for (int pt = 0; pt < 8; ++pt) {
        const float gx = op[pt][0] * inv - ti[0], gy = op[pt][1] * inv - ti[1], gz = op[pt][2] * inv - ti[2];
        const float lx = Ri[0] * gx + Ri[3] * gy + Ri[6] * gz;   // R^T (p - t)  (rigid_utils.py:1061-1073)
        const float ly = Ri[1] * gx + Ri[4] * gy + Ri[7] * gz;
        const float lz = Ri[2] * gx + Ri[5] * gy + Ri[8] * gz;
        put(128 + hd * 8 + pt, lx);
        put(160 + hd * 8 + pt, ly);
        put(192 + hd * 8 + pt, lz);
        put(224 + hd * 8 + pt, sqrtf(lx * lx + ly * ly + lz * lz + 1e-8f));
    }
}

__global__ __launch_bounds__(256) void k_ipa_attn(const IpaAttnParams p) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)p.ngroups * p.L * 4;
    if (idx >= total) return;
    const int hd = (int)(idx & 3);
    const long gi = idx >> 2;
    const int i = (int)(gi % p.L);
    const long g = gi / p.L;
    const int b = (int)(g % p.B);
    const float* pi = p.proj + gi * kIpaProj;
    float Ri[9], ti[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) Ri[k] = p.rot[((long)b * p.L + i) * 9 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) ti[k] = p.trans[((long)b * p.L + i) * 3 + k];
    const float mi = p.mask_bl[(long)b * p.L + i];
    float q[32], qp[8][3];
#pragma unroll
    for (int c = 0; c < 32; ++c) q[c] = pi[hd * 32 + c];
#pragma unroll
    for (int pt = 0; pt < 8; ++pt)
        rot_apply(Ri, ti, pi[384 + hd * 8 + pt], pi[416 + hd * 8 + pt], pi[448 + hd * 8 + pt], qp[pt][0], qp[pt][1],
                  qp[pt][2]);
    const float hwraw = p.head_w[hd];
    const float sp = (hwraw > 20.f) ? hwraw : log1pf(expf(hwraw));  // torch softplus (threshold 20)
    const float hw = sp * 0.09622504486493763f;                     // sqrt(1/(3*(8*9/2))) = sqrt(1/108)
    const float qk_scale = 0.10206207261596575f;                    // sqrt(1/(3*32))
    float o[32], op[8][3];
#pragma unroll
    for (int c = 0; c < 32; ++c) o[c] = 0.f;
#pragma unroll
    for (int pt = 0; pt < 8; ++pt) op[pt][0] = op[pt][1] = op[pt][2] = 0.f;
    float mrun = -3.0e38f, den = 0.f;
#pragma unroll 2   // two keys' loads in flight: the loop is a chain of L2 round trips (L = 256: 796 -> 657 us per launch)
    for (int j = 0; j < p.L; ++j) {
        const float* pj = p.proj + (g * p.L + j) * kIpaProj;
        float Rj[9], tj[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) Rj[k] = p.rot[((long)b * p.L + j) * 9 + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) tj[k] = p.trans[((long)b * p.L + j) * 3 + k];
        const float mj = p.mask_bl[(long)b * p.L + j];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) dot += q[c] * pj[128 + hd * 64 + c];
        float d2 = 0.f;
#pragma unroll
        for (int pt = 0; pt < 8; ++pt) {
            float kx, ky, kz;
            rot_apply(Rj, tj, pj[480 + hd * 16 + pt], pj[544 + hd * 16 + pt], pj[608 + hd * 16 + pt], kx, ky, kz);
            const float dx = qp[pt][0] - kx, dy = qp[pt][1] - ky, dz = qp[pt][2] - kz;
            d2 += dx * dx + dy * dy + dz * dz;
        }
        const float logit = dot * qk_scale - 0.5f * hw * d2 + 1e5f * (mi * mj - 1.0f);
        const float mnew = fmaxf(mrun, logit);
        const float alpha = expf(mrun - mnew);
        const float pw = expf(logit - mnew);
        mrun = mnew;
        den = den * alpha + pw;
#pragma unroll
        for (int c = 0; c < 32; ++c) o[c] = o[c] * alpha + pw * pj[128 + hd * 64 + 32 + c];
#pragma unroll
        for (int pt = 0; pt < 8; ++pt) {
            float vx, vy, vz;
            rot_apply(Rj, tj, pj[480 + hd * 16 + 8 + pt], pj[544 + hd * 16 + 8 + pt], pj[608 + hd * 16 + 8 + pt], vx, vy,
                      vz);
            op[pt][0] = op[pt][0] * alpha + pw * vx;
            op[pt][1] = op[pt][1] * alpha + pw * vy;
            op[pt][2] = op[pt][2] * alpha + pw * vz;
        }
    }
    const float inv = 1.0f / den;
    if (p.stats) p.stats[gi * 4 + hd] = mrun + logf(den);
    ipa_store_features(p, gi, hd, o, op, inv, Ri, ti);
}

// -------------------------------------------------------------------------------------------------
// The same attention for long sequences (ATLAS: L = 256, 4 x 256 x 256 logits per group): one workgroup per
// (group, head, 256-query tile), one thread per query, keys staged through LDS 32 at a time.
// What the per-thread kernel above pays per (query, key) PAIR is paid here once per KEY by the staging step: the
// global loads (every lane of a wave wanted the same key row: a chain of L dependent L2 round trips per thread) and
// the rotation of the 16 key / value points into the global frame (ipa.py:143-147: R_j p + t_j).  The inner loop
// then reads the tile from LDS with wave-uniform addresses (broadcast, conflict-free) and is pure fp32 FMA work;
// the softmax is renormalised once per 32-key tile (one max, one exp per key) instead of twice per key.
// 4.1 ms -> ~0.1 ms per launch at cfg-4 (B 1, L 256, 49 prepared steps).
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ipa_attn_tiled(const IpaAttnParams p) {
    __shared__ __attribute__((aligned(16))) float sk[kIpaKT][32];    // k vectors of this head
    __shared__ __attribute__((aligned(16))) float sv[kIpaKT][32];    // v vectors
    __shared__ __attribute__((aligned(16))) float skp[kIpaKT][24];   // key points, global frame [pt][xyz]
    __shared__ __attribute__((aligned(16))) float svp[kIpaKT][24];   // value points, global frame
    __shared__ float sm[kIpaKT];                                     // key mask (0 beyond L: see below)
    const int nqt = (p.L + 255) / 256;
    const int qt = blockIdx.x % nqt;
    const int hd = (blockIdx.x / nqt) & 3;
    const long g = blockIdx.x / (nqt * 4);
    const int b = (int)(g % p.B);
    const int tid = threadIdx.x;
    const int i = qt * 256 + tid;
    const bool qok = i < p.L;
    const int ic = qok ? i : p.L - 1;     // idle threads shadow the last query (loads stay in range), never store
    const long gi = g * p.L + ic;
    const float* pi = p.proj + gi * kIpaProj;
    float Ri[9], ti[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) Ri[k] = p.rot[((long)b * p.L + ic) * 9 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) ti[k] = p.trans[((long)b * p.L + ic) * 3 + k];
    const float mi = p.mask_bl[(long)b * p.L + ic];
    const float qk_scale = 0.10206207261596575f;                    // sqrt(1/(3*32))
    float q[32], qp[8][3];
#pragma unroll
    for (int c = 0; c < 32; c += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(pi + hd * 32 + c);
        q[c] = v[0] * qk_scale; q[c + 1] = v[1] * qk_scale; q[c + 2] = v[2] * qk_scale; q[c + 3] = v[3] * qk_scale;
    }
#pragma unroll
    for (int pt = 0; pt < 8; ++pt)
        rot_apply(Ri, ti, pi[384 + hd * 8 + pt], pi[416 + hd * 8 + pt], pi[448 + hd * 8 + pt], qp[pt][0], qp[pt][1],
                  qp[pt][2]);
    const float hwraw = p.head_w[hd];
    const float sp = (hwraw > 20.f) ? hwraw : log1pf(expf(hwraw));  // torch softplus (threshold 20)
    const float hwh = -0.5f * sp * 0.09622504486493763f;            // -1/2 softplus(w_h) sqrt(1/108)
    float o[32], op[8][3];
#pragma unroll
    for (int c = 0; c < 32; ++c) o[c] = 0.f;
#pragma unroll
    for (int pt = 0; pt < 8; ++pt) op[pt][0] = op[pt][1] = op[pt][2] = 0.f;
    float mrun = -3.0e38f, den = 0.f;
    // staging roles: thread -> (key of the tile, 8-float slice / point index)
    const int skey = tid >> 3, ssub = tid & 7;
    const int nsl = p.nsplit > 1 ? p.nsplit : 1, sl = blockIdx.y;
    const int per = ((p.L + kIpaKT - 1) / kIpaKT + nsl - 1) / nsl * kIpaKT;      // whole tiles per slice
    const int jlo = sl * per, jhi = jlo + per < p.L ? jlo + per : p.L;
    for (int j0 = jlo; j0 < jhi; j0 += kIpaKT) {
        __syncthreads();   // the previous tile has been consumed
        {
            const int j = j0 + skey;
            const int jc = j < p.L ? j : p.L - 1;
            const float* pj = p.proj + (g * p.L + jc) * kIpaProj;
            const f32x4 kk = *reinterpret_cast<const f32x4*>(pj + 128 + hd * 64 + 4 * ssub);
            const f32x4 vv = *reinterpret_cast<const f32x4*>(pj + 128 + hd * 64 + 32 + 4 * ssub);
            float Rj[9], tj[3];
#pragma unroll
            for (int k = 0; k < 9; ++k) Rj[k] = p.rot[((long)b * p.L + jc) * 9 + k];
#pragma unroll
            for (int k = 0; k < 3; ++k) tj[k] = p.trans[((long)b * p.L + jc) * 3 + k];
            float kx, ky, kz, vx, vy, vz;
            rot_apply(Rj, tj, pj[480 + hd * 16 + ssub], pj[544 + hd * 16 + ssub], pj[608 + hd * 16 + ssub], kx, ky, kz);
            rot_apply(Rj, tj, pj[480 + hd * 16 + 8 + ssub], pj[544 + hd * 16 + 8 + ssub], pj[608 + hd * 16 + 8 + ssub], vx,
                      vy, vz);
            *reinterpret_cast<f32x4*>(&sk[skey][4 * ssub]) = kk;
            *reinterpret_cast<f32x4*>(&sv[skey][4 * ssub]) = vv;
            skp[skey][3 * ssub] = kx; skp[skey][3 * ssub + 1] = ky; skp[skey][3 * ssub + 2] = kz;
            svp[skey][3 * ssub] = vx; svp[skey][3 * ssub + 1] = vy; svp[skey][3 * ssub + 2] = vz;
            // keys beyond L do not exist: mark them with a NaN-free sentinel that the consumer turns into weight 0
            if (ssub == 0) sm[skey] = j < p.L ? p.mask_bl[(long)b * p.L + jc] : -1.0f;
        }
        __syncthreads();
        float lg[kIpaKT];
        float tmax = -3.0e38f;
#pragma unroll
        for (int jj = 0; jj < kIpaKT; ++jj) {
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < 32; c += 4) {
                const f32x4 kv = *reinterpret_cast<const f32x4*>(&sk[jj][c]);
                dot += q[c] * kv[0] + q[c + 1] * kv[1] + q[c + 2] * kv[2] + q[c + 3] * kv[3];
            }
            float d2 = 0.f;
#pragma unroll
            for (int pt = 0; pt < 8; ++pt) {
                const float dx = qp[pt][0] - skp[jj][3 * pt], dy = qp[pt][1] - skp[jj][3 * pt + 1],
                            dz = qp[pt][2] - skp[jj][3 * pt + 2];
                d2 += dx * dx + dy * dy + dz * dz;
            }
            const float mj = sm[jj];
            const float l = dot + hwh * d2 + 1e5f * (mi * mj - 1.0f);   // ipa.py:161-203
            lg[jj] = mj < 0.f ? -3.0e38f : l;
            tmax = fmaxf(tmax, lg[jj]);
        }
        const float mnew = fmaxf(mrun, tmax);
        const float alpha = expf(mrun - mnew);
        mrun = mnew;
        den *= alpha;
#pragma unroll
        for (int c = 0; c < 32; ++c) o[c] *= alpha;
#pragma unroll
        for (int pt = 0; pt < 8; ++pt) {
            op[pt][0] *= alpha; op[pt][1] *= alpha; op[pt][2] *= alpha;
        }
#pragma unroll
        for (int jj = 0; jj < kIpaKT; ++jj) {
            const float pw = lg[jj] > -1.0e38f ? expf(lg[jj] - mnew) : 0.f;
            den += pw;
#pragma unroll
            for (int c = 0; c < 32; c += 4) {
                const f32x4 vv = *reinterpret_cast<const f32x4*>(&sv[jj][c]);
                o[c] += pw * vv[0]; o[c + 1] += pw * vv[1]; o[c + 2] += pw * vv[2]; o[c + 3] += pw * vv[3];
            }
#pragma unroll
            for (int pt = 0; pt < 8; ++pt) {
                op[pt][0] += pw * svp[jj][3 * pt];
                op[pt][1] += pw * svp[jj][3 * pt + 1];
                op[pt][2] += pw * svp[jj][3 * pt + 2];
            }
        }
    }
    if (!qok) return;
    if (nsl > 1) {   // this slice's softmax state; k_ipa_attn_merge finishes the row
        float* rec = p.part + (((long)sl * p.ngroups * p.L + gi) * 4 + hd) * kIpaFwdRec;
        rec[0] = mrun;
        rec[1] = den;
#pragma unroll
        for (int c = 0; c < 32; ++c) rec[2 + c] = o[c];
#pragma unroll
        for (int pt = 0; pt < 8; ++pt)
#pragma unroll
            for (int x = 0; x < 3; ++x) rec[34 + 3 * pt + x] = op[pt][x];
        return;
    }
    const float inv = 1.0f / den;
    if (p.stats) p.stats[gi * 4 + hd] = mrun + logf(den);
    ipa_store_features(p, gi, hd, o, op, inv, Ri, ti);
}

// One thread per (token, head): the slices' states merged in slice order (m = max m_z; everything scaled by exp(m_z - m)),
// then exactly the epilogue of the one-slice kernel.
__global__ __launch_bounds__(256) void k_ipa_attn_merge(const IpaAttnParams p) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long mtot = (long)p.ngroups * p.L;
    if (idx >= mtot * 4) return;
    const int hd = (int)(idx & 3);
    const long gi = idx >> 2;
    const int i = (int)(gi % p.L);
    const int b = (int)((gi / p.L) % p.B);
    float Ri[9], ti[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) Ri[k] = p.rot[((long)b * p.L + i) * 9 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) ti[k] = p.trans[((long)b * p.L + i) * 3 + k];
    float m = -3.0e38f;
    for (int z = 0; z < p.nsplit; ++z) m = fmaxf(m, p.part[(((long)z * mtot + gi) * 4 + hd) * kIpaFwdRec]);
    float o[32], op[8][3], den = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) o[c] = 0.f;
#pragma unroll
    for (int pt = 0; pt < 8; ++pt) op[pt][0] = op[pt][1] = op[pt][2] = 0.f;
    for (int z = 0; z < p.nsplit; ++z) {
        const float* rec = p.part + (((long)z * mtot + gi) * 4 + hd) * kIpaFwdRec;
        const float a = expf(rec[0] - m);     // an empty slice has m_z = -3e38 and den = o = 0
        den += a * rec[1];
#pragma unroll
        for (int c = 0; c < 32; ++c) o[c] += a * rec[2 + c];
#pragma unroll
        for (int pt = 0; pt < 8; ++pt)
#pragma unroll
            for (int x = 0; x < 3; ++x) op[pt][x] += a * rec[34 + 3 * pt + x];
    }
    const float inv = 1.0f / den;
    if (p.stats) p.stats[gi * 4 + hd] = m + logf(den);
    ipa_store_features(p, gi, hd, o, op, inv, Ri, ti);
}

// ---- launchers ----------------------------------------------------------------------------------
// =================================================================================================
// Flow-matching training target, forward only.
//   plan   (path.py:113-135 with GVPCPlan :177-187 / ICPlan linear):  xt = a x1 + s x0,  ut = a' x1 + s' x0
//          GVP: a = sin(pi t / 2), s = cos(pi t / 2);  linear: a = t, s = 1 - t.   t is per sample.
//   loss   (transport.py:13-17 mean_flat, :184):  loss[b] = sum((pred - ut)^2 * mask) / sum(mask)
// =================================================================================================
__global__ void k_path_plan(const float* __restrict__ t, const float* __restrict__ x0, const float* __restrict__ x1,
                            float* __restrict__ xt, float* __restrict__ ut, long per_sample, long total, int gvp) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float tb = t[i / per_sample];
    float a, sg, da, ds;
    if (gvp) {
        const float hp = 1.57079632679489662f;   // pi / 2
        const float sn = sinf(tb * hp), cs = cosf(tb * hp);
        a = sn; sg = cs; da = hp * cs; ds = -hp * sn;
    } else {
        a = tb; sg = 1.0f - tb; da = 1.0f; ds = -1.0f;
    }
    const float v1 = x1[i], v0 = x0[i];
    xt[i] = a * v1 + sg * v0;
    ut[i] = da * v1 + ds * v0;
}

// one workgroup per sample; fp32 partial sums per lane, wave reduce, 4-wave combine through LDS
__global__ __launch_bounds__(256) void k_masked_mse(const float* __restrict__ pred, const float* __restrict__ target,
                                                    const float* __restrict__ mask, float* __restrict__ loss,
                                                    long per_sample) {
    __shared__ float red[2][4];
    const long base = (long)blockIdx.x * per_sample;
    float num = 0.f, den = 0.f;
    long i = threadIdx.x;
    for (; i + 7 * 256 < per_sample; i += 8 * 256) {   // eight independent load triples in flight (B = 1: one workgroup
        float pv[8], tv[8], mv[8];                      // walks the whole sample; one load per trip cost 2 ms at cfg-5 size)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            pv[u] = pred[base + i + 256 * u];
            tv[u] = target[base + i + 256 * u];
            mv[u] = mask[base + i + 256 * u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float d = pv[u] - tv[u];
            num += d * d * mv[u];
            den += mv[u];
        }
    }
    for (; i < per_sample; i += 256) {
        const float d = pred[base + i] - target[base + i], m = mask[base + i];
        num += d * d * m;
        den += m;
    }
    num = wave_sum(num);
    den = wave_sum(den);
    if (lane_id() == 0) {
        red[0][wave_id()] = num;
        red[1][wave_id()] = den;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        loss[blockIdx.x] = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (red[1][0] + red[1][1] + red[1][2] + red[1][3]);
}

void launch_temb(const float* t_rows, int nrows, float tmul, const float* w0, const float* b0, const float* w2,
                 const float* b2, float* silu_out, hipStream_t s) {
    hipLaunchKernelGGL(k_temb, dim3(nrows), dim3(384), 0, s, t_rows, tmul, w0, b0, w2, b2, silu_out);
}
void launch_adaln(const float* st, int nrows, const float* w, const float* b, int nout, float* mod, hipStream_t s) {
    hipLaunchKernelGGL(k_adaln, dim3((nout + 3) / 4), dim3(256), 0, s, st, nrows, w, b, nout, mod);
}
void launch_rope_table(float* rope, const float* inv_freq, int npos, hipStream_t s) {
    hipLaunchKernelGGL(k_rope_table, dim3((npos * 16 + 255) / 256), dim3(256), 0, s, rope, inv_freq, npos);
}
void launch_gather_f32(const float* src, const int* idx, float scale, float* dst, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_f32, dim3((n + 255) / 256), dim3(256), 0, s, src, idx, scale, dst, n);
}
void launch_pack_rows(const float* w, int ld, const int* rowmap, int nft, int ksteps, float scale, bf16x8* dst,
                      hipStream_t s, int kappa, int part) {
    const long total = (long)nft * ksteps * 64;
    hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, ld, rowmap, nft, ksteps,
                       scale, dst, kappa, part);
}
void launch_path_plan(const float* t, const float* x0, const float* x1, float* xt, float* ut, long per_sample, long B,
                      int gvp, hipStream_t s) {
    const long total = per_sample * B;
    hipLaunchKernelGGL(k_path_plan, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, t, x0, x1, xt, ut, per_sample,
                       total, gvp);
}
// The same sums in two stages for big samples (a training step at B = 1 is ONE sample of 1.3 M elements: one workgroup
// walking it took 0.4 ms): chunks of 8192 elements -> (num, den) partials, added per sample in a fixed order.
// pred == nullptr: only den (the mask sum).
constexpr int kMseChunk = 8192;
__global__ __launch_bounds__(256) void k_masked_mse_partial(const float* __restrict__ pred, const float* __restrict__ target,
                                                            const float* __restrict__ mask, long per_sample, int nchunk,
                                                            float* __restrict__ part) {
    __shared__ float red[2][4];
    const long b = blockIdx.y;
    const long lo = (long)blockIdx.x * kMseChunk, hi = lo + kMseChunk < per_sample ? lo + kMseChunk : per_sample;
    const long base = b * per_sample;
    float num = 0.f, den = 0.f;
#pragma unroll 8
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        const float m = mask[base + i];
        den += m;
        if (pred) {
            const float d = pred[base + i] - target[base + i];
            num += d * d * m;
        }
    }
    num = wave_sum(num);
    den = wave_sum(den);
    if (lane_id() == 0) {
        red[0][wave_id()] = num;
        red[1][wave_id()] = den;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = part + (b * nchunk + blockIdx.x) * 2;
        o[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        o[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}
__global__ __launch_bounds__(64) void k_masked_mse_final(const float* __restrict__ part, int nchunk, float* __restrict__ loss,
                                                         float* __restrict__ den_out) {
    const float* p = part + (long)blockIdx.x * nchunk * 2;
    float num = 0.f, den = 0.f;
    for (int i = threadIdx.x; i < nchunk; i += 64) {
        num += p[2 * i];
        den += p[2 * i + 1];
    }
    num = wave_sum(num);
    den = wave_sum(den);
    if (threadIdx.x == 0) {
        if (loss) loss[blockIdx.x] = num / den;
        if (den_out) den_out[blockIdx.x] = den;
    }
}
// loss[b] = sum((pred - target)^2 mask) / sum(mask) (loss may be null), den_out[b] = sum(mask) (may be null).  scratch: >= 2 * B *
// ceil(per_sample / 8192) floats enables the two-stage form; without it one workgroup per sample (loss only).
void launch_masked_mse(const float* pred, const float* target, const float* mask, float* loss, long per_sample, long B,
                       hipStream_t s, float* scratch, size_t scratch_floats, float* den_out) {
    const long nchunk = (per_sample + kMseChunk - 1) / kMseChunk;
    if (scratch && nchunk > 1 && (size_t)(2 * B * nchunk) <= scratch_floats) {
        hipLaunchKernelGGL(k_masked_mse_partial, dim3((unsigned)nchunk, (unsigned)B), dim3(256), 0, s, pred, target, mask, per_sample,
                           (int)nchunk, scratch);
        hipLaunchKernelGGL(k_masked_mse_final, dim3((unsigned)B), dim3(64), 0, s, scratch, (int)nchunk, loss, den_out);
        return;
    }
    hipLaunchKernelGGL(k_masked_mse, dim3((unsigned)B), dim3(256), 0, s, pred, target, mask, loss, per_sample);
}
// pack[((w * 3 + ct) * 14 + ks) * 64 + lane] = W[96 w + 32 ct + (lane & 31)][2 ks + (lane >> 5)] (0 past D): k_embed's B operands
__global__ void k_pack_embed(const float* __restrict__ w, int D, float* __restrict__ pack) {
    const int lane = threadIdx.x, blk = blockIdx.x;   // blk = (w * 3 + ct) * 14 + ks
    const int ks = blk % kEmbK, tile = blk / kEmbK;
    const int c = 32 * tile + (lane & 31), d = 2 * ks + (lane >> 5);
    pack[blk * 64 + lane] = d < D ? w[(long)c * D + d] : 0.f;
}
void launch_pack_embed(const float* w, int D, float* pack, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_embed, dim3(4 * 3 * kEmbK), dim3(64), 0, s, w, D, pack);
}
void launch_embed(const EmbedParams& p, hipStream_t s) {
    const dim3 b(256);
    if ((p.N + 127) / 128 >= 384) {
        const dim3 g((unsigned)((p.N + 127) / 128));
        if (p.pos_embed && p.ipa_out) hipLaunchKernelGGL((k_embed<true, true, 128>), g, b, 0, s, p);
        else if (p.pos_embed) hipLaunchKernelGGL((k_embed<true, false, 128>), g, b, 0, s, p);
        else if (p.ipa_out) hipLaunchKernelGGL((k_embed<false, true, 128>), g, b, 0, s, p);
        else hipLaunchKernelGGL((k_embed<false, false, 128>), g, b, 0, s, p);
    } else {
        const dim3 g((unsigned)((p.N + 31) / 32));
        if (p.pos_embed && p.ipa_out) hipLaunchKernelGGL((k_embed<true, true, 32>), g, b, 0, s, p);
        else if (p.pos_embed) hipLaunchKernelGGL((k_embed<true, false, 32>), g, b, 0, s, p);
        else if (p.ipa_out) hipLaunchKernelGGL((k_embed<false, true, 32>), g, b, 0, s, p);
        else hipLaunchKernelGGL((k_embed<false, false, 32>), g, b, 0, s, p);
    }
}
void launch_ipa_init(const float* aa_emb, const int64_t* aatype, const float* rel7, const float* w7, const float* b7,
                     float* h, int ngroups, int B, int L, hipStream_t s) {
    hipLaunchKernelGGL(k_ipa_init, dim3((unsigned)((long)ngroups * L)), dim3(384), 0, s, aa_emb, aatype, rel7, w7, b7, h,
                       B, L);
}
void launch_write_floats(const float* host_vals, int n, float* dst, hipStream_t s) {
    for (int o = 0; o < n; o += 128) {
        FloatChunk c;
        c.n = n - o < 128 ? n - o : 128;
        for (int i = 0; i < c.n; ++i) c.v[i] = host_vals[o + i];
        hipLaunchKernelGGL(k_write_floats, dim3(1), dim3(128), 0, s, c, dst + o);
    }
}
void launch_add_inplace(float* dst, const float* src, long n, hipStream_t s) {
    hipLaunchKernelGGL(k_add_inplace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, src, n);
}
void launch_ipa_attn(const IpaAttnParams& p, hipStream_t s) {
    if (p.L >= 24) {   // long sequences: keys staged through LDS, one thread per query
        const int nqt = (p.L + 255) / 256;
        const long nblk = (long)p.ngroups * 4 * nqt, mtot = (long)p.ngroups * p.L;
        IpaAttnParams q = p;
        int nsplit = p.part ? (int)((64 + nblk - 1) / nblk) : 1;     // ~64 workgroups, whole 32-key tiles per slice
        const int ntile = (p.L + kIpaKT - 1) / kIpaKT;
        if (nsplit > ntile) nsplit = ntile;
        if (nsplit > 16) nsplit = 16;
        while (nsplit > 1 && (size_t)nsplit * mtot * 4 * kIpaFwdRec > p.part_floats) --nsplit;
        q.nsplit = nsplit;
        hipLaunchKernelGGL(k_ipa_attn_tiled, dim3((unsigned)nblk, (unsigned)nsplit), dim3(256), 0, s, q);
        if (nsplit > 1) hipLaunchKernelGGL(k_ipa_attn_merge, dim3((unsigned)((mtot * 4 + 255) / 256)), dim3(256), 0, s, q);
        return;
    }
    const long total = (long)p.ngroups * p.L * 4;
    hipLaunchKernelGGL(k_ipa_attn, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
}

}  // namespace mdg
