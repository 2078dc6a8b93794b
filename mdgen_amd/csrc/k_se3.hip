// k_se3.hip -- SE(3) frame algebra and the sampler's pre/post-processing, fp32, one thread per frame.
// The reference expresses each of these as ~50 element-wise aten kernels per op
// (rigid_utils.py:24-86 "written out by hand to avoid AMP downcasting"); here each is one launch.
#include "kernels.h"

// Bit-faithful fp32: no FMA contraction, so each expression rounds exactly like the reference's
// sequence of element-wise torch ops (matters where the reference relies on exact cancellation,
// e.g. the masked pre-omega torsion of residue 0 evaluates to exactly (0, 0); geometry.py:171-192).
#pragma clang fp contract(off)

namespace mdg {

struct Rig {
    float r[9];
    float t[3];
};

__device__ __forceinline__ void load_rot(const float* p, float* r) {
#pragma unroll
    for (int i = 0; i < 9; ++i) r[i] = p[i];
}
__device__ __forceinline__ void matmul3(const float* a, const float* b, float* c) {  // rigid_utils.py:24-61
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) c[3 * i + k] = a[3 * i] * b[k] + a[3 * i + 1] * b[3 + k] + a[3 * i + 2] * b[6 + k];
}
__device__ __forceinline__ void matvec3(const float* a, const float* v, float* o) {  // rigid_utils.py:64-86
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = a[3 * i] * v[0] + a[3 * i + 1] * v[1] + a[3 * i + 2] * v[2];
}
__device__ __forceinline__ void matTvec3(const float* a, const float* v, float* o) {
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = a[i] * v[0] + a[3 + i] * v[1] + a[6 + i] * v[2];
}
// Rigid.compose (rigid_utils.py:1031-1045): (R1 R2, R1 t2 + t1)
__device__ __forceinline__ Rig compose(const Rig& a, const Rig& b) {
    Rig c;
    matmul3(a.r, b.r, c.r);
    matvec3(a.r, b.t, c.t);
#pragma unroll
    for (int i = 0; i < 3; ++i) c.t[i] += a.t[i];
    return c;
}
// quat (w,x,y,z) -> rotation (rigid_utils.py:156-188)
__device__ __forceinline__ void quat2rot(float a, float b, float c, float d, float* r) {
    r[0] = a * a + b * b - c * c - d * d;
    r[1] = 2 * (b * c - a * d);
    r[2] = 2 * (b * d + a * c);
    r[3] = 2 * (b * c + a * d);
    r[4] = a * a - b * b + c * c - d * d;
    r[5] = 2 * (c * d - a * b);
    r[6] = 2 * (b * d - a * c);
    r[7] = 2 * (c * d + a * b);
    r[8] = a * a - b * b - c * c + d * d;
}
// rotation -> quat: dominant eigenvector of the symmetric 4x4 K/3 of rigid_utils.py:191-210.
// For a rotation K/3 + I/3 = (4/3) q q^T, so the column with the largest diagonal entry is already
// ~q; two further power-iteration steps absorb non-orthogonality of the input at the 1e-7 level
// (the reference uses torch.linalg.eigh).  Sign canonicalised to w >= 0 (wrapper.py:309).
__device__ __forceinline__ void rot2quat(const float* R, float* q) {
    const float xx = R[0], xy = R[1], xz = R[2], yx = R[3], yy = R[4], yz = R[5], zx = R[6], zy = R[7], zz = R[8];
    const float third = 1.0f / 3.0f;
    float M[4][4] = {{xx + yy + zz, zy - yz, xz - zx, yx - xy},
                     {zy - yz, xx - yy - zz, xy + yx, xz + zx},
                     {xz - zx, xy + yx, yy - xx - zz, yz + zy},
                     {yx - xy, xz + zx, yz + zy, zz - xx - yy}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) M[i][j] *= third;
        M[i][i] += third;
    }
    int best = 0;
    float bd = M[0][0];
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (M[i][i] > bd) {
            bd = M[i][i];
            best = i;
        }
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = best == 0 ? M[i][0] : best == 1 ? M[i][1] : best == 2 ? M[i][2] : M[i][3];
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        float n = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) n += v[i] * v[i];
        n = 1.0f / sqrtf(n);
        float u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = v[i] * n;
        if (it == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = u[i];
            break;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = M[i][0] * u[0] + M[i][1] * u[1] + M[i][2] * u[2] + M[i][3] * u[3];
    }
    const float sgn = v[0] < 0.f ? -1.0f : 1.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = v[i] * sgn;
}
// Rigid.from_3_points (rigid_utils.py:1175-1218), rotation stored row-major with columns e0,e1,e2
__device__ __forceinline__ Rig from3(const float* pnx, const float* org, const float* pxy) {
    float e0[3], e1[3], e2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        e0[i] = org[i] - pnx[i];
        e1[i] = pxy[i] - org[i];
    }
    float d = sqrtf(e0[0] * e0[0] + e0[1] * e0[1] + e0[2] * e0[2] + 1e-8f);
#pragma unroll
    for (int i = 0; i < 3; ++i) e0[i] /= d;
    const float dot = e0[0] * e1[0] + e0[1] * e1[1] + e0[2] * e1[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) e1[i] -= e0[i] * dot;
    d = sqrtf(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2] + 1e-8f);
#pragma unroll
    for (int i = 0; i < 3; ++i) e1[i] /= d;
    e2[0] = e0[1] * e1[2] - e0[2] * e1[1];
    e2[1] = e0[2] * e1[0] - e0[0] * e1[2];
    e2[2] = e0[0] * e1[1] - e0[1] * e1[0];
    Rig o;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        o.r[3 * i] = e0[i];
        o.r[3 * i + 1] = e1[i];
        o.r[3 * i + 2] = e2[i];
        o.t[i] = org[i];
    }
    return o;
}

__global__ void k_rigid_compose(long n, const float* r1, const float* t1, const float* r2, const float* t2, float* ro,
                                float* to) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rig a, b;
    load_rot(r1 + i * 9, a.r);
    load_rot(r2 + i * 9, b.r);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        a.t[k] = t1[i * 3 + k];
        b.t[k] = t2[i * 3 + k];
    }
    const Rig c = compose(a, b);
#pragma unroll
    for (int k = 0; k < 9; ++k) ro[i * 9 + k] = c.r[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) to[i * 3 + k] = c.t[k];
}

__global__ void k_rigid_invert(long n, const float* r, const float* t, float* ro, float* to) {  // :1075-1085
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float R[9], tt[3], o[3];
    load_rot(r + i * 9, R);
#pragma unroll
    for (int k = 0; k < 3; ++k) tt[k] = t[i * 3 + k];
    matTvec3(R, tt, o);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) ro[i * 9 + 3 * a + b] = R[3 * b + a];
#pragma unroll
    for (int k = 0; k < 3; ++k) to[i * 3 + k] = -o[k];
}

__global__ void k_rigid_apply(long n, long ppf, const float* r, const float* t, const float* pts, float* out,
                              int inverse) {  // :1047-1073
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * ppf) return;
    const long f = i / ppf;
    float R[9], tt[3], p[3], o[3];
    load_rot(r + f * 9, R);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        tt[k] = t[f * 3 + k];
        p[k] = pts[i * 3 + k];
    }
    if (inverse) {
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] -= tt[k];
        matTvec3(R, p, o);
    } else {
        matvec3(R, p, o);
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] += tt[k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) out[i * 3 + k] = o[k];
}

__global__ void k_quat_to_rot(long n, const float* q, int normalize, float* rot) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = q[i * 4], b = q[i * 4 + 1], c = q[i * 4 + 2], d = q[i * 4 + 3];
    if (normalize) {
        const float nn = sqrtf(a * a + b * b + c * c + d * d);
        a /= nn; b /= nn; c /= nn; d /= nn;
    }
    float R[9];
    quat2rot(a, b, c, d, R);
#pragma unroll
    for (int k = 0; k < 9; ++k) rot[i * 9 + k] = R[k];
}

// Rigid.from_3_points (rigid_utils.py:1175-1218, eps = 1e-8): one thread per frame
__global__ void k_from_3_points(long n, const float* __restrict__ pnx, const float* __restrict__ org,
                                const float* __restrict__ pxy, float* __restrict__ rot, float* __restrict__ trans) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Rig f = from3(pnx + i * 3, org + i * 3, pxy + i * 3);
#pragma unroll
    for (int k = 0; k < 9; ++k) rot[i * 9 + k] = f.r[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) trans[i * 3 + k] = f.t[k];
}

__global__ void k_rot_to_quat(long n, const float* rot, float* q) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float R[9], qq[4];
    load_rot(rot + i * 9, R);
    rot2quat(R, qq);
#pragma unroll
    for (int k = 0; k < 4; ++k) q[i * 4 + k] = qq[k];
}

// rel7[n] = to_tensor_7( rigid1^-1 o rigid2 )  (latent_model.py:194-195), quaternion sign w >= 0
__global__ void k_rel7(const float* r1, const float* t1, const float* r2, const float* t2, float* out7, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float A[9], B[9], at[3], bt[3];
    load_rot(r1 + i * 9, A);
    load_rot(r2 + i * 9, B);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        at[k] = t1[i * 3 + k];
        bt[k] = t2[i * 3 + k];
    }
    float At[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) At[3 * a + b] = A[3 * b + a];
    float R[9], d[3], tt[3], q[4];
    matmul3(At, B, R);
    // (R1^T, -R1^T t1) o (R2, t2) = (R1^T R2, R1^T t2 - R1^T t1) evaluated as the reference does
    float ia[3];
    matTvec3(A, at, ia);
    matvec3(At, bt, d);
#pragma unroll
    for (int k = 0; k < 3; ++k) tt[k] = d[k] - ia[k];
    rot2quat(R, q);
#pragma unroll
    for (int k = 0; k < 4; ++k) out7[i * 7 + k] = q[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) out7[i * 7 + 4 + k] = tt[k];
}

// NewMDGenWrapper.prep_batch latents (wrapper.py:298-327,339-342,362) + get_offsets (utils.py:7-14)
// BCAST: rots / trans / tors hold ONE frame per (b, l) that stands for every t (the rollout's conditioning frame
// expanded over T, sim_inference.py:72-79) -- same arithmetic on the same values, without materialising the copies.
// cond_interval > 0: additionally every cond_interval-th frame is a conditioning frame (wrapper.py:343-344, the
// upsampling models).
__global__ void k_prep_latents(int B, int T, int L, int tps, int bcast, int cond_interval, const float* rots, const float* trans,
                               const float* tors, float* latents, float* x_cond, int64_t* x_cond_mask) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long N = (long)B * T * L;
    if (i >= N) return;
    const int l = (int)(i % L);
    const int t = (int)((i / L) % T);
    const int b = (int)(i / ((long)L * T));
    const int D = tps ? 28 : 21;
    const long src = bcast ? (long)b * L + l : i;
    float R[9], tt[3];
    load_rot(rots + src * 9, R);
#pragma unroll
    for (int k = 0; k < 3; ++k) tt[k] = trans[src * 3 + k];
    float lat[28];
    const int nref = tps ? 2 : 1;
    for (int ref = 0; ref < nref; ++ref) {
        const long j = bcast ? src : ((long)b * T + (ref == 0 ? 0 : T - 1)) * L + l;
        float R0[9], t0[3];
        load_rot(rots + j * 9, R0);
#pragma unroll
        for (int k = 0; k < 3; ++k) t0[k] = trans[j * 3 + k];
        float R0t[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) R0t[3 * a + c] = R0[3 * c + a];
        float Rr[9], it[3], d[3], q[4];
        matmul3(R0t, R, Rr);
        matTvec3(R0, t0, it);   // R0^T t0   (invert: -R0^T t0)
        matvec3(R0t, tt, d);    // R0^T t
        rot2quat(Rr, q);
#pragma unroll
        for (int k = 0; k < 4; ++k) lat[ref * 7 + k] = q[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) lat[ref * 7 + 4 + k] = d[k] - it[k];
    }
    const int toff = tps ? 14 : 7;
#pragma unroll
    for (int k = 0; k < 14; ++k) lat[toff + k] = tors[src * 14 + k];
    const bool cond = (t == 0) || (tps && t == T - 1) || (cond_interval > 0 && t % cond_interval == 0);
    for (int k = 0; k < D; ++k) {
        if (latents) latents[i * D + k] = lat[k];
        x_cond[i * D + k] = cond ? lat[k] : 0.f;
    }
    x_cond_mask[i] = cond ? 1 : 0;
}

// inference() tail (wrapper.py:456-478) + frames_torsions_to_atom14 (geometry.py:61-79,236-334)
__global__ void k_samples_to_atom14(int B, int T, int L, int D, int tps, const float* samples, const float* rot0,
                                    const float* trans0, const int64_t* seqres, const float* default_frames,
                                    const float* lit_positions, const int64_t* atom14_group, const float* atom14_mask,
                                    float* atom14, int out_T, int out_t0) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long N = (long)B * T * L;
    if (i >= N) return;
    const int l = (int)(i % L);
    const int b = (int)(i / ((long)L * T));
    // output row: frame out_t0 + t of a trajectory of out_T frames (out_T == T, out_t0 == 0: the block itself)
    const long io = ((long)b * out_T + out_t0 + (int)((i / L) % T)) * L + l;
    const float* s = samples + i * D;
    // frames = rigids[:,0:1] o from_tensor_7(offsets, normalize_quats=True)
    Rig off, f0;
    {
        float a = s[0], bq = s[1], c = s[2], d = s[3];
        const float nn = sqrtf(a * a + bq * bq + c * c + d * d);
        a /= nn; bq /= nn; c /= nn; d /= nn;
        quat2rot(a, bq, c, d, off.r);
        off.t[0] = s[4]; off.t[1] = s[5]; off.t[2] = s[6];
    }
    load_rot(rot0 + ((long)b * L + l) * 9, f0.r);
#pragma unroll
    for (int k = 0; k < 3; ++k) f0.t[k] = trans0[((long)b * L + l) * 3 + k];
    const Rig bbf = compose(f0, off);
    const int aa = (int)seqres[(long)b * L + l];
    const float* tp = s + (tps ? 14 : 7);
    // 8 rigid-group frames (geometry.py:273-334)
    Rig grp[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        float sn = 0.f, cs = 1.0f;  // backbone group: (sin,cos) = (0,1)
        if (g > 0) {
            const float a = tp[2 * (g - 1)], c = tp[2 * (g - 1) + 1];
            const float nn = sqrtf(a * a + c * c);   // wrapper.py:474-476 (torch.linalg.norm)
            sn = a / nn;
            cs = c / nn;
        }
        const float* d4 = default_frames + ((long)aa * 8 + g) * 16;
        Rig dr, rx;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int c = 0; c < 3; ++c) dr.r[3 * a + c] = d4[4 * a + c];
            dr.t[a] = d4[4 * a + 3];
        }
        rx.r[0] = 1; rx.r[1] = 0; rx.r[2] = 0;
        rx.r[3] = 0; rx.r[4] = cs; rx.r[5] = -sn;
        rx.r[6] = 0; rx.r[7] = sn; rx.r[8] = cs;
        rx.t[0] = rx.t[1] = rx.t[2] = 0.f;
        grp[g] = compose(dr, rx);
    }
    grp[5] = compose(grp[4], grp[5]);
    grp[6] = compose(grp[5], grp[6]);
    grp[7] = compose(grp[6], grp[7]);
#pragma unroll
    for (int g = 0; g < 8; ++g) grp[g] = compose(bbf, grp[g]);
    for (int a = 0; a < 14; ++a) {
        const int g = (int)atom14_group[aa * 14 + a];
        const float m = atom14_mask[aa * 14 + a];
        const float* lp = lit_positions + ((long)aa * 14 + a) * 3;
        Rig fr = grp[0];
#pragma unroll
        for (int gg = 1; gg < 8; ++gg)
            if (g == gg) fr = grp[gg];
        float o[3];
        matvec3(fr.r, lp, o);
#pragma unroll
        for (int k = 0; k < 3; ++k) atom14[(io * 14 + a) * 3 + k] = (o[k] + fr.t[k]) * m;
    }
}

// Rollout glue (sim_inference.py:91-96): atom14_to_frames (geometry.py:218-231) + atom14_to_atom37
// (geometry.py:9-27) + atom37_to_torsions (geometry.py:82-202), one thread per residue.
__device__ __forceinline__ void atom37_pos(const float* a14, const int64_t* a37to14, const float* a37mask, int aa,
                                           int idx, float* o) {
    const int s = (int)a37to14[aa * 37 + idx];
    const float m = a37mask[aa * 37 + idx];
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = a14[s * 3 + k] * m;
}

// in_bstride: floats between consecutive batch elements of `atom14` (L * 42 when it is a plain (B, L, 14, 3) array;
// larger when the frame is picked out of a (B, frames, L, 14, 3) trajectory)
__global__ void k_atom14_to_cond(int B, int L, const float* atom14, long in_bstride, const int64_t* seqres,
                                 const int64_t* a37to14, const float* a37mask, const int64_t* chi_idx,
                                 const float* chi_mask, float* rots, float* trans, float* tors, float* tmask) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * L) return;
    const int l = (int)(i % L);
    const float* a14 = atom14 + (i / L) * in_bstride + (long)l * 42;
    const int aa = (int)seqres[i];
    {   // backbone frame: from_3_points(C, CA, N) right-composed with diag(-1, 1, -1)
        const Rig f = from3(a14 + 6, a14 + 3, a14 + 0);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            rots[i * 9 + 3 * a] = -f.r[3 * a];
            rots[i * 9 + 3 * a + 1] = f.r[3 * a + 1];
            rots[i * 9 + 3 * a + 2] = -f.r[3 * a + 2];
            trans[i * 3 + a] = f.t[a];
        }
    }
    float P[7][4][3];
    float M[7];
    const bool hasprev = l > 0;
    const float* p14 = a14 - 42;
    const int pa = hasprev ? (int)seqres[i - 1] : 0;
    float z3[3] = {0.f, 0.f, 0.f};
    auto prev = [&](int idx, float* o) {
        if (hasprev) atom37_pos(p14, a37to14, a37mask, pa, idx, o);
        else { o[0] = z3[0]; o[1] = z3[1]; o[2] = z3[2]; }
    };
    auto prevm = [&](int idx) { return hasprev ? a37mask[pa * 37 + idx] : 0.f; };
    auto curm = [&](int idx) { return a37mask[aa * 37 + idx]; };
    prev(1, P[0][0]); prev(2, P[0][1]);
    atom37_pos(a14, a37to14, a37mask, aa, 0, P[0][2]); atom37_pos(a14, a37to14, a37mask, aa, 1, P[0][3]);
    M[0] = prevm(1) * prevm(2) * curm(0) * curm(1);
    prev(2, P[1][0]);
    atom37_pos(a14, a37to14, a37mask, aa, 0, P[1][1]); atom37_pos(a14, a37to14, a37mask, aa, 1, P[1][2]);
    atom37_pos(a14, a37to14, a37mask, aa, 2, P[1][3]);
    M[1] = prevm(2) * curm(0) * curm(1) * curm(2);
    atom37_pos(a14, a37to14, a37mask, aa, 0, P[2][0]); atom37_pos(a14, a37to14, a37mask, aa, 1, P[2][1]);
    atom37_pos(a14, a37to14, a37mask, aa, 2, P[2][2]); atom37_pos(a14, a37to14, a37mask, aa, 4, P[2][3]);
    M[2] = curm(0) * curm(1) * curm(2) * curm(4);
    for (int c = 0; c < 4; ++c) {
        float mm = chi_mask[aa * 4 + c];
        for (int k = 0; k < 4; ++k) {
            const int idx = (int)chi_idx[(aa * 4 + c) * 4 + k];
            atom37_pos(a14, a37to14, a37mask, aa, idx, P[3 + c][k]);
            mm *= curm(idx);
        }
        M[3 + c] = mm;
    }
    for (int k = 0; k < 7; ++k) {
        const Rig f = from3(P[k][1], P[k][2], P[k][0]);
        float it[3], rel[3];
        matTvec3(f.r, f.t, it);          // Rigid.invert(): (R^T, -(R^T t))
        matTvec3(f.r, P[k][3], rel);     // .apply(p): R^T p + (-(R^T t))
#pragma unroll
        for (int a = 0; a < 3; ++a) rel[a] = rel[a] + (-it[a]);
        float sn = rel[2], cs = rel[1];
        const float den = sqrtf(sn * sn + cs * cs + 1e-8f);
        sn /= den;
        cs /= den;
        if (k == 2) { sn = -sn; cs = -cs; }
        tors[(i * 7 + k) * 2] = sn;
        tors[(i * 7 + k) * 2 + 1] = cs;
        tmask[i * 7 + k] = M[k];
    }
}

// ---- launchers ----------------------------------------------------------------------------------
#define GRID1D(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, s
void launch_rigid_compose(long n, const float* r1, const float* t1, const float* r2, const float* t2, float* ro,
                          float* to, hipStream_t s) {
    hipLaunchKernelGGL(k_rigid_compose, GRID1D(n), n, r1, t1, r2, t2, ro, to);
}
void launch_rigid_invert(long n, const float* r, const float* t, float* ro, float* to, hipStream_t s) {
    hipLaunchKernelGGL(k_rigid_invert, GRID1D(n), n, r, t, ro, to);
}
void launch_rigid_apply(long n, long ppf, const float* r, const float* t, const float* pts, float* out, int inverse,
                        hipStream_t s) {
    hipLaunchKernelGGL(k_rigid_apply, GRID1D(n * ppf), n, ppf, r, t, pts, out, inverse);
}
void launch_quat_to_rot(long n, const float* q, int normalize, float* rot, hipStream_t s) {
    hipLaunchKernelGGL(k_quat_to_rot, GRID1D(n), n, q, normalize, rot);
}
void launch_from_3_points(long n, const float* pnx, const float* org, const float* pxy, float* rot, float* trans,
                          hipStream_t s) {
    hipLaunchKernelGGL(k_from_3_points, GRID1D(n), n, pnx, org, pxy, rot, trans);
}
void launch_rot_to_quat(long n, const float* rot, float* q, hipStream_t s) {
    hipLaunchKernelGGL(k_rot_to_quat, GRID1D(n), n, rot, q);
}
void launch_rel7(const float* r1, const float* t1, const float* r2, const float* t2, float* out7, long n, hipStream_t s) {
    hipLaunchKernelGGL(k_rel7, GRID1D(n), r1, t1, r2, t2, out7, n);
}
void launch_prep_latents(int B, int T, int L, int tps, int bcast, int cond_interval, const float* rots, const float* trans,
                         const float* tors, float* latents, float* x_cond, int64_t* x_cond_mask, hipStream_t s) {
    const long n = (long)B * T * L;
    hipLaunchKernelGGL(k_prep_latents, GRID1D(n), B, T, L, tps, bcast, cond_interval, rots, trans, tors, latents, x_cond, x_cond_mask);
}
void launch_samples_to_atom14(int B, int T, int L, int D, int tps, const float* samples, const float* rot0,
                              const float* trans0, const int64_t* seqres, const float* default_frames,
                              const float* lit_positions, const int64_t* atom14_group, const float* atom14_mask,
                              float* atom14, int out_T, int out_t0, hipStream_t s) {
    const long n = (long)B * T * L;
    hipLaunchKernelGGL(k_samples_to_atom14, GRID1D(n), B, T, L, D, tps, samples, rot0, trans0, seqres, default_frames,
                       lit_positions, atom14_group, atom14_mask, atom14, out_T, out_t0);
}
void launch_atom14_to_cond(int B, int L, const float* atom14, long in_bstride, const int64_t* seqres,
                           const int64_t* a37to14, const float* a37mask, const int64_t* chi_idx, const float* chi_mask,
                           float* rots, float* trans, float* tors, float* tmask, hipStream_t s) {
    const long n = (long)B * L;
    hipLaunchKernelGGL(k_atom14_to_cond, GRID1D(n), B, L, atom14, in_bstride, seqres, a37to14, a37mask, chi_idx,
                       chi_mask, rots, trans, tors, tmask);
}

}  // namespace mdg
