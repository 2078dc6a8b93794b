// The two big products of the training step in bf16-operand mode, as 128 x 384 workgroup tiles:
//
//   k16_linear_wide   C[128 tokens][384 outputs] = A[128][k] W[384][k]^T       (forward layers, dX through a turned weight)
//   k16_dw_wide       dW[128 m][384 k] += dY[n][128]^T X[n][384]               (weight + bias gradients, split over n)
//
// Why this shape.  The trunk is 384 wide, so with 128 x 128 tiles every activation row is needed by three workgroups;
// measured with the PMC counters (scripts/pmc_train.sh) those three read it from HBM three times (a 64 000 x 384 x 384
// layer: 274 MiB fetched for a 94 MiB input -- 64 resident workgroups x 196 KB of rows do not fit an XCD's 4 MB L2), and
// the weight gradient re-read each operand per tile of the other one (1.7 GiB fetched for 375 MiB of operands).  With
// 384 output columns per workgroup the token operand is read from HBM exactly once; what is re-read (the weight in the
// forward product, X per 128 gradient rows) is the small or shared side and is served by the L2.
//
// Eight waves, one workgroup per CU: wave (wr, wc) owns 64 rows x 96 columns (2 x 3 MFMA tiles, 96 accumulator
// registers).  A step is 64 of the contraction: sixteen 16-byte loads per thread issued together (each load instruction of
// a wave covers whole 256- / 512-byte row segments), 24 MFMAs per wave on the previous step's LDS tiles, then the loaded
// registers are rounded to bf16 into the other LDS buffer; one barrier per step.  LDS: 2 x (128 + 384) rows x 144 bytes.
#include "common.h"
#include "kernels.h"
#include "linear.h"
#include "rows.h"

namespace mdg {
namespace {

constexpr int kWideRows = 128, kWideCols = 384, kWideBK = 64, kWideRowB = 144;
constexpr int kWideP = kWideRows * kWideRowB, kWideQ = kWideCols * kWideRowB;   // bytes of one buffer's P / Q tile

// acc[t][u] += P[64 wr + 32 t .. ][0..63] . Q[96 wc + 32 u .. ][0..63]^T over the four 16-wide k-steps of the LDS tiles.
// `issue(ks)` runs in front of each k-step's MFMAs (a hook for the stamps build; spreading the next step's sixteen global
// loads over the four k-steps instead of issuing them in one block before the MFMAs was measured and is not kept: the last
// quarter then leaves late and the conversion phase waits out its whole latency, 48.8 vs 47.7 ms per training step).
template <typename Issue>
__device__ __forceinline__ void wide_mma(const unsigned char* P, const unsigned char* Q, f32x16 (&acc)[2][3], int wr, int wc,
                                         int lane, Issue issue) {
    const int i = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        issue(ks);
        bf16x8 a[2], b[3];
#pragma unroll
        for (int t = 0; t < 2; ++t) a[t] = *reinterpret_cast<const bf16x8*>(P + (wr * 64 + t * 32 + i) * kWideRowB + ks * 32 + kh * 16);
#pragma unroll
        for (int u = 0; u < 3; ++u) b[u] = *reinterpret_cast<const bf16x8*>(Q + (wc * 96 + u * 32 + i) * kWideRowB + ks * 32 + kh * 16);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 3; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b[u], acc[t][u], 0, 0, 0);
    }
}

__device__ __forceinline__ u32x2 pack4(float a, float b, float c, float d) { return u32x2{pack_bf16(a, b), pack_bf16(c, d)}; }

}  // namespace

#ifdef MDGEN_DEV_WIDE_STAMPS
__device__ unsigned long long g_wide_stamps[512 * 2 * 24];   // s_memtime stamps of waves 0 and 5 of the first 512 workgroups
extern "C" int mdgen_dev_wide_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wide_stamps), (size_t)n * 8, 0, hipMemcpyDeviceToHost);
}
#endif

// Tile order as in k16_linear_fast: the column groups of one 128-row slice run back to back on one XCD.
// Requires (launcher): k % 64 == 0, 16-byte aligned operands and row strides, weight stored [m][k].
// One tile (virtual block index vb of the XCD-aware order).  lds: [2][P | Q].
__device__ __forceinline__ void linear_wide_tile(const LinearParams& p, int vb, int nrt, int ncg, unsigned char* lds) {
    const int xcd = vb & 7, slot = vb >> 3;
    const int rt = (slot / ncg) * 8 + xcd, cg = slot % ncg;
    if (rt >= nrt) return;
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const long row0 = (long)rt * kWideRows;
    const int colt = cg * kWideCols;
    const int wr = w >> 2, wc = w & 3;
    f32x16 acc[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = opaque_zero();
    const int r0 = tid >> 4, piece = tid & 15;        // rows r0 + 32 q, k0 + 4 piece .. + 3
    const int sg = p.seg_cols ? colt / p.seg_cols : 0;          // segments are whole numbers of column groups (launcher)
    const float* wbase = (p.seg_cols ? p.w_seg[sg] : p.w) + 4 * piece;
    const int ccol = colt - sg * p.seg_cols, mseg = p.seg_cols ? p.seg_cols : p.m;
    const float* abase = p.a + 4 * piece;
    // rows / columns past the end: clamped loads, results never stored
    long aoff[4], woff[12];
#pragma unroll
    for (int q = 0; q < 4; ++q) aoff[q] = (row0 + r0 + 32 * q < p.n ? row0 + r0 + 32 * q : p.n - 1) * p.lda;
#pragma unroll
    for (int q = 0; q < 12; ++q) woff[q] = (long)(ccol + r0 + 32 * q < mseg ? ccol + r0 + 32 * q : mseg - 1) * p.ldw;
    f32x4 av[4], wv[12];
    auto fetch_quarter = [&](int k0, int j) {   // one A row and three W rows of this thread
        av[j] = *reinterpret_cast<const f32x4*>(abase + aoff[j] + k0);
#pragma unroll
        for (int q = 3 * j; q < 3 * j + 3; ++q) wv[q] = *reinterpret_cast<const f32x4*>(wbase + woff[q] + k0);
    };
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fetch_quarter(k0, j);
    };
    auto stage = [&](int buf) {
        unsigned char* P = lds + buf * (kWideP + kWideQ);
        unsigned char* Q = P + kWideP;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<u32x2*>(P + (r0 + 32 * q) * kWideRowB + piece * 8) = pack4(av[q][0], av[q][1], av[q][2], av[q][3]);
#pragma unroll
        for (int q = 0; q < 12; ++q)
            *reinterpret_cast<u32x2*>(Q + (r0 + 32 * q) * kWideRowB + piece * 8) = pack4(wv[q][0], wv[q][1], wv[q][2], wv[q][3]);
    };
#ifdef MDGEN_DEV_WIDE_STAMPS
    unsigned long long st[24];
    int sti = 0;
#define WIDE_STAMP() do { if (sti < 24) st[sti++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WIDE_STAMP() do { } while (0)
#endif
    WIDE_STAMP();
    fetch(0);
    stage(0);
    __syncthreads();
    WIDE_STAMP();
    int buf = 0;
    for (int k0 = 0; k0 < p.k; k0 += kWideBK, buf ^= 1) {
        const bool more = k0 + kWideBK < p.k;
        const unsigned char* P = lds + buf * (kWideP + kWideQ);
#if defined(MDGEN_DEV_WIDE_NOMMA)
        if (more) fetch(k0 + kWideBK);
#elif defined(MDGEN_DEV_WIDE_NOLOAD)
        wide_mma(P, P + kWideP, acc, wr, wc, lane, [](int) {});
#else
#if defined(MDGEN_DEV_WIDE_STAMPS)
        wide_mma(P, P + kWideP, acc, wr, wc, lane, [&](int ks) {
            if (k0 < 3 * kWideBK) WIDE_STAMP();
            if (more) fetch_quarter(k0 + kWideBK, ks);
        });
#else
        if (more) fetch(k0 + kWideBK);
        wide_mma(P, P + kWideP, acc, wr, wc, lane, [](int) {});
#endif
#endif
#ifdef MDGEN_DEV_WIDE_STAMPS
        if (k0 < 3 * kWideBK) WIDE_STAMP();
        if (more) stage(buf ^ 1);
        if (k0 < 3 * kWideBK) WIDE_STAMP();
        __syncthreads();
        if (k0 < 3 * kWideBK) WIDE_STAMP();
#else
        if (more) stage(buf ^ 1);
        __syncthreads();
#endif
    }
#ifdef MDGEN_DEV_WIDE_NOSTORE
    if (acc[0][0][0] == 1.2345e33f) linear_epilogue(p, acc, row0, colt, wr, wc);
    else if (tid == 0 && acc[1][2][5] == 5.4321e33f) p.c[row0] = 0.f;
#else
    linear_epilogue(p, acc, row0, colt, wr, wc);
#endif
#ifdef MDGEN_DEV_WIDE_STAMPS
    WIDE_STAMP();
    if (p.k == 384 && p.m == 384 && !p.seg_cols && vb < 512 && (tid & 63) == 0 && (w == 0 || w == 5))
        for (int i = 0; i < 24; ++i) g_wide_stamps[(vb * 2 + (w ? 1 : 0)) * 24 + i] = i < sti ? st[i] : 0ull;
#endif
}

// (A persistent form -- 256 workgroups walking the tiles, so that a tile's store phase could overlap the next tile's loads --
// and a phase offset between neighbouring workgroups were both measured and changed nothing: 199.7 / 198.8 vs 199.1 us average;
// what adds up is per CU)
__global__ __launch_bounds__(512) void k16_linear_wide(const LinearParams p, int nrt, int ncg) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (kWideP + kWideQ)];
    linear_wide_tile(p, blockIdx.x, nrt, ncg, lds);
}

// ---- the same tile with the weight brought in by LDS-DMA ---------------------------------------------------------------
// What the ablation builds of k16_linear_wide show: per CU, the time of the global loads that return
// through the register file and the time of the LDS-fed MFMA part ADD UP (35 + 75 -> 122 us).  Three quarters of those
// register-returned bytes are the weight, which every workgroup re-reads (from L2) and re-rounds to bf16.  Here it is
// rounded and laid out ONCE per use by k16_pack_wstream -- 1 KiB MFMA B-operand fragments in exactly the order a
// workgroup consumes them: [column group][k-step of 64][16-wide k-step ks][32-column tile] -- and the kernel's eight waves
// copy a k-step's 48 fragments straight into LDS (global_load_lds_dwordx4: no registers, no conversion, no ds_write;
// fragment reads are lane-linear, hence conflict-free without padding).  Only the 128 token rows still pass through
// registers (4 loads per thread per step instead of 16).
constexpr int kWideQD = 48 * 1024;   // one k-step of the weight stream: 4 ks x 12 column tiles x 1 KiB

__global__ __launch_bounds__(256) void k16_pack_wstream(const float* __restrict__ s0, const float* __restrict__ s1,
                                                        const float* __restrict__ s2, int seg, int ld, int m, int k, int turned,
                                                        u32x4* __restrict__ out) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)m * k / 8) return;
    const int lane = (int)(gid & 63);
    const long frag = gid >> 6;
    const int nk = k / 64;
    const int ct = (int)(frag % 12), ks = (int)((frag / 12) & 3), kstep = (int)((frag / 48) % nk), cg = (int)(frag / (48L * nk));
    const int col = cg * kWideCols + ct * 32 + (lane & 31);
    const int kk0 = kstep * 64 + ks * 16 + 8 * (lane >> 5);
    float v[8];
    if (!turned) {
        const int si = col / seg;
        const float* src = (si == 0 ? s0 : si == 1 ? s1 : s2) + (long)(col - si * seg) * ld + kk0;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = a[j];
            v[4 + j] = b[j];
        }
    } else {
        const int si = kk0 / seg;    // 8 consecutive kk never straddle a segment (seg is a multiple of 64)
        const float* src = (si == 0 ? s0 : si == 1 ? s1 : s2) + (long)(kk0 - si * seg) * ld + col;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(long)j * ld];
    }
    out[gid] = u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
}

bool launch16_pack_wstream(const float* const* src, int nsrc, int seg, int ld, long n, int m, int k, int turned, void* out,
                           hipStream_t s) {
    const auto al = [](const void* q) { return ((unsigned long long)q & 15) == 0; };
    if (n < 1024 || m % kWideCols || k % 64 || seg % 64 || nsrc < 1 || nsrc > 3 || (ld & 3)) return false;
    if ((long)nsrc * seg != (turned ? k : m)) return false;
    for (int i = 0; i < nsrc; ++i)
        if (!al(src[i])) return false;
    const long items = (long)m * k / 8;
    hipLaunchKernelGGL(k16_pack_wstream, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, src[0], src[nsrc > 1 ? 1 : 0],
                       src[nsrc > 2 ? 2 : 0], seg, ld, m, k, turned, static_cast<u32x4*>(out));
    return true;
}

// Requires (launcher): p.wpack (k16_pack_wstream of this layer), k % 64 == 0, m % 384 == 0, 16-byte aligned token operand.
// ABF: the token operand is stored as bf16 rows (p.a reinterpreted, lda in elements) -- a tensor that is only ever a GEMM
// operand (the GELU output) is written rounded by its producer: the same values enter the MFMA, half the bytes cross HBM
// and the register file, no conversion here.
template <bool ABF>
__global__ __launch_bounds__(512) void k16_linear_wdma(const LinearParams p, int nrt, int ncg) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * (kWideP + kWideQD)];   // [2][Q (48 KiB) | P]
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int rt = (slot / ncg) * 8 + xcd, cg = slot % ncg;
    if (rt >= nrt) return;
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const long row0 = (long)rt * kWideRows;
    const int colt = cg * kWideCols;
    const int wr = w >> 2, wc = w & 3;
    f32x16 acc[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = opaque_zero();
    // fp32 rows: token rows r0 + 32 q (q < 4), k0 + 4 piece .. + 3;  bf16 rows: token rows rb + 64 q (q < 2), k0 + 8 pb .. + 7
    const int r0 = tid >> 4, piece = tid & 15;
    const int rb = tid >> 3, pb = tid & 7;
    const float* abase = p.a + 4 * piece;
    const uint16_t* abase16 = reinterpret_cast<const uint16_t*>(p.a) + 8 * pb;
    long aoff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // past the end: clamped, never stored
        const long r = ABF ? row0 + rb + 64 * (q & 1) : row0 + r0 + 32 * q;
        aoff[q] = (r < p.n ? r : p.n - 1) * p.lda;
    }
    f32x4 av[4];
    u32x4 av16[2];
    auto fetch = [&](int k0) {
        if (ABF) {
#pragma unroll
            for (int q = 0; q < 2; ++q) av16[q] = *reinterpret_cast<const u32x4*>(abase16 + aoff[q] + k0);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) av[q] = *reinterpret_cast<const f32x4*>(abase + aoff[q] + k0);
        }
    };
    auto stage = [&](int buf) {
        unsigned char* P = lds + buf * (kWideP + kWideQD) + kWideQD;
        if (ABF) {
#pragma unroll
            for (int q = 0; q < 2; ++q) *reinterpret_cast<u32x4*>(P + (rb + 64 * q) * kWideRowB + pb * 16) = av16[q];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<u32x2*>(P + (r0 + 32 * q) * kWideRowB + piece * 8) = pack4(av[q][0], av[q][1], av[q][2], av[q][3]);
        }
    };
    // this wave's six fragments of a k-step: two DMA groups (an instruction offset moves both ends, rows.h dma_frag)
    const int nk = p.k / kWideBK;
    const unsigned char* qsrc = p.wpack + ((long)cg * nk) * kWideQD + wu * 6144;
    const unsigned voff = lane * 16;
    const unsigned qdst = lds_addr(lds) + wu * 6144;
    auto issue_q = [&](int kstep, int buf) {
        const unsigned char* src = qsrc + (long)kstep * kWideQD;
        const unsigned dst = qdst + buf * (kWideP + kWideQD);
        dma_frag<0, true>(src, voff, dst);
        dma_frag<1024, false>(src, voff, dst);
        dma_frag<2048, false>(src, voff, dst);
        dma_frag<3072, false>(src, voff, dst);
        dma_frag<0, true>(src + 4096, voff, dst + 4096);
        dma_frag<1024, false>(src + 4096, voff, dst + 4096);
    };
    const int i = lane & 31, kh = lane >> 5;
    fetch(0);
    issue_q(0, 0);
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMAs are inline asm: hipcc neither counts nor drains them
    __syncthreads();
    int buf = 0;
    for (int kstep = 0; kstep < nk; ++kstep, buf ^= 1) {
        const bool more = kstep + 1 < nk;
        if (more) {   // the register loads first: the waits hipcc puts between them (it cannot see the DMAs) must not find DMAs in the queue
            fetch((kstep + 1) * kWideBK);
            issue_q(kstep + 1, buf ^ 1);
        }
        const unsigned char* Q = lds + buf * (kWideP + kWideQD);
        const unsigned char* P = Q + kWideQD;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 a[2], b[3];
#pragma unroll
            for (int t = 0; t < 2; ++t) a[t] = *reinterpret_cast<const bf16x8*>(P + (wr * 64 + t * 32 + i) * kWideRowB + ks * 32 + kh * 16);
#pragma unroll
            for (int u = 0; u < 3; ++u) b[u] = *reinterpret_cast<const bf16x8*>(Q + (ks * 12 + wc * 3 + u) * 1024 + lane * 16);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 3; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b[u], acc[t][u], 0, 0, 0);
        }
        if (more) stage(buf ^ 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    linear_epilogue(p, acc, row0, colt, wr, wc);
}

// part[z][m][k] = sum_{n in slice z} dY[n][m] X[n][k] for the 128 gradient rows m0.. and the 384 columns k0.. of this
// workgroup; bpart[z][m] = sum_n dY[n][m] (bias gradient; the workgroups of the first column group, when bpart != nullptr).
// Both operands are read along their contiguous dimension and transposed on the way into LDS: a thread carries four adjacent
// token rows of four columns, so that the four token values of a column are one 8-byte LDS store.
// Tile order: the (row tile, column group) pairs of one n-slice run back to back on one XCD.
// Requires (launcher): ldy, ldx, m, k multiples of 8, 16-byte aligned operands.
// XBF: X is stored as bf16 rows (x reinterpreted, ldx in elements): see k16_linear_wdma.  DYBF: dY likewise (round 6: the q | k | v
// gradients the sequence-resident attention backward writes as bf16; the bias gradient then sums the rounded values).
template <bool XBF, bool DYBF>
__global__ __launch_bounds__(512) void k16_dw_wide(const float* __restrict__ dy, int ldy, const float* __restrict__ x, int ldx,
                                                   long n, int m, int k, int nsplit, float* __restrict__ part,
                                                   float* __restrict__ bpart) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (kWideP + kWideQ)];
    const int mt = (m + kWideRows - 1) / kWideRows, kg = (k + kWideCols - 1) / kWideCols, nt = mt * kg;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int z = (slot / nt) * 8 + xcd, tile = slot % nt;
    if (z >= nsplit) return;
    const int m0 = (tile % mt) * kWideRows, k0 = (tile / mt) * kWideCols;
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const long per = ((n + nsplit - 1) / nsplit + kWideBK - 1) / kWideBK * kWideBK;
    const long nlo = (long)z * per, nhi = nlo + per < n ? nlo + per : n;
    const int wr = w >> 2, wc = w & 3;
    f32x16 acc[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = opaque_zero();
    // token rows 4 g .. 4 g + 3 of the step; columns 4 pc .. (+ 128 e for X).  Lane bits: pc = l[5:4] l[1:0] (+ 16 per odd
    // wave), g = l[3:2] (+ 4 per wave pair): a load instruction of a wave still covers whole 256-byte row segments, and
    // the sixteen lanes that share an LDS store cycle (ds_write_b64: contiguous groups of 16) spread over four row
    // groups x four column pieces = 2-way bank conflicts; with pc = tid & 31 they were sixteen column pieces of one row
    // group, rows 4 x 144 bytes apart: 8-way (SQ_LDS_BANK_CONFLICT = 75 % of the kernel's LDS cycles).
    const int g = 4 * (w >> 1) + ((lane >> 2) & 3), pc = 16 * (w & 1) + ((lane & 3) | ((lane >> 4) << 2));
    const int mc = m0 + 4 * pc < m ? m0 + 4 * pc : 0;
    int kc[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) kc[e] = k0 + 4 * (pc + 32 * e) < k ? k0 + 4 * (pc + 32 * e) : 0;   // past the end: clamped, never stored
    f32x4 av[4], bv[3][4];
    u32x2 bh[3][4];    // XBF: four bf16 per (piece, token row)
    u32x2 ah[4];       // DYBF: four bf16 per token row
    const uint16_t* x16 = reinterpret_cast<const uint16_t*>(x);
    const uint16_t* dy16 = reinterpret_cast<const uint16_t*>(dy);
    auto fetch_quarter = [&](long n0, int r) {   // token row 4 g + r of the step: one dY piece, three X pieces
        {
            const long rw = n0 + 4 * g + r;
            const long row = rw < nhi ? rw : nhi - 1;
            if (DYBF) ah[r] = *reinterpret_cast<const u32x2*>(dy16 + row * ldy + mc);
            else av[r] = *reinterpret_cast<const f32x4*>(dy + row * ldy + mc);
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                if (XBF) bh[e][r] = *reinterpret_cast<const u32x2*>(x16 + row * ldx + kc[e]);
                else bv[e][r] = *reinterpret_cast<const f32x4*>(x + row * ldx + kc[e]);
            }
            if (rw >= nhi) {
                av[r] = f32x4{0.f, 0.f, 0.f, 0.f};
                ah[r] = u32x2{0u, 0u};
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    bv[e][r] = f32x4{0.f, 0.f, 0.f, 0.f};
                    bh[e][r] = u32x2{0u, 0u};
                }
            }
        }
    };
    auto fetch = [&](long n0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) fetch_quarter(n0, r);
    };
    const bool colsum = bpart && k0 == 0;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    auto stage = [&](int buf) {
        unsigned char* P = lds + buf * (kWideP + kWideQ);
        unsigned char* Q = P + kWideP;
        if (colsum)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (DYBF) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (j & 1) ? bf16_hi(ah[r][j >> 1]) : bf16_lo(ah[r][j >> 1]);
                    cs[j] += (v[0] + v[1]) + (v[2] + v[3]);
                } else {
                    cs[j] += (av[0][j] + av[1][j]) + (av[2][j] + av[3][j]);
                }
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (DYBF) {   // as the XBF gather below
                const unsigned sel = (j & 1) ? 0x07060302u : 0x05040100u;
                *reinterpret_cast<u32x2*>(P + (4 * pc + j) * kWideRowB + 8 * g) =
                    u32x2{__builtin_amdgcn_perm(ah[1][j >> 1], ah[0][j >> 1], sel), __builtin_amdgcn_perm(ah[3][j >> 1], ah[2][j >> 1], sel)};
            } else {
                *reinterpret_cast<u32x2*>(P + (4 * pc + j) * kWideRowB + 8 * g) = pack4(av[0][j], av[1][j], av[2][j], av[3][j]);
            }
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                u32x2 q4;
                if (XBF) {   // column j of the four token rows: halves j & 1 of words j >> 1, gathered with two byte permutes
                    const unsigned sel = (j & 1) ? 0x07060302u : 0x05040100u;
                    q4 = u32x2{__builtin_amdgcn_perm(bh[e][1][j >> 1], bh[e][0][j >> 1], sel),
                               __builtin_amdgcn_perm(bh[e][3][j >> 1], bh[e][2][j >> 1], sel)};
                } else {
                    q4 = pack4(bv[e][0][j], bv[e][1][j], bv[e][2][j], bv[e][3][j]);
                }
                *reinterpret_cast<u32x2*>(Q + (4 * (pc + 32 * e) + j) * kWideRowB + 8 * g) = q4;
            }
        }
    };
    if (nlo < nhi) {
        fetch(nlo);
        stage(0);
    }
    __syncthreads();
    int buf = 0;
    for (long n0 = nlo; n0 < nhi; n0 += kWideBK, buf ^= 1) {
        const bool more = n0 + kWideBK < nhi;
        const unsigned char* P = lds + buf * (kWideP + kWideQ);
        if (more) fetch(n0 + kWideBK);
        wide_mma(P, P + kWideP, acc, wr, wc, lane, [](int) {});
        if (more) stage(buf ^ 1);
        __syncthreads();
    }
    const int hh = lane >> 5;
    float* dst = part + (long)z * m * k;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int col = k0 + wc * 96 + u * 32 + (lane & 31);
        if (col >= k) continue;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * 64 + t * 32 + mfma_row(r, hh);
                if (row < m) dst[(long)row * k + col] = acc[t][u][r];
            }
    }
    if (colsum) {   // sixteen row groups per column -> one value (the operand tiles are dead: the loop ended on a barrier)
        float* red = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int j = 0; j < 4; ++j) red[g * kWideRows + 4 * pc + j] = cs[j];
        __syncthreads();
        if (tid < kWideRows && m0 + tid < m) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) v += red[q * kWideRows + tid];
            bpart[(long)z * m + m0 + tid] = v;
        }
    }
}

bool launch16_linear_wide(const LinearParams& p, hipStream_t s) {
    const auto al = [](const void* q) { return ((unsigned long long)q & 15) == 0; };
    bool ok = !p.wtrans && p.k % 64 == 0 && (p.lda & (p.a_bf16 ? 7 : 3)) == 0 && (p.ldw & 3) == 0 && al(p.a) && p.m > 128 && p.n >= 1024;
    if (p.seg_cols) ok = ok && p.seg_cols % kWideCols == 0 && al(p.w_seg[0]) && al(p.w_seg[1]) && al(p.w_seg[2]);
    else ok = ok && al(p.w);
    if (!ok) return false;
    const int nrt = (int)((p.n + kWideRows - 1) / kWideRows), ncg = (p.m + kWideCols - 1) / kWideCols;
    if (p.wpack && p.m % kWideCols == 0) {
        const dim3 grid((unsigned)(8 * ((nrt + 7) / 8) * ncg));
        if (p.a_bf16) hipLaunchKernelGGL(k16_linear_wdma<true>, grid, dim3(512), 0, s, p, nrt, ncg);
        else hipLaunchKernelGGL(k16_linear_wdma<false>, grid, dim3(512), 0, s, p, nrt, ncg);
        return true;
    }
    if (p.a_bf16) return false;   // only the streamed kernel reads bf16 rows (the caller checks eligibility first)
    hipLaunchKernelGGL(k16_linear_wide, dim3((unsigned)(8 * ((nrt + 7) / 8) * ncg)), dim3(512), 0, s, p, nrt, ncg);
    return true;
}

// Same contract as the k16_dw launch inside launch32_dw_seg (k_fp32_bwd.hip): fills part[nsplit][m][k] (and bpart[nsplit][m]).
// Returns the number of slices used, 0 if the shape is not eligible (nothing launched).
int launch16_dw_wide(const float* dy, int ldy, const float* x, int ldx, long n, int m, int k, float* part, size_t part_floats,
                     bool want_db, float** bpart_out, hipStream_t s, bool x_bf16, bool dy_bf16) {
    const bool fast = ((ldy | m | ldx | k) & 7) == 0 && (((unsigned long long)dy | (unsigned long long)x) & 15) == 0;
    if (!fast || n < 4096) return 0;
    const int mt = (m + kWideRows - 1) / kWideRows, kg = (k + kWideCols - 1) / kWideCols, nt = mt * kg;
    // one workgroup per CU (147 KB of LDS): as many n-slices as fill the chip ONCE -- rounded DOWN: with 86 slices x 3 tiles =
    // 258 workgroups on 256 CUs the last two wait for a free CU and the launch takes two rounds instead of one
    int nsplit = 256 / nt;
    const int cap = (int)((n + 255) / 256);
    if (nsplit > cap) nsplit = cap;
    if (nsplit < 1) nsplit = 1;
    while (nsplit > 1 && (size_t)nsplit * m * (k + 1) > part_floats) --nsplit;
    if ((size_t)nsplit * m * (k + 1) > part_floats) return 0;
    float* bpart = want_db ? part + (size_t)nsplit * m * k : nullptr;
    const dim3 grid((unsigned)(8 * ((nsplit + 7) / 8) * nt));
    if (x_bf16 && dy_bf16) hipLaunchKernelGGL((k16_dw_wide<true, true>), grid, dim3(512), 0, s, dy, ldy, x, ldx, n, m, k, nsplit, part, bpart);
    else if (x_bf16) hipLaunchKernelGGL((k16_dw_wide<true, false>), grid, dim3(512), 0, s, dy, ldy, x, ldx, n, m, k, nsplit, part, bpart);
    else if (dy_bf16) hipLaunchKernelGGL((k16_dw_wide<false, true>), grid, dim3(512), 0, s, dy, ldy, x, ldx, n, m, k, nsplit, part, bpart);
    else hipLaunchKernelGGL((k16_dw_wide<false, false>), grid, dim3(512), 0, s, dy, ldy, x, ldx, n, m, k, nsplit, part, bpart);
    *bpart_out = bpart;
    return nsplit;
}

}  // namespace mdg
