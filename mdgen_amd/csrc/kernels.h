// kernels.h -- parameter blocks and launchers shared between the kernels (*.hip) and api.cpp
#pragma once
#include "common.h"

namespace mdg {

struct QkvParams {
    const float* h;        // residual stream [N][384] fp32
    long nrows;            // N (SMALL layout: natural-order panels)
    AxisMap ax;            // attention axis
    ModMap mm;
    int shift_chunk, scale_chunk;
    const bf16x8 *wq, *wk, *wv;   // packed fragments [12 ftile][24 kstep][64 lane][8]
    const float *bq, *bk, *bv;    // permuted biases [384]
    const float* rope;     // [P][kRopeRow]: per half h: cos(6) | pad | sin(6) | pad
    unsigned char *qf, *kf, *vf;  // FLASH layout fragment buffers
    __bf16* qkv_small;     // SMALL layout [token][3][16 head][2 half][12]
    int panels_per_seq;
    uint32_t* vmask;       // FLASH layout: key-validity words [seq][vmask_stride], one per 32-key tile (see flash_vmask)
    int vmask_stride;
    // k_ln_qkv_attn4 only (residue axis, L == 4: attention inside the QKV kernel)
    const float *bias_k, *bias_v;   // learned bias key / value [384]
    MaskMap mk;                     // key-padding mask
    __bf16* obuf;                   // attention output [token][384] bf16 (PROJ == false)
    // k_ln_qkv_attn4<true>: the out-projection + gated residual of the same sub-layer, in the same kernel
    float* h_rw;                    // == h (the rows a workgroup normalised are the rows it updates)
    const bf16x8* wo;               // packed out-projection weights
    const float* bo;
    int gate_chunk;
};

struct ProjParams {
    float* h;
    long nrows;
    ModMap mm;
    int gate_chunk;
    int gated;
    const bf16x8* w;       // packed [12 ftile][K/16][64][8]
    const float* bias;
    const __bf16* a_bf16;  // MODE 0/1 A rows
    // MODE 2 (micro attention)
    const __bf16* qkv_small;
    AxisMap ax;
    MaskMap mk;
    const float *bias_k, *bias_v;
    const float* rope;
};

struct MlpParams {
    float* h;
    long nrows;
    ModMap mm;
    int shift_chunk, scale_chunk, gate_chunk;
    const bf16x8 *w1, *w2;  // w1 [48 ftile][24][64][8]; w2 [12 ftile][96][64][8]
    const float *b1, *b2;
    unsigned long long* trace;   // measurement only (mdgen_profile_phase_trace): [wave][32] s_memtime stamps, or null
    long trace_cap;              // capacity of `trace` in 64-bit words
    // k_mlp<PF1, true>: the preceding (temporal) attention sub-layer's out-projection + gated residual runs in the same panels
    // first (k_proj<0>'s work: attention output rows `o`, packed W_o, bias, gate chunk), o == null: off
    const __bf16* o;
    const bf16x8* wo;
    const float* bo;
    int gate_chunk_o;
    // k_mlp8<PRE, kMlpSplit> (part != null): the hidden chunks of a panel over kMlpSplit workgroups.  part: fp32 partials of the fc2
    // product [panel][kMlpSplit][64][384]; hupd: each workgroup's private copy of the panel's residual rows after the fused
    // out-projection (same shape; PRE only); counters: one per panel, zero before the first launch (the last arriver resets it)
    float* part;
    float* hupd;
    unsigned* counters;
};
constexpr int kMlpSplit = 3;            // workgroups per panel in the split form
constexpr int kMlpSplitMaxPanels = 96;  // panels the context's split scratch covers (the form is for launches of <= ncu / 3 panels)

// k_mlp_rows (k_rows.hip): the MLP block in row-owner form; `wstream` = both weight matrices as one fragment stream in
// consumption order (api.hip mlp_stream_table)
struct MlpRowsParams {
    float* h;
    long nrows;
    ModMap mm;
    int shift_chunk, scale_chunk, gate_chunk;
    const unsigned char* wstream;   // 2304 fragments of 1 KiB
    const float *b1, *b2;
    // fused out-projection of the preceding attention sub-layer (o != null): attention output rows, W_o as 288 fragments in
    // consumption order (k-step major, natural k), its bias and gate chunk
    const __bf16* o;
    const unsigned char* wo_stream;
    const float* bo;
    int gate_chunk_o;
    unsigned long long* trace;      // measurement only: [wave][8] s_memtime stamps, or null
    long trace_cap;
    // gate fold (non-null; o == null, every row of the launch shares the modulation row): `wstream` is the (step, layer) stream with
    // the gate folded into fc2 and b2g = gate * b2 (launch_pack_fold): accumulators start from the residual rows, store-only epilogue
    const float* b2g;
    // tail (with b2g, the trunk's LAST layer): FinalLayer + Euler update (layers.py:57-74, integrators.py:106) run on the updated rows while
    // they are still in registers, and the rows are NOT stored (nothing reads the residual stream after the last layer): k_final's
    // launch, its 98 MB read and this kernel's 98 MB write are gone.  tail_w: emb_to_latent.linear.weight as 24 fragments in kappa
    // order (rows >= D zero), tail_b [32], tail_mod: the step's final adaLN row (shift chunk 0, scale chunk 1).
    const bf16x8* tail_w;
    const float* tail_b;
    const float* tail_mod;
    int tail_D, tail_euler;
    float tail_dt;
    float* tail_x;      // euler: state updated in place
    float* tail_out;    // !euler: velocity
    // ... and (emb_base != null; euler) the NEXT step's token embedding from the updated state (k_embed's work) written to h: see rows.h
    // rows_embed_tail.  emb_wl / emb_wc: launch_pack_embed_rows; emb_base: the next step's base rows of this view (launch_embed_base)
    const float *emb_wl, *emb_wc, *emb_base, *emb_mdelta, *emb_xcond;
    const bf16x8 *emb_wl_hi, *emb_wl_lo, *emb_wc_hi, *emb_wc_lo;   // the weights as bf16 pairs (launch_pack_rows kappa = 1, part 0 / 1); null: exact fp32 form
    const int64_t* emb_cmask;
    int emb_T, emb_L;
};
// latent_to_emb / cond_to_emb weight [384][D] -> [12 ft][NK4][64 lanes][4]: value (ft, q, lane, i) = W[32 ft + (lane & 31)][8 q + 4 (lane >> 5) + i]
// (0 past D), NK4 = 3 (D <= 24) or 4: the A operands of rows_embed_gemm
void launch_pack_embed_rows(const float* w, int D, float* pack, hipStream_t s);
void launch_sub_f32(const float* a, const float* b, float* dst, int n, hipStream_t s);   // dst = a - b
constexpr int kEmbRowsFloats = 12 * 4 * 64 * 4;
// base[s][bl][c] = bl[c] + bc[c] + mask_emb[0][c] + (pos_embed ? pos_embed[l][c] : 0) + ipa_out[s][bl][c]   (S * BL rows)
void launch_embed_base(const float* bl, const float* bc, const float* mask_emb, const float* pos_embed, const float* ipa_out, int S,
                       int BL, int L, float* base, hipStream_t s);

struct LnLinearParams {
    const float* h;
    long nrows;
    ModMap mm;              // mm.mod = [gamma | beta]
    const bf16x8* w;        // packed [nout/32][24][64][8]
    const float* bias;
    float* out;             // [nrows][nout] fp32
    int nout;               // multiple of 96
};

struct FinalParams {
    const float* h;
    long nrows;
    ModMap mm;
    int shift_chunk, scale_chunk;
    const bf16x8* w;        // packed [1 ftile][24][64][8] (rows >= D are zero)
    const float* bias;      // [32]
    int D;
    int euler;
    float dt;
    float* x;               // euler: state updated in place
    float* out;             // !euler: velocity
};

// Key-validity words of the tiled attention: one uint32 per (sequence, 32-key tile); bit set = the key is real (inside
// the sequence and not padded, or the learned bias key at position len).  Per sequence `stride` words, a multiple of
// 64 >= ntile + 1, zero beyond the last tile (k_flash reads them in 64-tile windows, one tile ahead).  They live in the
// slack of the V^T fragment region (a tile is allocated kFragBytes, V^T uses kFragV of it), behind the last fragment.
static_assert((kFragBytes - kFragV) * kH >= 4 * 64 + 4 * 1 + 256,
              "the key-validity words (<= ntile + 64 of them per sequence, 256-byte aligned) must fit the slack the V^T fragments leave");
__host__ __device__ inline int flash_vmask_stride(int ntile) { return (ntile + 64) & ~63; }
__host__ __device__ inline size_t flash_vmask_offset(long nseq, int ntile) {
    return (((size_t)nseq * kH * ntile * kFragV) + 255) & ~(size_t)255;
}

struct FlashParams {
    AxisMap ax;
    MaskMap mk;
    const unsigned char *qf, *kf, *vf;
    const float *bias_k, *bias_v;  // natural fp32 [384]
    const float* rope;
    __bf16* obuf;           // [N][384]
    const uint32_t* vmask;  // [seq][vmask_stride]: bit k of word t = key 32 t + k may be attended (written by k_ln_qkv)
    int vmask_stride;
    int force_robust;       // option attention_path: 1 = skip the fixed-anchor loop, always run the moving-shift loop
    int rotate;             // option flash_rotate: 1 = every 64-query chunk starts its walk over the key tiles at a different tile
};

// k_flash_proj: k_flash for all 16 heads of 64 queries + the sub-layer's out-projection + gated residual (k_proj<0>'s work)
struct FlashProjParams {
    FlashParams f;          // (f.obuf is not used: the attention output stays in LDS)
    float* h;               // residual stream, updated in place
    ModMap mm;
    int gate_chunk;
    const bf16x8* wo;       // packed out-projection weights [12 ftile][24 kstep][64 lane][8]
    const float* bo;
};

struct EmbedParams {
    const float *x, *x_cond;
    const int64_t* x_cond_mask;
    const float *wl, *bl, *wc, *bc;  // latent_to_emb / cond_to_emb fp32 [C][D], [C]
    const float *wl_pack, *wc_pack;  // the two weights in the kernel's B-operand order (launch_pack_embed): coalesced prologue loads
    const float* mask_emb;           // [2][C]
    const float* pos_embed;          // [crop][C] or nullptr
    const float* ipa_out;            // [B*L][C] for this step
    float* h;
    long N;
    int T, L, D;
};

struct IpaAttnParams {
    const float* proj;      // [M][672]
    const float *rot, *trans;  // [B][L][3][3], [B][L][3]
    const float* mask_bl;   // [B][L]
    const float* head_w;    // [4]
    __bf16* feat;           // [M][256]
    float* feat32;          // fp32 mode: features written here (fp32) instead of `feat`
    float* stats;           // training tape (nullable): [M][4 heads] log-sum-exp of the logits (m + log(sum exp))
    int ngroups, B, L;
    // k_ipa_attn_tiled with few groups (training at B = 1: four workgroups, one wave per SIMD, 185 us): the key loop cut
    // into `nsplit` slices (blockIdx.y) that leave (running max, denominator, o, o_pt) per (slice, token, head) in `part`
    // ([nsplit][M][4][58] floats) for k_ipa_attn_merge.  Set by launch_ipa_attn when `part` is given; 0 / null: one slice.
    int nsplit;
    float* part;
    size_t part_floats;
};
constexpr int kIpaFwdRec = 58;   // m | den | o(32) | o_pt(24)

struct FloatChunk {
    float v[128];
    int n;
};

int panel_waves_for(long grid, int forced, int ncu);   // 4 or 8 waves per 64-row panel for a launch of `grid` panels
void launch_ln_qkv(const QkvParams& p, bool small, hipStream_t s, bool pre = false, int waves = 4, bool split = false, bool half = false);
void launch_xcc_probe(int* out, int nblocks, hipStream_t s);
void launch_ln_qkv_attn4(const QkvParams& p, bool fuse_proj, hipStream_t s, bool half = false);   // half: 32-row workgroups (fuse_proj only)
void launch_proj(const ProjParams& p, int mode, hipStream_t s);
void launch_mlp(const MlpParams& p, hipStream_t s, int waves = 4);
void launch_mlp_rows(const MlpRowsParams& p, int nw, hipStream_t s);
constexpr int kMlpStreamFrags = 2304;   // 1 KiB fragments of one MLP weight stream (api.hip mlp_stream_table)
// per-(step, layer) MLP streams with the step's gate folded into fc2 (+ b2g = gate * b2); S * nl streams, nl <= 8
void launch_pack_fold(const float* mod, long mod_step_stride, int S, int nl, const int* goff, const float* const* w2,
                      const float* const* b2, const bf16x8* const* base, const int* tab, bf16x8* dst, float* b2g, hipStream_t s);
// rowmap (nullable): source row of packed row r (a permutation of the matrix's rows); kappa: K order inside a k-step -- 0
// natural, 1 rows.h kappa (operand = LayerNorm / GELU registers)
void launch_pack_stream(const float* w, int ld, int which, const int* tab, int nfrag, float scale, int kappa, bf16x8* dst,
                        hipStream_t s, const int* rowmap = nullptr);
void launch_ln_linear(const LnLinearParams& p, hipStream_t s);
void launch_final(const FinalParams& p, hipStream_t s);
void launch_flash(const FlashParams& p, hipStream_t s);
long flash_proj_jobs(const AxisMap& ax);
void launch_flash_proj(const FlashProjParams& p, int form, hipStream_t s);   // form 4: k_flash_proj (64-row panels), 8: k_flash_proj8 (128-row)
void launch_pack_embed(const float* w, int D, float* pack, hipStream_t s);   // pack: kEmbPackFloats floats
constexpr int kEmbPackFloats = 4 * 3 * 14 * 64;
void launch_embed(const EmbedParams& p, hipStream_t s);
void launch_path_plan(const float* t, const float* x0, const float* x1, float* xt, float* ut, long per_sample, long B,
                      int gvp, hipStream_t s);
void launch_masked_mse(const float* pred, const float* target, const float* mask, float* loss, long per_sample, long B,
                       hipStream_t s, float* scratch = nullptr, size_t scratch_floats = 0, float* den_out = nullptr);
void launch_ipa_attn(const IpaAttnParams& p, hipStream_t s);

// fp32-operand path (k_fp32.hip)
extern thread_local int g_k16_attn_form;       // k_attn16.hip: 1 = sequence-resident attention backward for axes of 129 .. 256 positions
extern thread_local int g_k32_bf16_operands;   // 1: k32_linear / k32_dw multiply bf16-rounded operands (training option train_precision = 16)
extern thread_local const char* g_k32_launch_error;   // set by a launcher that refused a shape (nothing launched)
const char* k32_take_launch_error();                  // ... and cleared by the entry point that reports it
void launch32_ln_mod(const float* x, long nrows, const ModMap& mm, int shift_chunk, int scale_chunk, int affine, float eps,
                     float* y, hipStream_t s, float* keep = nullptr, bool y_bf16 = false);   // keep: copy of x (training tape); y_bf16: y holds bf16 rows
void launch32_linear(const float* a, int lda, const float* w, int ldw, const float* bias, long n, int m, int k, int mode,
                     float* c, int ldc, int col0, const ModMap& mm, int gate_chunk, int gated, float scalar, hipStream_t s,
                     int wtrans = 0, float* c2 = nullptr,
                     const void* wpack = nullptr,    // launch16_pack_wstream's bf16 fragment stream of w (k_wide16.hip)
                     int flags = 0);                 // 1: a is bf16 rows (needs wpack), 2: c2 (mode 6) is written as bf16
// backward kernels of the training step (k_fp32_bwd.hip)
// db != nullptr: the bias gradient db[m] += column sums of dY may be computed by the same pass (returns true if it was;
// otherwise the caller runs launch32_colsum)
bool launch32_dw_seg(const float* dy, int ldy, const float* x, int ldx, long n, int mseg, int nseg, int k, float* const* dw,
                     float* const* db, float* part, size_t part_floats, hipStream_t s,
                     bool x_bf16 = false, bool dy_bf16 = false);   // nseg layers sharing x, dY side by side; *_bf16: stored as bf16 rows (wide kernel only)
bool launch32_dw(const float* dy, int ldy, const float* x, int ldx, long n, int m, int k, float* dw, float* part,
                 size_t part_floats, hipStream_t s, float* db = nullptr, bool x_bf16 = false,
                 bool dy_bf16 = false);   // x_bf16 / dy_bf16: x / dy are bf16 rows (wide kernel only)
void launch32_colsum(const float* a, int lda, const float* b, int ldb, const float* roww, int mode, long nrows, int ncols,
                     long tokens_per_group, float eps, float* out, long ldo, float* part, size_t part_floats, hipStream_t s);
void launch32_ln_bwd(const float* x, const float* dy, long nrows, const ModMap& mm, int scale_chunk, int affine, float eps,
                     float* dx, int accumulate, hipStream_t s);
void launch32_gate_mul(const float* a, long nrows, const ModMap& mm, int gate_chunk, int gated, float* out, hipStream_t s);
void launch32_attn_bwd(const float* qkv, int ld, const AxisMap& ax, const MaskMap& mk, const float* bias_k,
                       const float* bias_v, const float* inv_freq, const float* o, const float* dout, float* dqkv,
                       float* stats, float* dbias, hipStream_t s, const float* lse_in = nullptr);
void launch32_rope_bwd(float* buf, long ntok, int ld, long pos_div, int pos_mod, const float* inv_freq, float qscale,
                       hipStream_t s);
void launch32_loss_grad(const float* pred, const float* target, const float* mask, long per_sample, long B, float* den,
                        float* dpred, hipStream_t s, float* scratch = nullptr, size_t scratch_floats = 0);
void launch32_sum_frames(const float* a, int B, int T, int L, float* out, hipStream_t s);
void launch32_ipa_bwd(const IpaAttnParams& f, const float* dfeat, float* dproj, float* dhw, float* qrec, float* dheadw,
                      hipStream_t s, float* part = nullptr, size_t part_floats = 0);   // part: scratch for the sliced form
void launch32_gated_add(float* h, const float* u, long nrows, const ModMap& mm, int gate_chunk, int gated, hipStream_t s);
void launch32_gated_sum(float* h, const float* x, const float* u, long nrows, const ModMap& mm, int gate_chunk, hipStream_t s);
void launch32_gate_ln_mod(const float* xp, const float* up, long nrows, const ModMap& gm, int gate_chunk, const ModMap& mm,
                          int shift_chunk, int scale_chunk, float eps, float* y, float* keep, hipStream_t s, bool y_bf16 = false);
void launch32_indicator(const int64_t* cm, long n, float* ind0, float* ind1, hipStream_t s);
void launch32_embed_rows_bwd(const float* dx0, const int64_t* aatype, int ngroups, int B, int L, float* dw, hipStream_t s);
void launch32_temb_bwd(const float* t_rows, int nrows, float tmul, const float* w0, const float* b0, const float* w2,
                       const float* b2, const float* dst, float* emb, float* h1, float* dpre1, float* dpre2, hipStream_t s);
void launch32_rope(float* buf, long ntok, int ld, long pos_div, int pos_mod, const float* inv_freq, hipStream_t s);
// out[nb][m] = x[nb][K] W[K][m], nb small and K long (split over K); false: partial buffer too small, nothing launched
bool launch32_skinny_wt(const float* x, int ldx, const float* W, int ldw, int nb, int m, long K, float* out, float* part,
                        size_t part_floats, hipStream_t s);
// fused element-wise backward + adaLN column sums (k_fp32_bwd.hip); false: partial buffer too small, nothing launched
bool launch32_ln_bwd_sums(const float* x, const float* dy, long nrows, const ModMap& mm, int scale_chunk, float eps, float* dx,
                          int accumulate, long tokens_per_group, float* out, long ldo, float* part, size_t part_floats, hipStream_t s);
bool launch32_gate_bwd_sums(const float* dh, const float* u, long nrows, const ModMap& mm, int gate_chunk, float* du,
                            long tokens_per_group, float* out, long ldo, float* part, size_t part_floats, hipStream_t s,
                            bool du_bf16 = false);   // du_bf16: du is written as bf16 rows
void launch32_transpose(const float* src, int rows, int cols, float* dst, hipStream_t s, int ldd = 0);   // dst[c][r] (ld ldd, default rows) = src[r][c]
bool launch16_linear_seg3(const float* a, int lda, const float* const* w, int ldw, const float* const* bias, const float* scale,
                          long n, int mseg, int k, float* c, int ldc, int col0, hipStream_t s, const void* wpack = nullptr,
                          bool a_bf16 = false);   // a_bf16: `a` holds bf16 rows (lda in elements; the streamed kernel only)
// bf16 fragment stream of a weight for k16_linear_wdma: W(col, kk), col < m (a multiple of 384), kk < k (a multiple of 64),
// from nsrc <= 3 fp32 matrices of row stride ld: turned == 0: src[col / seg] is [seg][k] (layers side by side along the
// columns); turned == 1: src[kk / seg] is [seg][m] (dX = dY W: the contraction runs over the weights' rows).
// false: shape not eligible for the streamed kernel (nothing launched).  out: m * k * 2 bytes.
bool launch16_pack_wstream(const float* const* src, int nsrc, int seg, int ld, long n, int m, int k, int turned, void* out, hipStream_t s);
// bf16-operand (MFMA) attention of the training step, k_attn16.hip: same arguments as launch32_attn / launch32_attn_bwd; the
// backward needs the forward's log-sum-exp tape (lse_in != nullptr).
// rope_inside (only where attn16_seq_form(ax): the sequence-resident kernels): q, k of `qkv` are NOT rotated yet -- the kernels
// rotate them while they convert them (no k32_rope pass); forward and backward of a sub-layer must agree on it
bool attn16_seq_form(const AxisMap& ax);
void launch16_attn(const float* qkv, int ld, const AxisMap& ax, const MaskMap& mk, const float* bias_k, const float* bias_v,
                   const float* inv_freq, float* out, hipStream_t s, float* lse_out = nullptr, bool rope_inside = false);
void launch16_attn_bwd(const float* qkv, int ld, const AxisMap& ax, const MaskMap& mk, const float* bias_k,
                       const float* bias_v, const float* inv_freq, const float* o, const float* dout, float* dqkv,
                       float* stats, float* dbias, hipStream_t s, const float* lse_in, bool rope_inside = false,
                       bool out_bf16 = false);   // out_bf16 (sequence-resident form only): dqkv is written as bf16 rows (ld in elements)
void launch32_attn(const float* qkv, int ld, const AxisMap& ax, const MaskMap& mk, const float* bias_k, const float* bias_v,
                   const float* inv_freq, float* out, hipStream_t s, float* lse_out = nullptr);

// optimiser (k_optim.hip)
void launch_sumsq(const float* g, long n, float scale, float* partial, int nblocks, float* out, hipStream_t s);
void launch_adam(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int adamw, float bc1, float bc2_sqrt, float grad_scale, const float* sumsq,
                 float max_norm, hipStream_t s);
void launch_ema(float* ema, const float* p, long n, float one_minus_decay, hipStream_t s);

// small kernels (k_small.hip)
void launch_temb(const float* t_rows, int nrows, float tmul, const float* w0, const float* b0, const float* w2,
                 const float* b2, float* silu_out, hipStream_t s);
void launch_adaln(const float* st, int nrows, const float* w, const float* b, int nout, float* mod, hipStream_t s);
void launch_rope_table(float* rope, const float* inv_freq, int npos, hipStream_t s);
void launch_gather_f32(const float* src, const int* idx, float scale, float* dst, int n, hipStream_t s);
// part: 0 the weight rounded to bf16, 1 the bf16 of its rounding error (w - float(bf16(w))): the lo half of a bf16 pair
void launch_pack_rows(const float* w, int ld, const int* rowmap, int nft, int ksteps, float scale, bf16x8* dst,
                      hipStream_t s, int kappa = 0, int part = 0);
void launch_ipa_init(const float* aa_emb, const int64_t* aatype, const float* rel7, const float* w7, const float* b7,
                     float* h, int ngroups, int B, int L, hipStream_t s);
void launch_add_inplace(float* dst, const float* src, long n, hipStream_t s);
void launch_write_floats(const float* host_vals, int n, float* dst, hipStream_t s);
void launch_rel7(const float* r1, const float* t1, const float* r2, const float* t2, float* out7, long n, hipStream_t s);

// SE(3) / pre / post (k_se3.hip)
void launch_rigid_compose(long n, const float* r1, const float* t1, const float* r2, const float* t2, float* ro,
                          float* to, hipStream_t s);
void launch_rigid_invert(long n, const float* r, const float* t, float* ro, float* to, hipStream_t s);
void launch_rigid_apply(long n, long ppf, const float* r, const float* t, const float* pts, float* out, int inverse,
                        hipStream_t s);
void launch_quat_to_rot(long n, const float* q, int normalize, float* rot, hipStream_t s);
void launch_from_3_points(long n, const float* pnx, const float* org, const float* pxy, float* rot, float* trans,
                          hipStream_t s);
void launch_rot_to_quat(long n, const float* rot, float* q, hipStream_t s);
void launch_prep_latents(int B, int T, int L, int tps, int bcast, int cond_interval, const float* rots, const float* trans,
                         const float* tors, float* latents, float* x_cond, int64_t* x_cond_mask, hipStream_t s);
void launch_samples_to_atom14(int B, int T, int L, int D, int tps, const float* samples, const float* rot0,
                              const float* trans0, const int64_t* seqres, const float* default_frames,
                              const float* lit_positions, const int64_t* atom14_group, const float* atom14_mask,
                              float* atom14, int out_T, int out_t0, hipStream_t s);
void launch_atom14_to_cond(int B, int L, const float* atom14, long in_bstride, const int64_t* seqres,
                           const int64_t* a37to14, const float* a37mask, const int64_t* chi_idx, const float* chi_mask,
                           float* rots, float* trans, float* tors, float* tmask, hipStream_t s);

}  // namespace mdg
