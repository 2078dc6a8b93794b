/*
 * mdgen_amd.h -- C-ABI of the MI355X-native MDGen denoising sampler (libmdgen_amd.so).
 *
 * The reference (bjing2016/mdgen) is pure Python/PyTorch and has no FFI of its own; its boundary
 * for this path is the Python call surface.  Each entry point below
 * names the reference code it replaces (paths relative to the reference repo).  The Python host
 * (`mdgen_amd/`) keeps the reference's signatures and calls these through ctypes.
 *
 * Conventions
 *   - plain C, no C++/torch types.  All tensor arguments are DEVICE pointers (row-major,
 *     contiguous, layouts exactly as the reference's tensors) unless the name ends in `_host`.
 *   - the caller owns every input/output/workspace buffer; the library owns only the packed
 *     weights and cached hipGraph handles inside the opaque context.
 *   - every launch goes on the caller's `stream` (a hipStream_t passed as void*) and nothing inside a
 *     call synchronises the device (graph-capturable).  One exception, documented: `mdgen_sample_euler`
 *     forks the second half of the batch onto one context-owned stream and joins it back onto `stream`
 *     before returning (event fork/join, no host wait; option "streams" = 1 disables it).
 *   - return 0 on success, negative = invalid argument / state, positive = hipError_t.
 *     `mdgen_last_error()` returns a thread-local message.  No exceptions cross the ABI.
 *   - a context is not thread-safe; use one per device per process.
 *   - compiled for gfx950 only; C = 384, heads = 16, head_dim = 24, ffn = 1536 are compile-time.
 */
#ifndef MDGEN_AMD_H
#define MDGEN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a public struct or signature changes; mdgen_amd/_lib.py refuses a library whose version differs. */
#define MDGEN_ABI_VERSION 7   /* 6: mdgen_ws_layout.split; 7: mdgen_ws_layout.fold */

typedef struct mdgen_ctx mdgen_ctx;

/* 0 for a product library; 1 if the .so was built with a kernel experiment switch (csrc/dev.h, scripts/micro/): timing-only
 * builds whose results may be wrong.  mdgen_amd/_lib.py loads such a library only when MDGEN_AMD_LIB names it explicitly. */
int32_t mdgen_dev_build(void);

/* Model hyper-parameters (reference argparse flags, mdgen/parsing.py:79-120). */
typedef struct mdgen_model_desc {
    int32_t embed_dim;      /* must be 384 */
    int32_t mha_heads;      /* must be 16  */
    int32_t num_layers;     /* trunk layers == IPA layers (reference default 5), 1..8 */
    int32_t latent_dim;     /* 21 (forward-sim) or 28 (TPS); wrapper.py:196 */
    int32_t ipa_heads;      /* must be 4  */
    int32_t ipa_head_dim;   /* must be 32 */
    int32_t ipa_qk;         /* must be 8  */
    int32_t ipa_v;          /* must be 8  */
    int32_t abs_pos_emb;    /* 1: add pos_embed[1,crop,C] (latent_model.py:234-235) */
    int32_t crop;           /* rows of pos_embed (>= L when abs_pos_emb) */
    int32_t tps_condition;  /* 1: two-stream IPA with relative-frame inputs (latent_model.py:193-207) */
    float   time_multiplier;/* latent_model.py:243 (100) */
} mdgen_model_desc;

/* Problem shape of one call: x is (B,T,L,D). */
typedef struct mdgen_shape {
    int32_t B, T, L;
} mdgen_shape;

/* Byte offsets into the caller-provided workspace (for tests / debugging / profiling). */
typedef struct mdgen_ws_layout {
    size_t total_bytes;
    size_t h;          /* fp32 residual stream        [N][384]                         */
    size_t qf, kf, vf; /* bf16 attention operand fragments (DESIGN.md "fragment layout") */
    size_t obuf;       /* bf16 attention output       [N][384]                         */
    size_t mod;        /* fp32 adaLN table            [R][77*384]                      */
    size_t silu_t;     /* fp32 SiLU(t_embedder(t))    [R][384]                         */
    size_t ipa_out;    /* fp32 IPA-stack table        [S][B*L][384]                    */
    size_t h_ipa;      /* fp32 IPA-stack residual     [S*B*L][384] (x2 when TPS)       */
    size_t ipa_proj;   /* fp32 IPA projections        [S*B*L][672]                     */
    size_t ipa_feat;   /* bf16 IPA concat features    [S*B*L][256]                     */
    size_t mask_bl;    /* fp32 compact mask[:,0]      [B][L]                           */
    size_t rel7;       /* fp32 TPS relative frames    [2][B][L][7]                     */
    size_t tgrid;      /* fp32 per-step times         [S][B]                           */
    size_t f32_scratch;/* fp32 path: LN out | q,k,v | attention out | MLP hidden | IPA features (0 bytes unless the
                          context keeps fp32 weights)                                            */
    size_t split;      /* scratch of the split-panel MLP kernel (option "small_split"): arrival counters (1 KiB) | fp32 partials |
                          private residual rows, [panels <= 96][3][64][384] each                 */
    size_t fold;       /* option "mlp_fold": per (step, trunk layer) the MLP weight stream with that step's gate folded into fc2
                          [S][layers][2304 KiB bf16 fragments] | gate * fc2.bias [S][layers][384] fp32 (0 bytes unless t_shared and
                          the trunk's MLP launches take the row-owner kernel)                    */
    size_t embase;     /* option "mlp_tail": the x-independent part of the token embedding per (step, b, l), [S][B*L][384] fp32, for the
                          steps whose embedding is computed by the previous step's last MLP launch (0 bytes unless `fold` and S > 1) */
} mdgen_ws_layout;

const char* mdgen_last_error(void);
int32_t     mdgen_abi_version(void);

/* ---- context / weights -------------------------------------------------------------------
 * Replaces `NewMDGenWrapper.load_from_checkpoint(...).eval().to('cuda')` (sim_inference.py:129-130)
 * for the `model.*` sub-tree of the Lightning state_dict.  Weights are handed over one tensor
 * at a time under the reference's own state_dict key (e.g. "layers.0.mha_t.attn.q_proj.weight");
 * `data` is an fp32 DEVICE pointer; the library re-packs into its MFMA fragment formats on
 * `stream`.  `mdgen_ctx_finalize` fails if any required key was not provided. */
int32_t mdgen_ctx_create(mdgen_ctx** out, const mdgen_model_desc* desc);
int32_t mdgen_ctx_destroy(mdgen_ctx* ctx);
int32_t mdgen_ctx_set_weight(mdgen_ctx* ctx, const char* key, const float* data,
                             const int64_t* shape, int32_t ndim, void* stream);
int32_t mdgen_ctx_finalize(mdgen_ctx* ctx, void* stream);
/* Run-time options (no environment variables are read by the library):
 *   "streams"          1..8, default 2: contiguous sub-batches of an Euler rollout run on this many concurrent
 *                      streams (the caller's + context-owned ones, fork/join by events inside the call).  While the
 *                      option has not been set, the count is also limited to pieces of >= 256 64-row panels (a piece that
 *                      does not fill the chip by itself gains nothing from running beside another one);
 *   "keep_fp32_weights" 0/1, default 0; set to 1 BEFORE handing weights over to keep an fp32 copy of each (137 MB);
 *   "precision"        16 (default): bf16 MFMA operands, fp32 accumulate / softmax / LayerNorm / residual stream
 *                      (rel-L2 4-6e-3 per network evaluation against the fp32 reference);
 *                      32: fp32 operands everywhere (v_mfma_f32_32x32x2_f32, csrc/k_fp32.hip), the reference's own
 *                      arithmetic -- a tolerance mode ~10x slower; needs keep_fp32_weights;
 *   "residue_l4_path"  residue-axis attention sub-layer when L == 4: 2 (default) one kernel for the whole
 *                      sub-layer, 1 attention inside the QKV kernel + separate out-projection, 0 the general
 *                      L <= 8 path.  All three compute mha.py:258-397 + latent_model.py:457-462.
 *   "attention_path"   tiled attention (sequence > 8 positions; mha.py:359-396): 0 (default) softmax with a FIXED
 *                      shift anchored on the first key tile, an overflow test per query row at the end and an
 *                      automatic re-run of the affected (head, 64 queries) on the robust loop; 1 the robust loop
 *                      (running row max, shift re-anchored as it grows) for everything.  Same results to fp32
 *                      rounding; 1 is ~1.5x slower.
 *   "mlp_path"         the MLP block (latent_model.py:478-481, layers.py:77-84): 0 the 64-row resident-panel kernel
 *                      (csrc/k_gemm.hip k_mlp); 1 (default) the row-owner kernel (csrc/k_rows.hip k_mlp_rows: a wave owns 32
 *                      rows, activations in registers, weights as one LDS-DMA stream) for launches of >= 768 row tiles,
 *                      which fill the chip, and the panel kernel below that; 2 the row-owner kernel always.
 *   "mlp_fold"         1 (default) / 0: calls whose batch shares t (mdgen_sample_euler / mdgen_rollout_euler; integrators.py:99) have ONE
 *                      MLP gate vector per (step, layer): it is folded into the weights once per call -- W2' = diag(gate) W2 rounded to
 *                      bf16 once from the fp32 weight, b2' = gate * b2 -- so that h + gate * (W2 u + b2) (latent_model.py:481) becomes
 *                      "accumulators start at h + b2', accumulate W2' u, store": the row-owner kernel reads the residual rows once
 *                      instead of twice.  Same values to the bf16 rounding of gate * w instead of w.  Workspace: mdgen_ws_layout.fold
 *                      (S x layers x 2.36 MB); the report tags such launches "mlp@fold".
 *   "mlp_tail"         2 (default) / 1 / 0: with mlp_fold, the FinalLayer (layers.py:57-74: LN + modulate, Linear C -> D) and the Euler update of x
 *                      (integrators.py:106) run inside the LAST trunk layer's folded MLP kernel, on the updated rows while they are in
 *                      registers; that kernel then does not store its rows (nothing reads the residual stream after the last layer) and
 *                      there is no k_final launch: 196 MB less HBM traffic per network evaluation.  Not with trace_h.  Same values to
 *                      fp32 summation order; the report tags the launch "mlp@fold+final".  In a rollout the same launch goes on to
 *                      compute the NEXT step's token embedding (latent_model.py:233-246) from the state it has just updated and
 *                      writes it as that step's residual stream ("mlp@fold+final+embed"): steps 1 .. S-1 launch no k_embed (value 2;
 *                      1: the FinalLayer + Euler update only; workspace: mdgen_ws_layout.embase).
 *   "embed_split"      1 (default) / 0: the products of that embedding tail (W_l x, W_c x_cond: K = 21 / 28) on the bf16 MFMA with each operand
 *                      split into a bf16 pair hi + lo (16 mantissa bits per side, the lo x lo term dropped: 2^-17 of a product)
 *                      instead of v_mfma_f32_32x32x2_f32, which runs at a quarter of its nominal rate on gfx950.  0: the exact fp32 form.
 *   "fuse_proj"        the temporal attention's out-projection + gated residual (mha.py:397, latent_model.py:476) inside the
 *                      MLP kernel, ahead of the MLP: 0 off / 1 inside the row-owner kernel / 2 as a prologue phase of the
 *                      64-row panel kernel (k_mlp<3, true>; selects the panel kernel) / 3 (default) as 2 where the launch
 *                      takes the panel kernel anyway (fewer than 768 row tiles: one launch less per layer, +2 %).
 *   "fuse_proj_qkv"    1 (default) / 0: residue axis on the tiled-attention path (L > 8): its out-projection + gated residual
 *                      (mha.py:397, latent_model.py:462) runs inside the temporal sub-layer's LN -> q, k, v kernel, whose panels
 *                      then normalise rows that are still in L2 (k_ln_qkv<false, true>; one launch and one HBM read of the
 *                      residual stream less per layer).
 *   "flash_proj"       tiled attention (sequence > 8 positions) and the sub-layer's out-projection + gated residual (mha.py:359-397,
 *                      latent_model.py:462,476) in ONE launch (csrc/k_flash.hip k_flash_proj: a workgroup owns 64 queries of a
 *                      sequence for all 16 heads, the attention output stays in LDS as the projection's A operand): 1 (default)
 *                      for launches of >= 512 such workgroups (cfg-2, ATLAS), 0 never (k_flash, then k_proj<0> or a projection
 *                      deferred into the next kernel per fuse_proj / fuse_proj_qkv), 2 always.  Same values as the separate
 *                      kernels (same operands, same summation order).
 *   "flash_rotate"     1 (default) / 0: tiled attention, fixed-anchor loop: the 64-query chunks of a sequence walk its key tiles from
 *                      different starting tiles (chunk c of n starts at tile c * tiles / n and wraps): chunks that start together then
 *                      miss on different fragments instead of queueing behind one chain of HBM misses.  A sum over keys: same
 *                      values to fp32 rounding.
 *   "flash_proj_form"  0 (default) / 4 / 8: which fused kernel: k_flash_proj (four waves, a 64-query panel, two workgroups per CU) or
 *                      k_flash_proj8 (eight waves, a 128-query panel, four query tiles per wave, one workgroup per CU).  0: the
 *                      128-query form for sequences of >= 512 positions whose launch gives every CU a workgroup, else the 64-query
 *                      form.  Same values up to the order in which a query's keys are summed (fp32 rounding).
 *   "panel_waves"      0 (default) / 4 / 8: the 64-row panel kernels that exist in a four- and an eight-wave form (k_mlp / k_mlp8,
 *                      k_ln_qkv<false> / k_ln_qkv8) take the eight-wave form for launches of at most one workgroup per CU; 4 / 8
 *                      force one form whatever the launch size (tests, A/B runs).  mdgen_profile_report tags the class of such
 *                      a launch with "@p4" / "@p8".
 *   "small_split"      1 (default) / 0: launches far below one workgroup per CU (B = 1: sim_inference.py runs one trajectory; the IPA
 *                      stack) give a 64-row panel to several workgroups: the MLP block's twelve hidden chunks go to three workgroups
 *                      on one XCD (k_mlp8<., 3>, launches of <= CUs / 3 panels; fp32 partials of the second product meet in L2, the
 *                      last arriver adds them in a fixed order and runs the gated residual epilogue -- bit-reproducible, nobody
 *                      waits) and q, k | v of the temporal LN -> q, k, v kernel to two (k_ln_qkv8<true>, <= CUs / 2 panels).
 *                      Scratch: mdgen_ws_layout.split.  Same values to fp32 rounding (a three-term instead of a two-term sum);
 *                      the report tags such launches "@p8x3" / "@p8x2".  Off while a call runs sub-batch streams.
 *                      Round 6: ... and give the L = 4 residue sub-layer kernel and the split q, k | v kernel 32-ROW workgroups where
 *                      even those fit one per CU (k_ln_qkv_attn4<true, true>: B <= 2 at T 1000, the IPA stack; k_ln_qkv8<true, true>:
 *                      B = 1 at T 1000): twice the workgroups on twice the CUs, half the rows per SIMD -- a small launch lasts as long
 *                      as one wave's chain of work.  Same arithmetic per row; tags "@h32" / "@h32x2"; no scratch, so these two stay
 *                      on while a call runs sub-batch streams.
 *   "train_precision"  operands of the matrix products of mdgen_train_forward_backward (linear layers, weight gradients,
 *                      the attention's q k^T / p v and their backward): 32 (default) fp32, the exact mode; 16 rounded to
 *                      bf16 on the MFMA, fp32 accumulation, fp32 master weights and activations (train.py:13
 *                      set_float32_matmul_precision('medium')); softmax, LayerNorm, reductions stay fp32, GELU to 5e-6.
 *   "train_attn_form"  1 (default) / 0: with train_precision 16, attention axes of 129 .. 256 positions (the ATLAS training shapes)
 *                      run their backward pass in ONE launch of one workgroup per (sequence, head) that converts the sequence's
 *                      q, k, v, dO to bf16 tiles in LDS once and runs the query pass and the key pass out of LDS
 *                      (k16_attn_bwd_seq: the query pass and the key pass of a sequence as neighbouring workgroups of one XCD, each
 *                      with the other side's rows resident) instead of the chunked pair k16_attn_bwd_q / _kv; the forward likewise
 *                      (k16_attn_seq); these kernels apply RoPE to q, k while they convert them, so no RoPE pass is launched for
 *                      such an axis.  Same products; q is rounded to bf16 after the factor log2(e) (scores in log2 units), the
 *                      rotation uses the hardware sine / cosine (3e-5 rad); 0 keeps the chunked kernels for every length.
 *                      586 -> 419 us per backward launch, 28.9 -> 27.4 ms per step at cfg-5's per-GPU size.
 *   "train_defer_gate" 1 (default) / 0: forward pass of mdgen_train_forward_backward, trunk layers: a sub-layer does not apply its gated
 *                      residual update h += gate * u in a pass of its own; the next sub-layer's LayerNorm launch forms x + gate * u in
 *                      registers, writes it to the tape and normalises it (k32_gate_ln_mod); the stream is materialised once, after
 *                      the last layer.  Same arithmetic per element; 26.9 -> 25.9 ms per step at cfg-5's per-GPU size.
 *   "train_turn_ahead" 1 (default) / 0: with train_streams 2 and train_precision 16, the turned weights of the dX products too small
 *                      for the streamed kernel (W^T through k32_transpose: the IPA stack's 256-row launches, ~50 per step) are
 *                      computed on the second stream at the START of the call, beside the forward pass, from the list of requests the
 *                      previous call recorded; a call that asks for something else falls back to a launch in place and records
 *                      anew.  Same values; 25.6 -> 25.2 ms per step.  (Images live in a context-owned buffer, ~30 MB at cfg-5.)
 *   "train_y_bf16", "train_dqkv_bf16", "train_du_bf16", "train_dhid_bf16"   1 (default) / 0, each: with train_precision 16, a
 *                      tensor of the training step that is only ever a GEMM operand is stored as bf16 rows by its producer (launches of
 *                      >= 4096 rows: the trunk) -- the trunk's taped LayerNorm + modulate outputs; dq | dk | dv of the
 *                      sequence-resident attention backward; du = gate * dh; d pre = d hid * gelu'(pre).  The kernels that read them
 *                      round them to bf16 anyway: weight and activation gradients are bit-identical, the bias gradients of the
 *                      layers whose dY is stored rounded differ by ~1e-3 (column sums of the stored values).  25.2 -> 24.2 ms per step.
 *   "train_streams"    2 (default) / 1: mdgen_train_forward_backward launches the weight / bias gradients (nothing reads them
 *                      before the optimiser) on a second stream of the context, beside the backward pass's critical path on the
 *                      caller's stream; it joins the caller's stream before the call returns, and milestone events are recorded
 *                      once both streams have reached them.  Bit-identical gradients; 33.9 -> 29.9 ms per step at cfg-5's
 *                      per-GPU size.
 * Returns -4 for an unknown name, -2 for a value out of range. */
int32_t mdgen_ctx_set_option(mdgen_ctx* ctx, const char* name, int32_t value);
/* number of state_dict keys the model needs; name of the i-th (for loaders / tests) */
int32_t mdgen_ctx_num_weights(const mdgen_ctx* ctx);
const char* mdgen_ctx_weight_name(const mdgen_ctx* ctx, int32_t i);

/* ---- workspace ---------------------------------------------------------------------------
 * `n_steps` = number of distinct time rows prepared per call (1 for a single forward, S for an
 * S-step Euler rollout).  `t_shared` = 1 when all batch elements share t within a step (always true
 * in sampling, integrators.py:99), which de-duplicates the adaLN table. */
int32_t mdgen_workspace_layout(const mdgen_ctx* ctx, const mdgen_shape* shape, int32_t n_steps,
                               int32_t t_shared, mdgen_ws_layout* out);

/* ---- denoiser ----------------------------------------------------------------------------
 * `LatentMDGenModel.forward` / `.forward_inference` (latent_model.py:212-269), non-design path:
 *   out[B,T,L,D] = model(x, t, mask, start_frames, end_frames, x_cond, x_cond_mask, aatype)
 * x, x_cond, out: fp32 (B,T,L,D); t: fp32 (B); mask: fp32 (B,T,L) in {0,1};
 * x_cond_mask: int64 (B,T,L) in {0,1}; aatype: int64 (B,L) in [0,20];
 * start/end frames: rot (B,L,3,3) + trans (B,L,3) fp32 (the fields of the reference's `Rigid`);
 * end_* may be NULL unless tps_condition.  rel7 (nullable; two-sided models only): fp32 (2,B,L,7), the relative-frame inputs
 * of latent_to_emb_f / _r exactly as the caller's reference computes them -- rel7[0] = (start^-1 o end).to_tensor_7(),
 * rel7[1] = (end^-1 o start).to_tensor_7() (latent_model.py:193-195).  Their quaternion SIGN is whatever torch.linalg.eigh
 * returns (rigid_utils.py:191-210, never canonicalised on this path) and it reaches a Linear, so a checkpoint trained with
 * the reference sees exactly its own inputs only when they are handed over; NULL: the library computes them from the
 * frames with the sign fixed to w >= 0.  trace_h (nullable): fp32 [(num_layers+1)][N][384],
 * the residual stream before layer 0 and after every trunk layer; trace_ipa (nullable): fp32
 * [B*L][384], the IPA-stack output (latent_model.py:245-246). */
int32_t mdgen_denoiser_forward(mdgen_ctx* ctx, const mdgen_shape* shape,
                               const float* x, const float* t, const float* mask,
                               const float* start_rot, const float* start_trans,
                               const float* end_rot, const float* end_trans, const float* rel7,
                               const float* x_cond, const int64_t* x_cond_mask, const int64_t* aatype,
                               float* out, float* trace_h, float* trace_ipa,
                               void* workspace, size_t workspace_bytes, void* stream);

/* `Sampler.sample_ode(sampling_method='euler', num_steps=n_steps+1)` -> `ode.sample` ->
 * torchdiffeq fixed-grid Euler (transport.py:408-451, integrators.py:95-114) with the velocity
 * drift (transport.py:242-244): x <- x + (t[i+1]-t[i]) * model(x, t[i]) on t = linspace(0,1,n_steps+1).
 * `x` holds the noise zs on entry and samples[-1] on exit (the only state the wrapper reads,
 * wrapper.py:447).  use_graph=1 captures the whole rollout into a hipGraph on first use (keyed on
 * shape + pointers) and replays it afterwards; `stream` must then be a non-default stream. */
int32_t mdgen_sample_euler(mdgen_ctx* ctx, const mdgen_shape* shape, int32_t n_steps,
                           float* x, const float* mask,
                           const float* start_rot, const float* start_trans,
                           const float* end_rot, const float* end_trans, const float* rel7,
                           const float* x_cond, const int64_t* x_cond_mask, const int64_t* aatype,
                           void* workspace, size_t workspace_bytes, int32_t use_graph, void* stream);

/* Residue constant tables (device pointers; data of mdgen/residue_constants.py:1124-1216, 1367-1480), as the two
 * geometry entry points below take them one by one. */
typedef struct mdgen_residue_tables {
    const float*   default_frames;     /* [21][8][4][4] */
    const float*   lit_positions;      /* [21][14][3]   */
    const int64_t* atom14_group;       /* [21][14]      */
    const float*   atom14_mask;        /* [21][14]      */
    const int64_t* atom37_to_atom14;   /* [21][37]      */
    const float*   atom37_mask;        /* [21][37]      */
    const int64_t* chi_atom_indices;   /* [21][4][4]    */
    const float*   chi_angles_mask;    /* [21][4]       */
} mdgen_residue_tables;

/* The driver loop of `sim_inference.py:100-113` (`do`: num_rollouts x `rollout`, :61-98) in ONE call and -- with
 * use_graph -- one hipGraph: for block r < n_blocks
 *     ex      = conditioning frame expanded over T                      (sim_inference.py:72-79)
 *     prep    = prep_batch(ex)  -> x_cond, x_cond_mask                  (wrapper.py:298-342)
 *     samples = Euler(zs[r], model(., start_frames = cond frame))       (wrapper.py:439-447; as mdgen_sample_euler)
 *     atom14[:, r*T:(r+1)*T] = frames_torsions_to_atom14(cond o offsets, torsions)   (wrapper.py:456-478)
 *     cond frame <- atom14_to_frames / atom37_to_torsions of the block's last frame   (sim_inference.py:91-96)
 * with nothing returning to the host between blocks.  Forward-simulation models only (sim_condition).
 *   zs            (n_blocks, B, T, L, D) fp32: noise on entry, each block's samples[-1] on exit
 *   mask          (B, T, L) fp32;  seqres (B, L) int64 (residue types: the model's aatype AND the geometry's)
 *   cond_*        the conditioning frame: rots (B,L,3,3), trans (B,L,3), torsions (B,L,7,2); UPDATED in place to
 *                 the frame that would condition block n_blocks (so a further call continues the trajectory)
 *   x_cond, x_cond_mask   caller-owned scratch (B,T,L,D) fp32 / (B,T,L) int64
 *   atom14        (B, n_blocks*T, L, 14, 3) fp32 out
 * workspace: as mdgen_sample_euler (mdgen_workspace_layout with n_steps, t_shared = 1). */
int32_t mdgen_rollout_euler(mdgen_ctx* ctx, const mdgen_shape* shape, int32_t n_steps, int32_t n_blocks,
                            float* zs, const float* mask,
                            float* cond_rots, float* cond_trans, float* cond_torsions,
                            const int64_t* seqres, float* x_cond, int64_t* x_cond_mask,
                            const mdgen_residue_tables* tables, float* atom14,
                            void* workspace, size_t workspace_bytes, int32_t use_graph, void* stream);

/* ---- measurement ----------------------------------------------------------------------------
 * Per-kernel-class timing with hipEvents recorded on the launch stream (bench.py's roofline leg).
 * While enabled, launches are bracketed by event pairs and hipGraph capture/replay is bypassed.
 * `mdgen_profile_report` synchronises `stream`, writes a JSON object
 *   {"<class>": {"count": n, "ms": total_ms}, ...}   into buf (NUL-terminated) and resets the log.  Classes of the 64-row panel
 * kernels that exist in two forms carry "@p4" / "@p8" (four / eight waves per panel; option panel_waves), "@p8x3" / "@p8x2" the split
 * forms, the fused attention "@q64" / "@q128" (k_flash_proj / k_flash_proj8), the gate-folded row-owner MLP "mlp@fold" ("+final", "+final+embed": its
 * tails, option mlp_tail).  One entry is
 * not a kernel class: "@context": {"count": split-form MLP launches since the last report, "xcd_round_robin": 0/1 (the placement
 * probe of mdgen_ctx_create: workgroups with equal blockIdx % 8 share an XCD -- where it fails the split MLP form is never picked),
 * "ncu": compute units}. */
int32_t mdgen_profile_enable(mdgen_ctx* ctx, int32_t on);
/* Measurement only: the NEXT trunk MLP launch (the dominant kernel) writes per-wave s_memtime phase stamps into
 * dev_buf ([workgroup*4 + wave][32] uint64; slot meaning: csrc/k_gemm.hip `stamp`).  One-shot; pass NULL to cancel.
 * Only meaningful with profiling enabled or use_graph == 0 (a captured graph would replay the pointer).
 * Which kernel is traced: the row-owner kernel (k_mlp_rows) where the launch takes it, else the FOUR-wave panel kernel k_mlp<3> --
 * also for launches that would otherwise take the eight-wave k_mlp8 (at most one workgroup per CU), which carries no stamps: a
 * trace of such a launch measures k_mlp<3>, not the product's kernel for that size. */
int32_t mdgen_profile_phase_trace(mdgen_ctx* ctx, uint64_t* dev_buf, int64_t capacity_words);
int32_t mdgen_profile_report(mdgen_ctx* ctx, void* stream, char* buf, size_t buflen);
/* Host only (no device, no context): which kernel classes a call of this shape launches and how often -- the library's own
 * orchestration code (the code path of latent_model.py:212-260 / transport.py:408-451's replacements above) run in a plan mode
 * that skips every HIP call.  mode 0: mdgen_sample_euler as the product runs it (sub-batch streams); 1: mdgen_denoiser_forward;
 * 2: mdgen_sample_euler as it runs under mdgen_profile_enable (one stream); 3: mdgen_denoiser_forward with trace_h.  options: "name=value,..." with mdgen_ctx_set_option's
 * names (bf16 path only).  ncu / xcd_round_robin: the two device facts mdgen_ctx_create would have probed (see "@context" of
 * mdgen_profile_report).  Writes {"streams": n, "prepare": {"<class>": launches, ...} (the step-invariant part: adaLN table, IPA stack, fold pack),
 * "views": [{"B": samples of the sub-batch view, "classes": {...}}, ...]} with the class names of mdgen_profile_report.  tests/test_dispatch_cpu.py sweeps shapes with it and fails when a combination of
 * kernel forms has no oracle-backed GPU test registered. */
int32_t mdgen_debug_dispatch_plan(const mdgen_shape* shape, int32_t n_steps, int32_t mode, int32_t tps_condition,
                                  int32_t num_layers, int32_t ncu, int32_t xcd_round_robin, const char* options,
                                  char* buf, size_t buflen);

/* Host-only (no GPU): how a call of `shape` is cut into contiguous sub-batch launch views.  One launch addresses
 * the residual stream with 32-bit byte offsets (token * 1536), i.e. at most 2 796 202 token rows; larger batches
 * run as several views (at least `streams` of them).  Fails with -2 if a single sample (T*L tokens) exceeds the
 * limit -- `mdgen_workspace_layout`, `mdgen_denoiser_forward` and `mdgen_sample_euler` reject such shapes too. */
int32_t mdgen_debug_view_plan(const mdgen_shape* shape, int32_t streams, int32_t* n_views,
                              int32_t* max_batch_per_view);

/* Host-only (no GPU): the weight-row / bias permutations behind the attention fragment layout
 * (DESIGN.md "fragment layout"), each int32[384], for layout tests:
 *   map_qk[w*96+rho], map_vflash[w*96+col], map_vsmall[w*96+rho] = source feature of a packed weight row;
 *   perm_qk[((w*2+h)*4+hd)*12+e], perm_vsmall[...] = source feature of lane-ordered bias slot. */
int32_t mdgen_debug_layout_maps(int32_t* map_qk, int32_t* map_vflash, int32_t* map_vsmall,
                                int32_t* perm_qk, int32_t* perm_vsmall);

/* Host-only (no GPU): the weight-fragment stream of the row-owner MLP kernel (csrc/k_rows.hip), for layout tests.
 * out[f] = mat << 16 | row_tile << 8 | k_step for fragment f (2304 of them); mat 0 = fc1 (layers.py:77-84 `fc1`), 1 = fc2.
 * Returns the number of entries, or a negative status. */
int32_t mdgen_debug_mlp_stream_table(int32_t* out, int32_t capacity);

/* Test hooks (GPU): the training step's linear layer and weight gradient on raw device buffers, through exactly the
 * kernel dispatch of mdgen_train_forward_backward -- precision 32: fp32 products (k32_linear / k32_dw); 16: bf16-rounded
 * operands with fp32 accumulation (128 x 384-tile streamed kernels for >= 1024 / 4096 rows, one-wave tiles for a few
 * hundred rows, the general kernels otherwise).  All pointers are fp32 device memory.
 *   linear: c[n][m] (row stride ldc) = a[n][k] (lda) . w[m][k]^T (ldw) + bias[m] (bias may be NULL);
 *           scratch: >= m * k * 2 bytes (the bf16 weight stream of the streamed kernel), may be NULL (then the weight is
 *           read as fp32 rows);
 *   dw:     dw[m][k] += dy[n][m]^T (ldy) . x[n][k] (ldx), db[m] += column sums of dy (db may be NULL);
 *           part: scratch of part_floats floats for the split partial sums (>= 2 m (k + 1)).
 * Reference: torch.nn.functional.linear and its autograd (mdgen/model/layers.py Mlp, mha.py projections). */
int32_t mdgen_debug_train_linear(int32_t precision, const float* a, int32_t lda, const float* w, int32_t ldw, const float* bias,
                                 int64_t n, int32_t m, int32_t k, float* c, int32_t ldc, void* scratch, void* stream);
int32_t mdgen_debug_train_dw(int32_t precision, const float* dy, int32_t ldy, const float* x, int32_t ldx, int64_t n, int32_t m,
                             int32_t k, float* dw, float* db, float* part, int64_t part_floats, void* stream);

/* Test hook (GPU): the training step's attention of one axis, forward + backward, on raw device buffers (fp32):
 *   qkv[ntok][1152]: q (already scaled and rotated) | k (rotated) | v of 16 heads x 24 (mha.py:258-268); sequence `s`, position
 *   `i` is token (s / inner) * outer_stride + (s % inner) * inner_stride + i * pos_stride; mask[ntok]: 0 = padded key;
 *   bias_k, bias_v[384]: the learned bias key / value (rotated at position len by the kernels); inv_freq[12].
 * Forward: out[ntok][384], lse[ntok][16].  Backward from dout[ntok][384]: dqkv[ntok][1152] = (d q, d k taken back through RoPE,
 * d q also through the q scale; d v), dbias[nseq][768] = per-sequence (d bias_k | d bias_v); stats[ntok][16][2] scratch.
 * precision 32: k32_attn* + k32_rope_bwd; 16: k16_attn* (bf16 operands on the MFMA, inverse RoPE in the store stage) as the training
 * step dispatches them; 160: the chunked k16 kernels for every length (the A/B of option "train_attn_form"); 161 (len 129 .. 256
 * only): q, k of `qkv` are given UNROTATED (q scaled) and the sequence-resident kernels rotate them while they convert them,
 * as the training step runs them (it launches no RoPE pass for such an axis). */
int32_t mdgen_debug_train_attention(int32_t precision, const float* qkv, int64_t ntok, int32_t nseq, int32_t len, int32_t inner,
                                    int32_t outer_stride, int32_t inner_stride, int32_t pos_stride, const float* mask,
                                    const float* bias_k, const float* bias_v, const float* inv_freq, const float* dout,
                                    float* out, float* lse, float* dqkv, float* dbias, float* stats, void* stream);

/* ---- SE(3) frame algebra, fp32 (mdgen/rigid_utils.py) --------------------------------------
 * n = number of frames; rot: [n][3][3]; trans/pts: [n][3]; quat: [n][4] (w,x,y,z). */
int32_t mdgen_rigid_compose(int64_t n, const float* r1, const float* t1, const float* r2, const float* t2,
                            float* r_out, float* t_out, void* stream);            /* :1031-1045 */
int32_t mdgen_rigid_invert(int64_t n, const float* r, const float* t, float* r_out, float* t_out,
                           void* stream);                                          /* :1075-1085 */
int32_t mdgen_rigid_apply(int64_t n, int64_t pts_per_frame, const float* r, const float* t,
                          const float* pts, float* out, int32_t inverse, void* stream); /* :1047-1073 */
int32_t mdgen_quat_to_rot(int64_t n, const float* quat, int32_t normalize, float* rot, void* stream); /* :168-188, :324 */
int32_t mdgen_rot_to_quat(int64_t n, const float* rot, float* quat, void* stream); /* :191-210 (sign: w >= 0) */
/* Rigid.from_3_points(p_neg_x, origin, p_xy, eps = 1e-8) (:1175-1218): Gram-Schmidt frame, R columns e0, e1, e2,
 * t = origin.  Points (n, 3); rot (n, 3, 3); trans (n, 3). */
int32_t mdgen_from_3_points(int64_t n, const float* p_neg_x, const float* origin, const float* p_xy, float* rot,
                            float* trans, void* stream);

/* ---- sampler pre/post-processing (mdgen/wrapper.py) ----------------------------------------
 * `NewMDGenWrapper.prep_batch` latents (wrapper.py:298-327, 339-342, 362) incl. `utils.get_offsets`
 * (utils.py:7-14): offsets = rigid[b,0]^-1 o rigid[b,t] as [quat(w>=0) | trans], TPS appends the
 * offsets w.r.t. frame T-1; latents = [offsets | torsions(14)]; cond frames = 0 (and T-1 for TPS), plus every
 * cond_interval-th frame when cond_interval > 0 (`--cond_interval`, wrapper.py:343-344: the upsampling models; 0 = none).
 * rots (B,T,L,3,3), trans (B,T,L,3), torsions (B,T,L,7,2) -> latents, x_cond (B,T,L,D),
 * x_cond_mask int64 (B,T,L). */
int32_t mdgen_prep_latents(const mdgen_shape* shape, int32_t tps, int32_t cond_interval, const float* rots, const float* trans,
                           const float* torsions, float* latents, float* x_cond, int64_t* x_cond_mask,
                           void* stream);

/* `NewMDGenWrapper.inference` tail (wrapper.py:456-478) + `geometry.frames_torsions_to_atom14`
 * (geometry.py:61-79, 236-334): samples (B,T,L,D) + first-frame rigids (rot0 (B,L,3,3), trans0
 * (B,L,3)) + seqres int64 (B,L) -> atom14 (B,T,L,14,3).  Residue constant tables (device):
 * default_frames [21][8][4][4], lit_positions [21][14][3], atom14_group int64 [21][14],
 * atom14_mask [21][14] (mdgen/residue_constants.py:1124-1216). */
int32_t mdgen_samples_to_atom14(const mdgen_shape* shape, int32_t latent_dim, int32_t tps,
                                const float* samples, const float* rot0, const float* trans0,
                                const int64_t* seqres, const float* default_frames,
                                const float* lit_positions, const int64_t* atom14_group,
                                const float* atom14_mask, float* atom14, void* stream);

/* Rollout glue (sim_inference.py:91-96): last frame atom14 (B,L,14,3) -> next block's conditioning
 * frame: `geometry.atom14_to_frames` (geometry.py:218-231) + `atom14_to_atom37` + `atom37_to_torsions`
 * (geometry.py:9-27, 82-202).  Tables: atom37_to_atom14 int64 [21][37], atom37_mask [21][37],
 * chi_atom_indices int64 [21][4][4], chi_angles_mask [21][4].  Outputs rots (B,L,3,3),
 * trans (B,L,3), torsions (B,L,7,2), torsion_mask (B,L,7). */
int32_t mdgen_atom14_to_cond(int32_t B, int32_t L, const float* atom14, const int64_t* seqres,
                             const int64_t* atom37_to_atom14, const float* atom37_mask,
                             const int64_t* chi_atom_indices, const float* chi_angles_mask,
                             float* rots, float* trans, float* torsions, float* torsion_mask,
                             void* stream);

/* Flow-matching training target (transport.py:138-189 `training_losses`, velocity model; path.py:113-135 `plan`
 * with GVPCPlan :177-187 or the linear ICPlan): per sample b with time t[b],
 *   xt = alpha x1 + sigma x0,  ut = alpha' x1 + sigma' x0;   GVP: alpha = sin(pi t/2), sigma = cos(pi t/2);
 *   Linear: alpha = t, sigma = 1 - t.   x0/x1/xt/ut: (B, per_sample) fp32; path_type 0 = Linear, 1 = GVP. */
int32_t mdgen_path_plan(int64_t B, int64_t per_sample, int32_t path_type, const float* t, const float* x0,
                        const float* x1, float* xt, float* ut, void* stream);

/* Masked mean squared error per sample (transport.py:13-17 `mean_flat`, :184):
 *   loss[b] = sum((pred - target)^2 * mask) / sum(mask)  over the per_sample elements of sample b. */
int32_t mdgen_masked_mse(int64_t B, int64_t per_sample, const float* pred, const float* target, const float* mask,
                         float* loss, void* stream);

/* ---- optimiser side of the training step ----------------------------------------------
 * Parameters, gradients and the two Adam moments each live in ONE flat fp32 device buffer of n elements.
 *
 * mdgen_grad_sumsq: out[0] = sum((grads[i] * scale)^2), deterministic (fixed block slices, fp64 accumulation); `scratch`: 1024 floats, 8-byte aligned.
 *   Feeds gradient clipping (train.py:56 gradient_clip_val -> torch.nn.utils.clip_grad_norm_, 2-norm) without a
 *   host round trip: pass `out` as `sumsq` below.
 * mdgen_adam_step: torch.optim.Adam / AdamW (wrapper.py:167-172; betas (0.9, 0.999), eps 1e-8 are torch's defaults;
 *   adamw != 0: decoupled weight decay).  g = grads * grad_scale * min(1, max_norm / (sqrt(sumsq) + 1e-6)) when sumsq is
 *   non-NULL (clip_grad_norm_), else grads * grad_scale (grad_scale = 1 / world_size averages a summed all-reduce).
 *   `step` is the 1-based update count (bias correction 1 - beta^step).
 * mdgen_ema_update: ema -= (ema - params) * (1 - decay)   (ema.py:41-58). */
int32_t mdgen_grad_sumsq(int64_t n, const float* grads, float scale, float* scratch, int32_t scratch_floats,
                         float* out, void* stream);
int32_t mdgen_adam_step(int64_t n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                        int32_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                        int32_t adamw, float grad_scale, const float* sumsq, float max_norm, void* stream);
int32_t mdgen_ema_update(int64_t n, float* ema, const float* params, float decay, void* stream);

/* ---- training step: forward + backward -------------------------------------------------
 * `NewMDGenWrapper.general_step` (wrapper.py:367-403) after the flow-matching plan: pred = model(xt, t, ...);
 * loss[b] = mean_flat((pred - target)^2, loss_mask) (transport.py:13-17,184); total = mean_b loss[b]; and
 * d total / d theta for every trainable tensor, ADDED into `grads` (flat fp32) at grad_offsets[i] (in floats) for the
 * context's i-th state_dict key (mdgen_ctx_weight_name; -1 = do not compute / frozen).  Runs the fp32-operand form of
 * the network (option keep_fp32_weights must have been set before the weights were handed over), with a tape in
 * `tape` (mdgen_train_workspace_bytes; 256-byte aligned); `workspace` as for mdgen_denoiser_forward (n_steps 1,
 * t_shared 0).  Forward-simulation models (end_rot / end_trans NULL) and the two-sided TPS model (end frames
 * required; its IPA stack runs twice on shared weights, latent_model.py:193-205).  xt, target, loss_mask, pred:
 * (B,T,L,D); t, loss: (B). */
int32_t mdgen_train_workspace_bytes(const mdgen_ctx* ctx, const mdgen_shape* shape, size_t* bytes);
/* Gradient milestones, for overlapping the DDP all-reduce with the backward pass (Lightning DDP's bucketed hooks,
 * train.py:46-77): the backward pass completes parameter groups in the order
 *   0: emb_to_latent.* | 1 .. nl: layers.{nl-1 .. 0}.* | nl+1: latent_to_emb, cond_to_emb, mask_to_emb |
 *   nl+2 .. 2nl+1: ipa_layers.{nl-1 .. 0}.* | 2nl+2: everything else (aatype_to_emb, latent_to_emb_f/_r, t_embedder)
 * and records the caller's hipEvent_t events[k] (NULL = skip) on the training stream when group k's gradients are
 * final.  The list stays in force until replaced (n = 0 clears it); the events remain the caller's. */
/* Point the fp32 weights the training kernels read at `flat + offsets[i]` (i = index of mdgen_ctx_weight_name; -1: keep the
 * context's own copy, e.g. frozen buffers): the optimiser then updates what the next step reads, with no hand-back.
 * Needs keep_fp32_weights = 1 and loaded weights.  The caller keeps `flat` alive as long as the context trains. */
int32_t mdgen_train_bind_params(mdgen_ctx* ctx, float* flat, const int64_t* offsets);
int32_t mdgen_train_num_milestones(const mdgen_ctx* ctx);
int32_t mdgen_train_set_milestone_events(mdgen_ctx* ctx, void* const* events, int32_t n);
int32_t mdgen_train_forward_backward(mdgen_ctx* ctx, const mdgen_shape* shape, const float* xt, const float* t,
                                     const float* mask, const float* start_rot, const float* start_trans,
                                     const float* end_rot, const float* end_trans, const float* rel7,
                                     const float* x_cond, const int64_t* x_cond_mask, const int64_t* aatype,
                                     const float* target, const float* loss_mask, float* loss, float* pred,
                                     float* grads, const int64_t* grad_offsets, void* workspace, size_t workspace_bytes,
                                     void* tape, size_t tape_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MDGEN_AMD_H */
