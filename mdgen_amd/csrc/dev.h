// dev.h -- the ONE place that knows the experiment switches of the kernels (included by common.h).
//
// Product builds (mdgen_amd/build.py) define none of them.  The experiment builds of scripts/micro/flash_variants.sh pass
// -DMDGEN_DEV_BUILD together with the switch; a switch without it is a build error, so a stray -D cannot silently turn a
// correctness path off (MDGEN_DEV_FLASH_NOFALLBACK) or produce wrong values (MDGEN_DEV_ROWS_COALESCED) in a product
// library, and mdgen_dev_switches() (api.hip) lets a loader see what a given .so was built with.
//
//   MDGEN_DEV_FLASH_STAMPS      k_flash / k_flash_proj: per-wave s_memtime stamps (scripts/micro/flash_stamps.py, scripts/r05/fproj_stamps.py)
//   MDGEN_DEV_FLASH_NOLOAD      k_flash: K / V loads left out, timing only
//   MDGEN_DEV_FLASH_NOFALLBACK  k_flash: no re-run with the robust loop: shows what the fixed anchor alone does
//   MDGEN_DEV_FLASH_TRUNC / _NOPRIO / _NOEARLY   k_flash_proj(8) (round 6 A/B): truncating P pack; WITHOUT the alternating wave priority (128-row form) /
//                               the residual rows requested ahead of the out-projection GEMM (both product since round 6)
//   MDGEN_DEV_QKV_STAMPS        k_ln_qkv / k_ln_qkv_attn4: phase stamps         (scripts/micro/qkv_stamps.py, scripts/r05/attn4_stamps.py)
//   MDGEN_DEV_ATTN4_FULL        k_ln_qkv_attn4: small launches keep 64-row workgroups (round 6 A/B of the half-panel form)
//   MDGEN_DEV_MLP8_STAMPS       k_mlp8: per-wave phase stamps held in SGPRs (scripts/r06/mlp8_stamps.py)
//   MDGEN_DEV_MLP_STAMPX        k_mlp: stamp inside one fc1 stage              (scripts/micro/mlp_stampx.py)
//   MDGEN_DEV_ROWS_NOGELU       k_mlp_rows: main loop without its VALU work, timing only (values wrong)
//   MDGEN_DEV_ROWS_COALESCED    k_mlp_rows: row loads as coalesced 1 KiB requests, timing only (values wrong)
//   MDGEN_DEV_ATTN16_NOEXP / _NOSTAGE / _NOMMA / _NOBAR   k16_attn*: one ingredient of the chunk loop left out, timing only (values wrong)
//   MDGEN_DEV_ATTN16_SEQNOLOAD / _SEQNOLOOP / _SEQNOSTORE   k16_attn_bwd_seq (round 6): the fill's global loads / all but one tile of each pass /
//                               the result stores left out, timing only (values wrong)
//   MDGEN_DEV_WIDE_STAMPS       k16_linear_wide: s_memtime stamps per k-step phase (scripts/micro/wide_stamps.py)
//   MDGEN_DEV_WIDE_NOLOAD / _NOMMA / _NOSTORE   k16_linear_wide: one phase of the k-step left out, timing only (values wrong)
#pragma once



#if defined(MDGEN_DEV_FLASH_TRUNC) || defined(MDGEN_DEV_FLASH_NOPRIO) || defined(MDGEN_DEV_FLASH_NOEARLY) || \
    defined(MDGEN_DEV_FLASH_STAMPS) || defined(MDGEN_DEV_FLASH_NOLOAD) || defined(MDGEN_DEV_FLASH_NOFALLBACK) || \
    defined(MDGEN_DEV_QKV_STAMPS) || defined(MDGEN_DEV_MLP8_STAMPS) || defined(MDGEN_DEV_ATTN4_FULL) ||  defined(MDGEN_DEV_MLP_STAMPX) || defined(MDGEN_DEV_ROWS_NOGELU) ||           \
    defined(MDGEN_DEV_ROWS_COALESCED) || defined(MDGEN_DEV_WIDE_NOLOAD) || defined(MDGEN_DEV_WIDE_NOMMA) ||        \
    defined(MDGEN_DEV_WIDE_NOSTORE) || defined(MDGEN_DEV_WIDE_STAMPS) || \
    defined(MDGEN_DEV_ATTN16_NOEXP) || defined(MDGEN_DEV_ATTN16_NOSTAGE) ||         \
    defined(MDGEN_DEV_ATTN16_NOMMA) || defined(MDGEN_DEV_ATTN16_NOBAR) || defined(MDGEN_DEV_ATTN16_SEQNOLOAD) ||     \
    defined(MDGEN_DEV_ATTN16_SEQNOLOOP) || defined(MDGEN_DEV_ATTN16_SEQNOSTORE)
#ifndef MDGEN_DEV_BUILD
#error "an MDGEN_DEV_* experiment switch is set without -DMDGEN_DEV_BUILD: product libraries are built with none of them"
#endif
#define MDGEN_DEV_ANY 1
#endif
