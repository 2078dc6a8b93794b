"""Which oracle-backed `-m gpu` test covers which combination of kernel forms (round-5 verdict, weak #3 / round-6 item 2(d)).

`mdgen_debug_dispatch_plan` (host only: the library's own orchestration code in a plan mode that skips every HIP call) names the
kernel classes a call launches.  A *signature* is the set of trunk classes of one sub-batch view (or the set of IPA-stack classes of
a call).  Every entry below is a call shape whose signature is compared, element by element, with the CPU oracle by a GPU test:
  covered_by = "<test function in tests/test_gpu_parity.py>"  -- an existing test runs this very shape, or
  covered_by = None                                            -- tests/test_gpu_parity.py::test_dispatch_registry_case_vs_oracle runs it.
tests/test_dispatch_cpu.py sweeps the shapes the entry points are used with (B = 1 .. 16 at T 1000 x L 4, L = 8 .. 724 at T 250,
cfg-3's shapes; streams 1 / 2 / default) and FAILS when a signature turns up that no entry here produces.

mode: "forward" (mdgen_denoiser_forward, no trace), "forward+trace" (with trace_h: what `return_trace=True` tests run),
"euler" (mdgen_sample_euler: the generic GPU test runs S = 2 and compares x2 - x0 with two Euler steps of the oracle).
options: library options the case sets (default: none) -- e.g. `streams` 2 where the default stream count would not produce the combination.
"""

CASES = [
    # ---- tetrapeptides (L = 4: the residue axis is k_ln_qkv_attn4<true>), by launch size
    dict(name="B1_T1000_L4", mode="forward+trace", B=1, T=1000, L=4, n_pad=0, covered_by="test_small_launches_split_a_panel_over_workgroups_vs_oracle"),
    dict(name="B2_T1000_L4", mode="forward+trace", B=2, T=1000, L=4, n_pad=0, covered_by="test_forward_headline_regime_vs_reference_and_oracle"),
    dict(name="B2_T1000_L4_euler_2streams", mode="euler", B=2, T=1000, L=4, n_pad=0, options={"streams": 2}, covered_by=None),   # two one-sample views on two streams: 32-row L = 4 / q, k | v forms beside the UNSPLIT eight-wave MLP (its scratch serves one stream)
    dict(name="B3_T700_L4", mode="forward+trace", B=3, T=700, L=4, n_pad=0, covered_by=None),          # 129..256 panels: eight-wave forms, unsplit
    dict(name="B5_T1000_L4", mode="forward+trace", B=5, T=1000, L=4, n_pad=0, covered_by="test_panel_kernels_257_to_383_panels_vs_oracle"),
    dict(name="B8_T1000_L4_fwd", mode="forward+trace", B=8, T=1000, L=4, n_pad=0, covered_by="test_headline_kernel_mix_at_B8_T1000_vs_oracle"),
    dict(name="B8_T1000_L4_euler", mode="euler", B=8, T=1000, L=4, n_pad=0, covered_by="test_headline_kernel_mix_at_B8_T1000_vs_oracle"),
    dict(name="B6_T1024_L4_euler", mode="euler", B=6, T=1024, L=4, n_pad=0, covered_by=None),          # row-owner MLP, but < 512 fused-attention jobs: k_flash + k_proj<0>
    dict(name="B6_T1024_L4_fwd", mode="forward+trace", B=6, T=1024, L=4, n_pad=0, covered_by=None),
    dict(name="B64_T100_L4_euler", mode="euler", B=64, T=100, L=4, n_pad=0, covered_by=None),          # cfg-3's batch on few GPUs: 64-query fused attention on short sequences
    dict(name="tps_B64_T100_L4_euler", mode="euler", B=64, T=100, L=4, n_pad=0, tps=True, covered_by=None),   # the two-sided model (D = 28) through the folded MLP and both tails
    # ---- longer chains (tiled residue axis), T = 250 unless a smaller shape has the same signature
    dict(name="B1_T250_L8", mode="forward+trace", B=1, T=250, L=8, n_pad=1, covered_by=None),          # 4 < L <= 8: micro-attention in k_proj<2>
    dict(name="B1_T250_L16", mode="forward+trace", B=1, T=250, L=16, n_pad=2, covered_by=None),        # <= 85 panels: split forms with the tiled residue axis
    dict(name="B1_T250_L33", mode="forward+trace", B=1, T=250, L=33, n_pad=3, covered_by=None),        # 129 panels: eight-wave forms
    dict(name="B1_T250_L80", mode="forward+trace", B=1, T=250, L=80, n_pad=10, covered_by="test_panel_kernels_257_to_383_panels_vs_oracle"),
    dict(name="B1_T250_L128_euler", mode="euler", B=1, T=250, L=128, n_pad=5, covered_by=None),        # fused attention on the temporal axis only
    dict(name="B1_T250_L128_fwd", mode="forward+trace", B=1, T=250, L=128, n_pad=5, covered_by=None),
    dict(name="B1_T250_L256_fwd", mode="forward+trace", B=1, T=250, L=256, n_pad=16, covered_by="test_forward_cfg4_full_size_vs_reference_and_oracle"),
    dict(name="B1_T192_L192_euler", mode="euler", B=1, T=192, L=192, n_pad=7, covered_by=None),        # ATLAS's signature (both axes fused, folded MLP + final tail) at 58 % of its tokens
    dict(name="B1_T64_L512_euler", mode="euler", B=1, T=64, L=512, n_pad=9, covered_by=None),          # residue axis on the 128-query fused form
    dict(name="B1_T64_L512_fwd", mode="forward+trace", B=1, T=64, L=512, n_pad=9, covered_by=None),    # ... with a trace: folded MLP, separate k_final
    dict(name="B2_T64_L512_fwd", mode="forward+trace", B=2, T=64, L=512, n_pad=9, covered_by=None),    # per-sample t (B > 1 forward): unfolded row-owner MLP, 128-query residue form
    # ---- the IPA stack's launch sizes at S = 49 (S * B * L rows): compared through the table a rollout leaves in the workspace
    dict(name="ipa_ATLAS_S49", mode="euler", B=1, T=2, L=256, n_pad=16, S=49, part="ipa", covered_by="test_ipa_table_of_all_steps_vs_oracle"),
    dict(name="ipa_shard_S49", mode="euler", B=32, T=2, L=4, n_pad=0, S=49, part="ipa", covered_by="test_ipa_table_of_all_steps_vs_oracle"),
]
