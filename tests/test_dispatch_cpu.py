"""Host-only: the map (shape, options) -> kernel forms -> oracle-backed GPU test, machine-checked.

`mdgen_debug_dispatch_plan` runs the library's own orchestration code in a plan mode (no HIP call is made), so what it
reports is what the product launches.  This file sweeps the shapes the entry points are used with, collects the distinct
combinations of kernel forms ("signatures") and fails when one of them is produced by no entry of tests/dispatch_registry.py
-- i.e. when a launch-size threshold or a new kernel form opens a combination that no `-m gpu` test compares with the oracle
(how round 4's k_mlp<3> hole came about).  Reference: latent_model.py:446-483 (what every form computes)."""
import os
import re

import pytest

from dispatch_registry import CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODE = {"euler": 0, "forward": 1, "euler_profiled": 2, "forward+trace": 3}
NOT_TRUNK = ("embed",)   # (k_embed has one form per launch size and is compared through the h0 trace of every forward test)


def plan(B, T, L, mode, S=2, tps=False, options=None):
    from mdgen_amd._lib import dispatch_plan
    return dispatch_plan(B, T, L, n_steps=S, mode=MODE[mode], tps=tps, options=options)


def view_signatures(p):
    return {frozenset(k for k in v["classes"] if k not in NOT_TRUNK) for v in p["views"]}


def ipa_signature(p):
    return frozenset(k for k in p["prepare"] if k.startswith("ipa."))


def registry_signatures():
    trunk, ipa = {}, {}
    for c in CASES:
        p = plan(c["B"], c["T"], c["L"], c["mode"], S=c.get("S", 2), tps=c.get("tps", False), options=c.get("options"))   # (euler cases run two steps: the embedding-as-tail feeds the second)
        if c.get("part") != "ipa":          # (an IPA-table entry's rollout is not compared beyond the table)
            for s in view_signatures(p):
                trunk.setdefault(s, []).append(c["name"])
        if c.get("part") == "ipa" or c["mode"].startswith("forward"):   # forward tests compare the `ipa_out` trace
            ipa.setdefault(ipa_signature(p), []).append(c["name"])
    return trunk, ipa


def sweep():
    """(tag, plan) over the shapes the entry points are used with."""
    out = []
    for B in range(1, 17):                                   # sim_inference.py --batch 1 .. 16 at the headline's T x L
        for st in (None, 1, 2):
            o = {} if st is None else {"streams": st}
            out.append((f"euler B{B} T1000 L4 streams {st}", plan(B, 1000, 4, "euler", options=o)))
    for L in (8, 9, 16, 24, 33, 40, 64, 80, 96, 128, 160, 200, 256, 300, 400, 512, 724):   # ATLAS chains (sim_inference.py:32-59), B = 1
        out.append((f"euler B1 T250 L{L}", plan(1, 250, L, "euler")))
        out.append((f"forward+trace B1 T250 L{L}", plan(1, 250, L, "forward+trace")))
    for B in (1, 8, 32, 64, 128, 256):                       # cfg-3: TPS, T 100, batch 256 over 8 / 4 / 2 / 1 GPUs
        out.append((f"euler tps B{B} T100 L4", plan(B, 100, 4, "euler", tps=True)))
    for B in (1, 2, 3, 4, 5, 8, 16):                         # forward() (training-side evaluation, tests)
        out.append((f"forward+trace B{B} T1000 L4", plan(B, 1000, 4, "forward+trace")))
        out.append((f"forward B{B} T1000 L4", plan(B, 1000, 4, "forward")))
    return out


def test_plan_is_available_without_a_gpu_and_names_the_headline_kernels():
    p = plan(16, 1000, 4, "euler", S=49)
    assert p["streams"] == 2 and [v["B"] for v in p["views"]] == [8, 8]
    for v in p["views"]:   # BASELINE.json configs[1]: two sub-batch views of B 8
        assert v["classes"] == {"attn_L_fused": 245, "embed": 1, "flash_proj_T@q128": 245, "ln_qkv_T": 245, "mlp@fold": 196,
                                "mlp@fold+final": 1, "mlp@fold+final+embed": 48}, v   # (steps 1 .. 48: no k_embed, no k_final)
    assert p["prepare"]["fold_pack"] == 1 and p["prepare"]["adaln_table"] == 1
    # options reach the plan: without the fold the separate final layer is back
    q = plan(16, 1000, 4, "euler", S=49, options={"mlp_fold": 0})
    assert q["views"][0]["classes"]["mlp"] == 245 and q["views"][0]["classes"]["final_euler"] == 49 and "fold_pack" not in q["prepare"]
    assert q["views"][0]["classes"]["embed"] == 49
    # under the profiler the rollout is one stream of B 16
    assert [v["B"] for v in plan(16, 1000, 4, "euler_profiled")["views"]] == [16]


def test_every_registry_entry_names_an_existing_gpu_test():
    src = open(os.path.join(ROOT, "tests", "test_gpu_parity.py")).read()
    names = set(re.findall(r"^def (test_\w+)\(", src, re.M))
    assert "test_dispatch_registry_case_vs_oracle" in names
    for c in CASES:
        assert c["covered_by"] is None or c["covered_by"] in names, c


def test_every_combination_of_kernel_forms_in_the_sweep_has_an_oracle_backed_gpu_test():
    trunk, ipa = registry_signatures()
    missing = {}
    for tag, p in sweep():
        for s in view_signatures(p):
            if s not in trunk:
                missing.setdefault(("trunk",) + tuple(sorted(s)), []).append(tag)
    assert not missing, "kernel-form combinations without a registered oracle test:\n" + "\n".join(
        f"  {k[1:]}  <- {v[:4]}" for k, v in missing.items())
    # the IPA stack (S * B * L rows per launch): its forms at S = 49 for the bench shapes
    miss_ipa = {}
    for tag, (B, T, L, tps) in {"cfg-2": (16, 1000, 4, False), "B = 1": (1, 1000, 4, False), "ATLAS": (1, 250, 256, False),
                                "cfg-3 shard": (32, 100, 4, True)}.items():
        s = ipa_signature(plan(B, T, L, "euler", S=49, tps=tps))
        known = set().union(*ipa.keys())
        if not s <= known:
            miss_ipa[tag] = sorted(s - known)
    assert not miss_ipa, miss_ipa
    print(f"{len(trunk)} trunk signatures and {len(ipa)} IPA-stack signatures registered")
