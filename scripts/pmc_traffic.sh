#!/bin/bash
# HBM traffic per launch of the big kernels (default: cfg-2 and cfg-4; or workloads named as arguments), from rocprofv3 PMC counters, as MI355X_MICROARCH.md
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass), no tracing domains;
# FETCH_SIZE is reported in KiB and, on gfx950, counts 64 B per 128-B request of a wide coalesced read -> doubled.
# Output: gpurun_out/pmc_traffic.json (copy to profiles/).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
RE="k_mlp|k_ln_qkv|k_proj|k_flash|k_final|k_embed|k_ipa_attn"
WLS="${@:-tetrapeptide_fwdsim_crop4_T1000_B16 atlas_crop256_T250_B1}"
for wl in $WLS; do
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_${c}_$wl
  (cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-include-regex "$RE" --output-format csv -d $R/gpurun_out/pmc_${c}_$wl -o pmc -- python $R/scripts/kbench.py $wl 1 > $R/gpurun_out/pmc_${c}_$wl.log 2>&1)
done
done
python - "$R" $WLS <<'PY'
import sys, glob, csv, collections, json
R = sys.argv[1]
meta = {"mode": "single stream, eager launches (library option streams = 1); per kernel: the launches with the largest grid "
                "(the trunk's; the IPA stack launches the same kernels on S*B*L rows)",
        "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950 wide-read correction); KiB -> bytes",
        "workloads": {}}
for wl in sys.argv[2:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"{R}/gpurun_out/pmc_{c}_{wl}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c:
                    acc[(r["Kernel_Name"], int(r["Grid_Size"]))][c].append(float(r["Counter_Value"]))
    out = {}
    biggest = {}
    for (k, g) in acc:
        biggest[k] = max(biggest.get(k, 0), g)
    for (k, g), d in acc.items():
        if g != biggest[k] or "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
            continue
        f = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]); w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        out[k] = {"grid_threads": g, "launches_sampled": len(d["FETCH_SIZE"]),
                  "FETCH_SIZE_KiB_raw": round(f, 1), "WRITE_SIZE_KiB_raw": round(w, 1),
                  "hbm_read_bytes": round(2 * f * 1024), "hbm_write_bytes": round(w * 1024),
                  "hbm_bytes_per_launch": round((2 * f + w) * 1024)}
    meta["workloads"][wl] = {"kernels": out}
    print(wl)
    for k, v in out.items():
        print(f"  {k[:50]:50s} read {v['hbm_read_bytes']/1e6:8.1f} MB  write {v['hbm_write_bytes']/1e6:8.1f} MB")
json.dump(meta, open(f"{R}/gpurun_out/pmc_traffic.json", "w"), indent=1)
PY
