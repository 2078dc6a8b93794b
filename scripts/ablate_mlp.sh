for d in 0 1 2 4 8 16 10 14 30 31; do echo "== MDGEN_DEBUG_KMLP=$d $(MDGEN_DEBUG_KMLP=$d timeout 300 python scripts/kbench.py 2>&1 | grep -E ' mlp ')"; done
