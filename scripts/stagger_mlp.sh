for d in 0 2 4 6 8 12 16 24; do echo "== MDGEN_STAGGER_MLP=$d $(MDGEN_STAGGER_MLP=$d timeout 300 python scripts/kbench.py 2>&1 | grep -E ' mlp ')"; done
