for d in 0 1 2 4 3 5 6 7; do echo "== MDGEN_DEBUG_KPROJ=$d"; MDGEN_DEBUG_KPROJ=$d timeout 300 python scripts/kbench.py 2>&1 | grep -E "proj_L|proj_T"; done
