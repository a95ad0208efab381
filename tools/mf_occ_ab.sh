for i in 1 2; do for o in 4 5; do MI_MF_OCC=$o python tools/mf_ablate.py 59 2>&1 | head -1 | sed "s/^/OCC=$o /"; done; done
