#!/bin/bash
# experiment (round 4): smoother degree on the Q1 level of the fine cells alone (MI_MG_NU_L1), headline mesh
for V in 0 1 3; do
  MI_MG_NU_L1=$V python bench.py --steps 8 --warmup 2 --cpu-cells 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MI_MG_NU_L1=$V: %.1f ms per step, %.1f CG iterations per step' % (d['ms_per_step'], d['config']['cg_iterations_per_step']))"
done
