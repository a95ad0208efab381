#!/bin/bash
# A/B (round 4): up to how many slices a launch takes the one-workgroup-per-slice product (sell_spmv_split)
for V in 160 512 1200; do
  MI_SELL_SPLIT_MAX=$V python bench.py --steps 8 --warmup 2 --cpu-cells 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MI_SELL_SPLIT_MAX=$V: %.2f ms per step, %.1f CG iterations per step' % (d['ms_per_step'], d['config']['cg_iterations_per_step']))"
done
for V in 160 512; do
  MI_SELL_SPLIT_MAX=$V python bench.py --slabs 8 --steps 8 --warmup 2 --cpu-cells 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('8 slabs MI_SELL_SPLIT_MAX=$V: %.2f ms per step' % (d['ms_per_step']))"
done
