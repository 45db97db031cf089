#!/bin/bash
for g in 1600 3400 5000 6800 8400 10000 13600 20000; do
  GFAL_SCAN_GROUPS=$g python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('groups $g: wg %d scan %.2f ms step %.2f ms' % (d['config']['workgroups'], r['kernel_ms'], d['ms_per_step']))"
done
