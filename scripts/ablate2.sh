#!/bin/bash
# swap in ablation builds of the scorer and time the bench batch (results are WRONG on purpose)
cd gfalign_amd/csrc
cp libgfalign_scorer.so /tmp/keep.so
for v in 1 2; do
  cp libgfalign_scorer_ab$v.so libgfalign_scorer.so
  (cd ../.. && python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('ablate $v: scan %.2f ms dp %.2f ms' % (r['kernel_ms'], r['dp_kernel_ms']))")
done
cp /tmp/keep.so libgfalign_scorer.so
