#!/bin/bash
# time the bench batch and a deep search batch with the worklist appends compiled out (results WRONG on purpose)
cp gfalign_amd/csrc/libgfalign_scorer.so /tmp/keep.so
for v in keep ab5; do
  if [ $v = ab5 ]; then cp gfalign_amd/csrc/libgfalign_scorer_ab5.so gfalign_amd/csrc/libgfalign_scorer.so; fi
  echo "== $v"
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('bench: scan %.2f ms dp %.2f ms' % (r['kernel_ms'], r['dp_kernel_ms']))"
  python scripts/search_batch.py config3 10000 20000 50000 2>&1 | grep "search batch P"
done
cp /tmp/keep.so gfalign_amd/csrc/libgfalign_scorer.so
