#!/bin/bash
# k_scan time with parts compiled out (results WRONG on purpose): 2 = no subpath search, 6 = no overhang triage
for v in 2 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -DGFAL_ABLATE=$v -I include \
    -o gfalign_amd/csrc/libgfalign_scorer_ab$v.so gfalign_amd/csrc/scorer.hip || exit 1
done
cp gfalign_amd/csrc/libgfalign_scorer.so /tmp/keep.so
for v in keep 2 6; do
  if [ $v != keep ]; then cp gfalign_amd/csrc/libgfalign_scorer_ab$v.so gfalign_amd/csrc/libgfalign_scorer.so; fi
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('ablate $v: scan %.2f ms dp %.2f ms' % (r['kernel_ms'], r['dp_kernel_ms']))"
done
cp /tmp/keep.so gfalign_amd/csrc/libgfalign_scorer.so
