#!/bin/bash
# GPU box: the bench step with k_scan2 as one launch that knows every alignment length
# (GFAL_SCAN2_SPLIT=0: spills at 64 VGPRs) and as one launch per length group (=1: the
# main groups spill-free), twice each, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do for v in 0 1; do
  GFAL_SCAN2_SPLIT=$v python3 bench.py --workload ${1:-config3} --steps 30 --warmup 3 --no-cpu-baseline --no-search-mode 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('${1:-config3} split=$v: ms/step %.3f scan %.3f dp %.3f' % (d['ms_per_step'], r['kernel_ms'], r['dp_kernel_ms']))"
done; done
