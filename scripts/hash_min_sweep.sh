#!/bin/bash
# GPU box: bench step against GFAL_HASH_MIN_ITEMS (items a length needs to become a k_scan2 segment)
cd "$GRAFT_REPO_ROOT" || exit 1
for v in ${2:-48 100 200 500 1200}; do
  GFAL_HASH_MIN_ITEMS=$v python3 bench.py --workload ${1:-config3} --steps 10 --warmup 2 --no-cpu-baseline --no-search-mode 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('${1:-config3} hash_min_items $v: ms/step %.3f scan %.3f dp %.3f wg %d' % (d['ms_per_step'], r['kernel_ms'], r['dp_kernel_ms'], d['config']['workgroups']))"
done
