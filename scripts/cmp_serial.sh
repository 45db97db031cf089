#!/bin/bash
# GPU box: the k_scan2 launches of a call side by side on the side streams (default) and
# one after the other on the caller's stream (GFAL_SCAN2_SERIAL=1): the bench step and
# search-sized blocking calls
cd "$GRAFT_REPO_ROOT" || exit 1
for v in parallel serial; do
  if [ $v = serial ]; then export GFAL_SCAN2_SERIAL=1; fi
  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-search-mode 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$v', 'ms/step %.3f scan %.3f dp %.3f' % (d['ms_per_step'], r['kernel_ms'], r['dp_kernel_ms']))"
  python3 scripts/small_batch_probe.py config3 128,512,2048 | sed "s/^/$v /"
done
