#!/bin/bash
# GPU box: the bench step with k_scan2 built for 768-thread workgroups (6 waves per
# SIMD, 80 VGPRs) next to the in-tree 1024-thread build (8 waves, 64 VGPRs).
cd "$GRAFT_REPO_ROOT" || exit 1
for so in "" tools/exp/libscorer_t768.so; do
  GFALIGN_SCORER_SO=$so python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-search-mode 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$so' or 'in-tree', 'ms/step %.3f scan %.3f dp %.3f tile %d wg %d' % (d['ms_per_step'], r['kernel_ms'], r['dp_kernel_ms'], d['config']['tile_paths'], d['config']['workgroups']))"
done
