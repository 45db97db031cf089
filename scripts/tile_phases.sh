#!/bin/bash
# k_tile's phases (GFAL_DEBUG_TILE stops it after A / B / C; counters are then wrong): kernel time per setting
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for v in 0 1 2 3 4; do
  GFAL_DEBUG_TILE=$v GFAL_DEBUG_SCAN2=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/tp_$v -- python3 scripts/scan_probe.py ${1:-config3} "" > /dev/null 2>&1
  echo "GFAL_DEBUG_TILE=$v: $(python3 scripts/kstats.py gpurun_out/r03/tp_$v | grep -E '^k_tile ' )"
done
