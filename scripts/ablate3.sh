#!/bin/bash
# k_scan3 with parts compiled out (GFAL_ABLATE3 = 1 loads only, 2 + LDS reads, 3 no counting; results wrong on purpose)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for v in ${ABL:-0 1 2 3}; do
  so=gfalign_amd/csrc/libgfalign_scorer_ab3_$v.so; [ $v = 0 ] && so=gfalign_amd/csrc/libgfalign_scorer.so
  GFALIGN_SCORER_SO=$PWD/$so rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/ab_$v -- python3 scripts/scan_probe.py ${1:-config3} "" > /dev/null 2>&1
  echo "ablate $v: $(python3 scripts/kstats.py gpurun_out/r03/ab_$v | grep -E '^k_scan3' )"
done
