#!/bin/bash
# GPU box: `gfalign search -m <steps>` on a synthetic tangle with candidates scored
# from their parents (default) and in full (GFALIGN_INCREMENTAL=0); same rows, times.
#   bash scripts/e2e_incr.sh <config> [steps] [speculate list]
cfg=${1:-config3}; steps=${2:-20000}; specs=${3:-128}
cd "$GRAFT_REPO_ROOT" || exit 1
d=/tmp/ei_$cfg
python3 scripts/gen_files.py $cfg $d > /dev/null
V=$(grep -c '^S' $d/g.gfa)
for spec in ${specs//,/ }; do
  for inc in 1 0; do
    GFALIGN_SPECULATE=$spec GFALIGN_INCREMENTAL=$inc ./gfalign_amd/csrc/gfalign search -f $d/g.gfa -g $d/a.gaf -n $d/nodes.tsv \
      -s utig4-0 -d utig4-$((V-1)) -m $steps --verbose > /tmp/ei_out_$inc.txt 2> /tmp/ei_err_$inc.txt || { echo FAILED; tail -3 /tmp/ei_err_$inc.txt; exit 1; }
    echo "$cfg -m $steps spec $spec incremental=$inc: $(grep '^time' /tmp/ei_err_$inc.txt) / $(grep '^scored' /tmp/ei_err_$inc.txt | cut -c1-150)"
  done
  cmp /tmp/ei_out_1.txt /tmp/ei_out_0.txt && echo "  rows identical ($(wc -l < /tmp/ei_out_1.txt) lines)"
done
