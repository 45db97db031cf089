#!/bin/bash
# GPU box: search-loop time of `gfalign search -m 20000` against GFAL_CHILD_CHUNKS
cfg=${1:-config3}
cd "$GRAFT_REPO_ROOT" || exit 1
d=/tmp/ei_$cfg
[ -f $d/g.gfa ] || python3 scripts/gen_files.py $cfg $d > /dev/null
V=$(grep -c '^S' $d/g.gfa)
for c in ${2:-1 2 4 8 16 32}; do
  GFAL_CHILD_CHUNKS=$c ./gfalign_amd/csrc/gfalign search -f $d/g.gfa -g $d/a.gaf -n $d/nodes.tsv -s utig4-0 -d utig4-$((V-1)) -m 20000 --verbose 2>&1 >/dev/null | grep '^time' | sed "s/^/$cfg chunks $c: /"
done
