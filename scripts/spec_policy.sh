#!/bin/bash
# GPU box: `gfalign search -m <steps>` under the two speculation policies (best-first by
# alignment support of the edges, breadth-first), a few budgets: same rows, batches,
# candidates scored, loop time.   bash scripts/spec_policy.sh <config> [steps] [budgets]
cfg=${1:-config3}; steps=${2:-20000}; specs=${3:-128,256}
cd "$GRAFT_REPO_ROOT" || exit 1
d=/tmp/sp_$cfg
[ -f $d/g.gfa ] || python3 scripts/gen_files.py $cfg $d > /dev/null
V=$(grep -c '^S' $d/g.gfa)
run() {   # policy budget [extra env...]
  local pol=$1 spec=$2; shift 2
  env GFALIGN_SPEC_POLICY=$pol GFALIGN_SPECULATE=$spec "$@" ./gfalign_amd/csrc/gfalign search -f $d/g.gfa -g $d/a.gaf -n $d/nodes.tsv \
      -s utig4-0 -d utig4-$((V-1)) -m $steps --verbose > /tmp/sp_out_$pol.txt 2> /tmp/sp_err_$pol.txt || { echo FAILED; tail -3 /tmp/sp_err_$pol.txt; exit 1; }
  echo "$cfg $pol budget $spec $*: $(grep -o 'search [0-9.]* s (candidates [0-9.]* s, scoring [0-9.]* s' /tmp/sp_err_$pol.txt) / $(grep -o 'scored [0-9]* candidate paths in [0-9]* batches' /tmp/sp_err_$pol.txt) / $(grep -o 'needed [0-9]*' /tmp/sp_err_$pol.txt)"
}
for spec in ${specs//,/ }; do
  run bfs $spec; cp /tmp/sp_out_bfs.txt /tmp/sp_ref.txt
  run best $spec; cmp /tmp/sp_out_best.txt /tmp/sp_ref.txt || echo "  ROWS DIFFER"


  run best $spec GFALIGN_SPEC_MIN_LIKE=0.0005; cmp /tmp/sp_out_best.txt /tmp/sp_ref.txt || echo "  ROWS DIFFER"
done
