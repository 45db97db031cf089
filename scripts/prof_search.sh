#!/bin/bash
# Run on the GPU box: kernel time shares inside `gfalign search -m <steps>` on a
# synthetic tangle.   bash scripts/prof_search.sh <config> [steps] [speculate]
cfg=${1:-config3}; steps=${2:-20000}; spec=${3:-128}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python3 scripts/gen_files.py $cfg /tmp/ps_$cfg > /dev/null
d=/tmp/ps_$cfg
V=$(grep -c '^S' $d/g.gfa)
export GFALIGN_SPECULATE=$spec
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ps_${cfg} -- ./gfalign_amd/csrc/gfalign search -f $d/g.gfa -g $d/a.gaf -n $d/nodes.tsv -s utig4-0 -d utig4-$((V-1)) -m $steps --verbose > gpurun_out/ps_${cfg}.out 2> gpurun_out/ps_${cfg}.err
tail -2 gpurun_out/ps_${cfg}.err
python3 scripts/kstats.py gpurun_out/ps_${cfg}
