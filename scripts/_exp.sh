mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --no-cpu-baseline --no-search-mode --steps 200 --warmup 10"
for e in "X=1" "X=2"; do
env $e $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', d['ms_per_step'], d['config'].get('counter_checksum'), d['roofline']['kernel_ms'], d['roofline']['dp_kernel_ms'])"
done

rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03/tl_sel -- python3 scripts/timeline_probe.py config3 1 0 > gpurun_out/r03/tl_sel.log 2>&1
python3 scripts/ktimeline.py gpurun_out/r03/tl_sel | grep "k_dp\|k_unperm"

