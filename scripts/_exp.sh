mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --no-cpu-baseline --no-search-mode --steps 200 --warmup 10"
for e in "X=1" "X=2"; do
env $e $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', d['ms_per_step'], d['config'].get('counter_checksum'), d['roofline']['kernel_ms'], d['roofline']['dp_kernel_ms'])"
done
python3 scripts/scan_probe.py config5 "" 2>&1 | grep -v amdgpu.ids
timeout -k 10 900 python -m pytest tests/test_gpu_scan2.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -n 4
