mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 0 3; do
  GFAL_DEBUG_TILE=$v GFAL_DEBUG_SCAN2=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/tp_$v -- python3 scripts/scan_probe.py config3 "" > /dev/null 2>&1
  echo "GFAL_DEBUG_TILE=$v: $(python3 scripts/kstats.py gpurun_out/r03/tp_$v | grep -E '^k_tile ' )"
done
B="python3 bench.py --no-cpu-baseline --no-search-mode --steps 200 --warmup 10"
for e in "X=1" "X=2"; do
env $e $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', d['ms_per_step'], d['config'].get('counter_checksum'), d['roofline']['kernel_ms'])"
done
python3 scripts/scan_probe.py config5 "" 2>&1 | grep -v amdgpu.ids
timeout -k 10 900 python -m pytest tests/test_gpu_scan2.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -n 4
