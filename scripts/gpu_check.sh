#!/bin/bash
# usage (on the GPU box, via gpurun): bash scripts/gpu_check.sh <tag> [bench args]
# runs the gpu test-suite, then a short bench; writes logs under gpurun_out/
tag=$1; shift
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$tag.log 2>&1
rc=$?
tail -4 gpurun_out/pytest_gpu_$tag.log
if [ $rc -ne 0 ]; then echo "TESTS FAILED rc=$rc"; exit $rc; fi
timeout -k 10 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/bench_$tag.log 2>&1 || { tail -5 gpurun_out/bench_$tag.log; exit 1; }
tail -1 gpurun_out/bench_$tag.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('paths/s %.0f  ms/step %.2f  k_scan3 %.2f ms  scan phase %.2f ms  dp %.2f ms  call %.2f ms  frac %s  tile %d wg %d dp_pairs %d' % (
    d['value'], d['ms_per_step'], r['kernel_ms'], r['scan_phase_ms'], r['dp_kernel_ms'], r['call_ms'],
    ('%.3f' % r['frac']) if r['frac'] is not None else 'withheld (counters of another build)',
    d['config']['tile_paths'], d['config']['workgroups'], d['config']['dp_pairs_per_step']))
"
