#!/bin/bash
# Run on the GPU box (gpurun):  bash scripts/collect_profiles.sh <round-tag>
# Writes rocprofv3 output under gpurun_out/prof_<tag>_*; scripts/save_profiles.py
# then condenses it into profiles/.  Counter passes are separate rocprofv3 runs
# (kernel-trace + pmc only), as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
B="python3 bench.py --no-cpu-baseline --no-search-mode"
say() { echo "[collect] $*"; }
say stats;  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_stats -- $B --steps 20 --warmup 3 > gpurun_out/prof_${tag}_stats.log 2>&1
say fetch;  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${tag}_fetch -- $B --steps 3 --warmup 1 > /dev/null 2>&1
say write;  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${tag}_write -- $B --steps 3 --warmup 1 > /dev/null 2>&1
say tcc;    rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d gpurun_out/prof_${tag}_tcc -- $B --steps 3 --warmup 1 > /dev/null 2>&1
say sq1;    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH --output-format csv -d gpurun_out/prof_${tag}_sq1 -- $B --steps 3 --warmup 1 > /dev/null 2>&1
say sq2;    rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/prof_${tag}_sq2 -- $B --steps 3 --warmup 1 > /dev/null 2>&1
say grbm;   rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_${tag}_grbm -- $B --steps 3 --warmup 1 > /dev/null 2>&1
# the same instruction counters with the item loops switched off (GFAL_DEBUG_SCAN2=1: every workgroup does its
# prologue -- node masks, table clear, list insert -- and returns; counters are wrong on purpose): what k_scan3
# issues per step minus this is the item loop's share ("useful" in bench.py's roofline)
say sq1p;   GFAL_DEBUG_SCAN2=1 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH --output-format csv -d gpurun_out/prof_${tag}_sq1p -- $B --steps 3 --warmup 1 > /dev/null 2>&1
python3 -c "from gfalign_amd.scorer import load_library; print(load_library().gfal_build_id().decode())" > gpurun_out/build_id_${tag}.txt
say calib;  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${tag}_calib -- ./tools/fetch_calib > gpurun_out/prof_${tag}_calib.log 2>&1
say valu_rate; ./tools/valu_rate > gpurun_out/valu_rate_${tag}.jsonl 2> /dev/null
{
  echo "## search-sized batches (scripts/small_batch_probe.py config3)"; python3 scripts/small_batch_probe.py config3 8,32,128,512,2048
  echo "## candidate batches of the search itself (scripts/search_batch.py config3 10000 20000 <skip>)"
  python3 scripts/search_batch.py config3 10000 20000 0; python3 scripts/search_batch.py config3 10000 20000 50000
  echo "## gfalign search end to end (scripts/e2e_search.py config3 20000 32,128,512)"; python3 scripts/e2e_search.py config3 20000 32,128,512
  echo "## gfalign search -m 20000, candidates from their parents (default) and in full (scripts/e2e_incr.sh)"
  bash scripts/e2e_incr.sh config3 20000 128 2>&1 | cut -c1-260; bash scripts/e2e_incr.sh config5 20000 128 2>&1 | cut -c1-260
  echo "## one rank's share of an N-GPU run (scripts/shard_curve.py config3)"; python3 scripts/shard_curve.py config3 2>&1 | grep alignments
  echo "## one rank's share of 8 (scripts/shard_probe.py 8 8192)"; python3 scripts/shard_probe.py 8 8192 2>&1 | grep shard
  echo "## config 5, one GPU (scripts/scan_probe.py config5)"; python3 scripts/scan_probe.py config5 "" "GFAL_SCAN=1" 2>&1 | grep -v amdgpu.ids
} > gpurun_out/workloads_${tag}.txt 2>&1
say bench; python3 bench.py --steps 20 --warmup 3 > gpurun_out/bench_${tag}_full.json 2> gpurun_out/bench_${tag}_full.err
tail -c 600 gpurun_out/bench_${tag}_full.json
