#!/bin/bash
# Run on the GPU box (gpurun):  bash scripts/quick_prof.sh <tag> [bench args]
# Kernel times (rocprofv3 --kernel-trace --stats) and the SQ instruction / wait
# counters of a short bench.py run, printed per kernel.  Scratch output under
# gpurun_out/qp_<tag>_*.
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
B="python3 bench.py --no-cpu-baseline --no-search-mode $*"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/qp_${tag}_stats -- $B --steps 10 --warmup 2 > gpurun_out/qp_${tag}_stats.log 2>&1
python3 scripts/kstats.py gpurun_out/qp_${tag}_stats
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH --output-format csv -d gpurun_out/qp_${tag}_sq1 -- $B --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/qp_${tag}_sq2 -- $B --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT --output-format csv -d gpurun_out/qp_${tag}_sq3 -- $B --steps 2 --warmup 1 > /dev/null 2>&1
python3 scripts/pmc_kernels.py gpurun_out/qp_${tag}_sq1 gpurun_out/qp_${tag}_sq2 gpurun_out/qp_${tag}_sq3
