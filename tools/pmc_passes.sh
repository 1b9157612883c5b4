#!/bin/bash
# rocprofv3 counter passes (separate runs, --kernel-trace only; never combined with sys/hip/hsa tracing).
# Usage (on the GPU box, from the repo root):  bash tools/pmc_passes.sh <tag>
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_${TAG}_fetch -o p -- $CMD > gpurun_out/pmc_${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_${TAG}_write -o p -- $CMD > gpurun_out/pmc_${TAG}_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc_${TAG}_sq -o p -- $CMD > gpurun_out/pmc_${TAG}_sq.log 2>&1
ls -la gpurun_out/pmc_${TAG}_*
