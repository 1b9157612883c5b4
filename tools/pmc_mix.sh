#!/bin/bash
# Instruction-mix counter passes (separate runs, --kernel-trace only).  Usage on the GPU box: bash tools/pmc_mix.sh <tag>
set -u
TAG=${1:-mix}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES -d gpurun_out/pmc_${TAG}_a -o p -- $CMD > gpurun_out/pmc_${TAG}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE -d gpurun_out/pmc_${TAG}_b -o p -- $CMD > gpurun_out/pmc_${TAG}_b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/pmc_${TAG}_c -o p -- $CMD > gpurun_out/pmc_${TAG}_c.log 2>&1
python tools/pmc_summary.py ${TAG}_mix gpurun_out/pmc_${TAG}_a/p_results.db gpurun_out/pmc_${TAG}_b/p_results.db gpurun_out/pmc_${TAG}_c/p_results.db > gpurun_out/pmc_${TAG}_summary.txt 2>&1
cp profiles/${TAG}_mix_pmc.* gpurun_out/ 2>/dev/null
tail -3 gpurun_out/pmc_${TAG}_a.log
