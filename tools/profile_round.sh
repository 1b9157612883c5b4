#!/bin/bash
# One-call profile capture on the GPU box: rocprofv3 kernel-trace stats of the bench command + the counter passes,
# summarised into profiles/<tag>_* (copied under gpurun_out/ so they travel back).  Usage: bash tools/profile_round.sh r01e
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out profiles
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG} -o ${TAG} -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_${TAG}.log 2>&1
python tools/rocprof_summary.py gpurun_out/prof_${TAG}/${TAG}_results.db profiles/${TAG}_kernel_stats.txt >> gpurun_out/prof_${TAG}.log 2>&1
bash tools/pmc_passes.sh ${TAG} > gpurun_out/pmc_${TAG}.log 2>&1
python tools/pmc_summary.py ${TAG} gpurun_out/pmc_${TAG}_fetch/p_results.db gpurun_out/pmc_${TAG}_write/p_results.db gpurun_out/pmc_${TAG}_sq/p_results.db > gpurun_out/pmc_${TAG}_summary.txt 2>&1
cp profiles/${TAG}_* gpurun_out/
rm -rf gpurun_out/prof_${TAG} gpurun_out/pmc_${TAG}_fetch gpurun_out/pmc_${TAG}_write gpurun_out/pmc_${TAG}_sq
tail -2 gpurun_out/prof_${TAG}.log
