#!/bin/bash
# Dynamic instruction counts / wave cycles of k_lq PHASE BY PHASE: one rocprofv3 --pmc pass per ablation stop over the -DHB_ABLATE
# variant (csrc/build.sh --ablate), summarised by tools/lq_phase_pmc.py.   usage: tools/lq_phase_pmc.sh <tag>
set +e
tag=${1:-r03}; out=$PWD/gpurun_out/$tag/lqpmc; mkdir -p $out
stops=${2:-"126 9 1 2 3 30 31 4 5 32 33 34 0"}   # k_lq_trip (round 6); the one-node kernel of rounds 1-5: "10 6 7 9 1 2 3 30 31 4 5 32 33 34 0" with reserved 129
export TMPDIR=/tmp
R=$PWD
cd /tmp
for stop in $stops; do
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $out/s$stop -o p -- \
    python $R/tools/perf_quick.py --lib $R/variants/libhunter_hip_ablate.so --stop $stop > $out/s$stop.log 2>&1
done
cd $R
python tools/lq_phase_pmc.py $out --trip > $out/summary.txt 2>&1
find $out -name "*.db" -delete
cat $out/summary.txt
