#!/bin/bash
export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/r06tl; mkdir -p $out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $out/prof_tl -o run -- python $R/bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline > $out/prof_tl.log 2>&1)
db=$(find $out/prof_tl -name "*results.db" | head -1)
python tools/rocprof_timeline.py $db "" 16.0 > $out/r06_timeline_headline.txt
rm -rf $out/prof_tl
grep -E "k_lq_trip|k_ric_bwd|k_ric_fwd|k_wbc|k_ls_eval" $out/r06_timeline_headline.txt | head -70
