#!/bin/bash
# Everything DESIGN.md / README quote for a round, collected on ONE GPU box in one call and written under gpurun_out/<tag>/final/ with
# the names profiles/ uses (copy them there and commit):   gpurun --timeout 2400 -- 'bash tools/evidence.sh r04'
#   <tag>_bench.json                        python bench.py --steps 200 --warmup 5 (the driver's line incl. extras and cpu_baseline)
#   <tag>_kernel_stats.{txt,json}           rocprofv3 --kernel-trace --stats, bench --chunks 1 (single stream: true kernel durations)
#   <tag>_kernel_stats_chunked.{txt,json}   the same with the headline's four instance ranges (durations under overlap; launch counts)
#   <tag>_kernel_stats_b512.{txt,json}      the same at 512 instances (k_ric_bwd4 row)
#   <tag>_kernel_stats_full_tick / _config4_share / _rollout
#   <tag>_full_tick_breakdown.json, <tag>_occupancy_sweep.json, <tag>_wbc_eps_sensitivity.json
#   <tag>_pmc.{txt,json} + pmc_latest.json  counter passes (separate runs, no trace domains next to --pmc)
#   <tag>_lq_phase_pmc.txt                  k_lq phase by phase (ablation build)
#   <tag>_timeline_b512.txt                 8 ms of the kernel trace of configs[3]'s share (512 instances, two ranges), per queue
set +e
tag=${1:-r04}
R=$PWD; out=$R/gpurun_out/$tag/final; mkdir -p $out
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > $out/device.txt 2>&1
prof() {  # name, command...
  name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof_$name -o run -- "$@" > $out/prof_$name.log 2>&1)
  db=$(find $out/prof_$name -name "*results.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db $out/${tag}_kernel_stats$([ $name = main ] || echo _$name).txt > /dev/null 2>&1
  rm -rf $out/prof_$name
}
(timeout 900 python bench.py --steps 200 --warmup 5 > $out/${tag}_bench.json) 2> $out/bench.err
prof main python $R/bench.py --steps 20 --warmup 3 --chunks 1 --no-extras --no-cpu-baseline
prof chunked python $R/bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline
prof b512 python $R/bench.py --batch 512 --steps 40 --warmup 3 --chunks 1 --no-extras --no-cpu-baseline
prof full_tick python $R/tools/bench_tick.py --steps 10
# timeline of configs[3]'s per-GPU share (512 instances, per-instance commands, two free-running ranges): which kernels are co-resident
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $out/prof_tl512 -o run -- python $R/bench.py --batch 512 --random-cmd --steps 60 --warmup 5 --no-extras --no-cpu-baseline > $out/prof_tl512.log 2>&1)
db=$(find $out/prof_tl512 -name "*results.db" | head -1)
[ -n "$db" ] && python tools/rocprof_timeline.py $db "" 8.0 > $out/${tag}_timeline_b512.txt 2>> $out/prof_tl512.log
rm -rf $out/prof_tl512
prof config4_share python $R/bench.py --hierarchical --batch 1024 --nodes 200 --steps 20 --warmup 3 --chunks 1 --no-extras --no-cpu-baseline
prof rollout python $R/tools/bench_rollout.py
(timeout 600 python tools/bench_tick.py > $out/${tag}_full_tick_breakdown.json) 2> $out/tick.err
(timeout 900 python tools/occupancy_sweep.py > $out/${tag}_occupancy_sweep.json) 2> $out/sweep.err
(timeout 300 python tools/wbc_eps_sensitivity.py > $out/${tag}_wbc_eps_sensitivity.json) 2> $out/eps.err
# counters
bash tools/gpu_pmc.sh ${tag}pmc > $out/pmc.log 2>&1
cp gpurun_out/${tag}pmc/pmc.txt $out/${tag}_pmc.txt 2>/dev/null; cp gpurun_out/${tag}pmc/pmc.json $out/${tag}_pmc.json 2>/dev/null
# the files name the round, not the scratch directories they were collected in
sed -i "s/\"tag\": \"[^\"]*\"/\"tag\": \"$tag\"/" $out/${tag}_pmc.json 2>/dev/null
sed -i "1s/(\([^)]*\))/($tag)/" $out/${tag}_pmc.txt 2>/dev/null
sed -i "1s#of .*/prof_\([a-z0-9_]*\)/.*#of gpurun_out/$tag/prof_\1/run_results.db#" $out/${tag}_kernel_stats*.txt 2>/dev/null
[ -f variants/libhunter_hip_ablate.so ] && bash tools/lq_phase_pmc.sh $tag > $out/lqpmc.log 2>&1 && cp gpurun_out/$tag/lqpmc/summary.txt $out/${tag}_lq_phase_pmc.txt
ls -la $out
head -c 900 $out/${tag}_bench.json
