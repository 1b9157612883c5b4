#!/bin/bash
# One GPU-box call: tests, bench, kernel-trace profile (and optional extras).  usage: tools/gpu_call.sh <tag> [extras...]
set +e
tag=${1:-r02}; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > $out/device.txt 2>&1
(time timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 --deselect tests/test_closed_loop.py 2>&1 | tail -40) > $out/pytest.log 2>&1
for extra in "$@"; do
  case $extra in
    closed) (time timeout 900 python -m pytest tests/test_closed_loop.py -m gpu -q 2>&1 | tail -30) > $out/pytest_closed.log 2>&1 ;;
    bench) (timeout 900 python bench.py --steps 200 --warmup 5 > $out/bench.json) 2> $out/bench.err ;;
    benchquick) (timeout 600 python bench.py --steps 100 --warmup 5 --no-extras --no-cpu-baseline > $out/bench_quick.json) 2> $out/bench_quick.err ;;
    prof) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$out/prof -o run -- python $OLDPWD/bench.py --steps 20 --warmup 3 --chunks 1 --no-extras --no-cpu-baseline > $OLDPWD/$out/prof_bench.json 2> $OLDPWD/$out/prof.err) ;;
    pcs) (cd /tmp && timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval 100 --kernel-trace -d $OLDPWD/$out/pcs -o run -- python $OLDPWD/bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OLDPWD/$out/pcs_bench.json 2> $OLDPWD/$out/pcs.err; ls -la $OLDPWD/$out/pcs >> $OLDPWD/$out/pcs.err 2>&1) ;;
    sweep) (timeout 900 python tools/occupancy_sweep.py > $out/occupancy_sweep.json) 2> $out/sweep.err ;;
    smoke) (timeout 300 python __graft_entry__.py smoke > $out/smoke.log) 2>&1 ;;
    proftick) R=$PWD; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$out/prof_tick -o run -- python $R/tools/bench_tick.py --steps 10 > $R/$out/prof_tick.log 2>&1); db=$(find $out/prof_tick -name "*results.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db $out/kernel_stats_full_tick.txt > /dev/null 2>&1; rm -rf $out/prof_tick ;;
    profhwbc) R=$PWD; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$out/prof_hwbc -o run -- python $R/bench.py --hierarchical --batch 1024 --nodes 200 --steps 20 --warmup 3 --chunks 1 --no-extras --no-cpu-baseline > $R/$out/prof_hwbc.json 2> $R/$out/prof_hwbc.err); db=$(find $out/prof_hwbc -name "*results.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db $out/kernel_stats_config4_share.txt > /dev/null 2>&1; rm -rf $out/prof_hwbc ;;
    profrollout) R=$PWD; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$out/prof_roll -o run -- python $R/tools/bench_rollout.py > $R/$out/prof_rollout.log 2>&1); db=$(find $out/prof_roll -name "*results.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db $out/kernel_stats_rollout.txt > /dev/null 2>&1; rm -rf $out/prof_roll ;;
    tick) (timeout 600 python tools/bench_tick.py > $out/tick.log) 2>&1 ;;
  esac
done
db=$(find $out/prof -name "*results.db" 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db $out/kernel_stats.txt > /dev/null 2>&1; rm -rf $out/prof/*/*.db 2>/dev/null
ls -la $out
tail -5 $out/pytest.log
head -c 1500 $out/bench.json 2>/dev/null
