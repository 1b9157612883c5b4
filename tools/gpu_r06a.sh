#!/bin/bash
# round-6 A/B: trip kernel vs the one-node kernel, trip lengths, machine-LICM on / off
export TMPDIR=/tmp
out=gpurun_out/r06a; mkdir -p $out
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_failure_surface.py tests/test_gpu_timed_path.py -m gpu -q -x 2>&1 | tail -15) > $out/pytest.log 2>&1
tail -5 $out/pytest.log
(
python tools/perf_quick.py --lib variants/lib_base.so --steps 30
python tools/perf_quick.py --steps 30
python tools/perf_quick.py --steps 30 --reserved 129
python tools/perf_quick.py --steps 30 --reserved 123
python tools/perf_quick.py --steps 30 --reserved 122
python tools/perf_quick.py --steps 30 --reserved 121
python tools/perf_quick.py --steps 30 --chunks 4
python tools/perf_quick.py --lib variants/lib_base.so --steps 30 --chunks 4
python tools/perf_quick.py --steps 60 --batch 512
python tools/perf_quick.py --lib variants/lib_base.so --steps 60 --batch 512
) > $out/perf.log 2>&1
cat $out/perf.log
