#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r06b; mkdir -p $out
(time timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -15) > $out/pytest_all.log 2>&1
tail -6 $out/pytest_all.log
(timeout 900 python bench.py --steps 100 --warmup 5 > $out/bench.json) 2> $out/bench.err
head -c 3000 $out/bench.json; tail -3 $out/bench.err
