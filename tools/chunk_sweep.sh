#!/bin/bash
# headline rate as a function of the number of instance ranges (bench.py --chunks); usage: tools/chunk_sweep.sh [batch]
B=${1:-4096}
for c in 1 2 3 4 6 8; do
  python bench.py --batch $B --steps 100 --warmup 5 --chunks $c --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunks', $c, round(j['value']), round(j['ms_per_step'],3))"
done
