#!/bin/bash
export TMPDIR=/tmp
for b in 1 2 4 8; do
  timeout 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --traffic none --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$b', d['value'], d['ms_per_step'])"
done
