#!/bin/bash
# A/B on the GPU box: single-leaf primitives in the free wide slots (default) against RSX_NO_WIDE_ALL=1: tools/r4_wideall.sh <workload> <steps>
for round in 1 2; do
for v in 0 1; do
  if [ $v = 1 ]; then export RSX_NO_WIDE_ALL=1; else unset RSX_NO_WIDE_ALL; fi
  echo -n "RSX_NO_WIDE_ALL=$v: "
  timeout 600 python bench.py --workload ${1:-c1} --steps ${2:-10} --warmup 2 --no-pmc --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
done
