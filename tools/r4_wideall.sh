#!/bin/bash
# A/B on the GPU box: every primitive of a small analytic room answered before the walk (default) against RSX_NO_WIDE_ALL=1
for v in 0 1; do
  if [ $v = 1 ]; then export RSX_NO_WIDE_ALL=1; else unset RSX_NO_WIDE_ALL; fi
  echo "RSX_NO_WIDE_ALL=$v"
  timeout 600 python bench.py --workload c1 --steps 10 --warmup 2 --no-pmc --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('kernels'))"
done
