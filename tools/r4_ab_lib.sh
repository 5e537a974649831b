#!/bin/bash
# A/B on the GPU box: tools/r4_ab_lib.sh <workload> <steps> lib1 lib2 ... (names under source_amd/lib/variants/, "tree" = the in-tree library); two rounds each
wl=$1; steps=$2; shift 2
for round in 1 2; do
  for v in "$@"; do
    if [ $v = tree ]; then unset RSX_LIB; else export RSX_LIB=$PWD/source_amd/lib/variants/librsx_$v.so; fi
    echo -n "$v: "
    timeout 900 python bench.py --workload $wl --steps $steps --warmup 2 --no-pmc --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g %s, %.3f ms per step' % (d['value'], d['unit'], d['ms_per_step']))"
  done
done
