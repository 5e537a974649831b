#!/bin/bash
# round 4, experiments (GPU box): path kernel at three waves per SIMD (c1), redo-grid size among overlapping slices (c5), host-material rates
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r4_exp2; mkdir -p $OUT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for rep in 1 2; do for v in base pw3; do lib=$R/source_amd/lib/variants/librsx_$v.so; [ $v = base ] && lib=$R/source_amd/lib/librsx.so
  echo -n "c1 $v: "; RSX_LIB=$lib timeout 300 python bench.py --workload c1 --no-pmc --no-cpu-baseline 2>/dev/null | q; done; done 2>&1 | tee $OUT/c1_pw3.txt
for g in 100000 32 8; do echo -n "c5 redo grid $g: "; RSX_REDO_GRID=$g timeout 400 python bench.py --workload c5 --steps 3 --warmup 1 --no-pmc --no-cpu-baseline 2>/dev/null | q; done 2>&1 | tee $OUT/c5_redo.txt
echo -n "c5 pw3: "; RSX_LIB=$R/source_amd/lib/variants/librsx_pw3.so timeout 400 python bench.py --workload c5 --steps 3 --warmup 1 --no-pmc --no-cpu-baseline 2>/dev/null | q | tee -a $OUT/c5_redo.txt
timeout 900 python tools/host_material_rate.py 128 2 > $OUT/host_materials.txt 2>&1; head -8 $OUT/host_materials.txt; grep -A 30 'Ordered by' $OUT/host_materials.txt | head -45
