#!/bin/bash
# round 5 (GPU box): level form after prefetch / one atomic per chunk / important list — tests, A/B, counters
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r5_wf2; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "staged or cornell or prism_scene or importance" > $OUT/tests.txt 2>&1
tail -5 $OUT/tests.txt
for cfg in cornell lambert; do
  for wf in 0 1; do
    echo "== $cfg RSX_WAVEFRONT=$wf"
    RSX_WAVEFRONT=$wf timeout 300 python tools/kbench.py 6 $cfg 2>&1 | tail -1
  done
done 2>&1 | tee $OUT/kbench.txt
echo "== cornell wf3 variant"; RSX_LIB=$R/source_amd/lib/variants/librsx_wf3.so timeout 300 python tools/kbench.py 6 cornell 2>&1 | tail -1 | tee -a $OUT/kbench.txt
bash tools/r5_pmc_levels.sh cornell 2>&1 | awk 'NR<=14 || /per kernel/ || /^  /' 
