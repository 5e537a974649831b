#!/bin/bash
# round 5 (GPU box): the level-by-level path passes — parity tests, then one-kernel form vs level form on the path-traced workloads
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r5_wf1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "staged or lambert or dielectric or cornell or prism_scene or importance or furnace or arena_grows or handed" > $OUT/tests.txt 2>&1
tail -15 $OUT/tests.txt
for cfg in cornell lambert; do
  for wf in 0 1; do
    echo "== $cfg RSX_WAVEFRONT=$wf"
    RSX_WAVEFRONT=$wf timeout 300 python tools/kbench.py 6 $cfg 2>&1 | tail -4
  done
done 2>&1 | tee $OUT/kbench.txt
for below in 4096 262144; do echo "== cornell drain below $below"; RSX_WF_DRAIN_BELOW=$below timeout 300 python tools/kbench.py 6 cornell 2>&1 | tail -1; done | tee -a $OUT/kbench.txt
cd /tmp && export TMPDIR=/tmp
RSX_WAVEFRONT=1 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_cornell -o cornell --output-format csv -- python $R/tools/kbench.py 4 cornell > $OUT/prof_cornell.log 2>&1
ls $OUT/prof_cornell
