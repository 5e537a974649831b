#!/bin/bash
# round 6 (GPU box): the kept evidence for the stall fix — the fault brought back by environment, then the library as it is
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6_stalls_final; rm -rf $O; mkdir -p $O
for spec in "RSX_HOST_THREADS=256 RSX_HOST_SPIN=1 world" "RSX_HOST_THREADS=256 RSX_HOST_SPIN=1 scene" "RSX_HOST_THREADS=256 RSX_HOST_SPIN=1 none" "A=1 world" "A=1 scene" "A=1 frame"; do
  set -- $spec
  if [ $# = 3 ]; then envs="$1 $2"; trig=$3; else envs="$1"; trig=$2; fi
  echo "=== env $envs, trigger: $trig" | tee -a $O/log.txt
  env $envs timeout 200 python tools/r6_world_stalls.py 60 cornell $trig 2>&1 | tee -a $O/log.txt
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "three_worlds" 2>&1 | tail -3 | tee -a $O/log.txt
