#!/bin/bash
# round 6 (GPU box): what triggers the ~80 ms stalls after a second world is built: tools/r6_stalls.sh "ENV=.. trigger" ...
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6_stalls4; mkdir -p $O
for spec in "$@"; do
  set -- $spec
  echo "=== $1 $2" | tee -a $O/log.txt
  env $1 timeout 200 python tools/r6_world_stalls.py 60 cornell $2 2>&1 | grep -v "^    call 0 \|^kfd proc\|^total evicted" | tee -a $O/log.txt
done
