#!/bin/bash
# round 5 (GPU box): measurement bundle — tools/r5_bundle.sh <tag> "<workloads>" [tests]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
TAG=${1:-r05}; WLS=${2:-c3}
if [ "${3:-}" = tests ]; then
  mkdir -p $R/gpurun_out/${TAG}_extra
  timeout 1800 python -m pytest tests -x -q -m gpu > $R/gpurun_out/${TAG}_extra/tests.txt 2>&1; tail -4 $R/gpurun_out/${TAG}_extra/tests.txt
fi
for wl in $WLS; do
  case $wl in c5) steps="2 1";; c1) steps="10 2";; *) steps="20 3";; esac
  bash tools/profile.sh $TAG $wl $steps
done
