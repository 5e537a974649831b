#!/bin/bash
# round 4 (GPU box): measurement bundle for the given workloads + the stress / latency scripts whose numbers DESIGN.md quotes.
# usage: tools/r4_bundle.sh <tag> "<workloads>" [stress]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
TAG=${1:-r04}; WLS=${2:-c3}
for wl in $WLS; do bash tools/profile.sh $TAG $wl; done
if [ "${3:-}" = stress ]; then
  OUT=$R/gpurun_out/${TAG}_extra; mkdir -p $OUT
  (echo "== tools/stress_parity.py"; timeout 600 python tools/stress_parity.py; echo "== tools/stress_mesh.py 2"; timeout 400 python tools/stress_mesh.py 2;
   echo "== tools/stress_world.py"; timeout 600 python tools/stress_world.py; echo "== tools/stress_csg.py 40 60000 6"; timeout 600 python tools/stress_csg.py 40 60000 6) > $OUT/stress.txt 2>&1
  tail -5 $OUT/stress.txt
  (echo "== tools/hit_latency.py"; timeout 200 python tools/hit_latency.py; echo "== tools/host_prof.py"; timeout 200 python tools/host_prof.py) > $OUT/host_latency.txt 2>&1
  tail -5 $OUT/host_latency.txt
fi
