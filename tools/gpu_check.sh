#!/bin/bash
# GPU box: the round's standard check — GPU test suite, default bench line (with PMC children), one-rank distributed paths.
# usage: tools/gpu_check.sh <tag> [pytest-args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-chk}; shift
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q "$@" > "$OUT/pytest.log" 2>&1; echo "pytest rc $?" | tee -a "$OUT/pytest.log"; tail -5 "$OUT/pytest.log"
timeout 600 python bench.py --steps 20 --warmup 3 --pmc-keep "$OUT" > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc $?"; cut -c1-1500 "$OUT/bench.json"; tail -3 "$OUT/bench.err"
for sh in tile sample; do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --no-pmc --sharding $sh > "$OUT/dist_$sh.json" 2> "$OUT/dist_$sh.err"; echo "dist $sh rc $?"; cut -c1-600 "$OUT/dist_$sh.json"; tail -3 "$OUT/dist_$sh.err"
done
