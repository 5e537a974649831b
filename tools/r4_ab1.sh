#!/bin/bash
# round 4, batch 1 (GPU box): register-diet packet kernel against the round-3 library; c3 whole-frame parity; PMC of the new build
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r4_ab1; mkdir -p $OUT
AB_VARIANTS="r03 base w3" bash tools/ab_bench.sh 2>&1 | tee $OUT/ab.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "c3_full_size or packet_walk or fused_welford or passes_per_call" 2>&1 | tail -5 | tee $OUT/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --pmc-keep $OUT > $OUT/bench.json 2> $OUT/bench.err; cut -c1-2500 $OUT/bench.json; tail -2 $OUT/bench.err
