#!/bin/bash
# Runs on the GPU box (via gpurun): the round's measurement bundle of one workload, outputs under gpurun_out/prof_<tag>_<wl>/.
#   1. rocprofv3 --kernel-trace --stats of the bench command (kernel durations as the profiler sees them),
#   2. the bench line itself, un-profiled, with its own PMC child runs (bench.py --pmc-keep: counters in their own rocprofv3 runs,
#      one counter group per run, never combined with a trace option) and the CPU baseline.
# usage: tools/profile.sh <tag> [c3|c2|c4|flat|c1|c5] [steps] [warmup]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${2:-c3}
OUT=$R/gpurun_out/prof_${1:-r02}_$WL
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o k --output-format csv -- python $R/bench.py --workload $WL --steps ${3:-20} --warmup ${4:-3} --no-cpu-baseline --no-pmc > "$OUT/trace.log" 2>&1
tail -1 "$OUT/trace.log" | cut -c1-200
cd $R
timeout 900 python bench.py --workload $WL --steps ${3:-20} --warmup ${4:-3} --pmc-keep "$OUT" > "$OUT/bench.json" 2> "$OUT/bench.err"
cut -c1-300 "$OUT/bench.json"; tail -2 "$OUT/bench.err"
ls "$OUT"
