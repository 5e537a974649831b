#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the bench command, outputs under gpurun_out/.
# PMC counters are collected in their own runs (one counter group per run, never combined with sys/hip tracing).
# usage: tools/profile.sh <tag> [c3|c2|c4]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${2:-c3}
KB=$WL; [ "$WL" = c3 ] && KB=c3full; [ "$WL" = c4 ] && KB=c4full    # flat, c2: same name
OUT=$R/gpurun_out/prof_${1:-r01}_$WL
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o k --output-format csv -- $CMD > "$OUT/trace.log" 2>&1
tail -1 "$OUT/trace.log" | cut -c1-300
# PMC on a short un-pipelined run (counters serialise kernels anyway); fewer passes keep the profiling time bounded
PCMD="python $R/tools/kbench.py 4 $KB"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
    tag=$(echo "$grp" | tr ' ' '_' | cut -c1-40)
    RSX_PIPELINE=1 KB_WARM=1 timeout 300 rocprofv3 --pmc $grp -d "$OUT/pmc_$tag" -o k --output-format csv -- $PCMD > "$OUT/pmc_$tag.log" 2>&1
done
ls "$OUT"
