#!/bin/bash
# round 5 (GPU box): SQ counters of every kernel launch of ONE path-traced pass, in launch order (the level kernels one by one):
# tools/r5_pmc_levels.sh [kbench config] — VALU busy, lane utilisation, waves waiting, instructions per launch.
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-cornell}
OUT=$R/gpurun_out/pmc_levels_$CFG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
KB_WARM=1 timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE \
    -d "$OUT/sq" -o k --output-format csv -- python $R/tools/kbench.py 1 $CFG > "$OUT/sq.log" 2>&1
KB_WARM=1 timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/f" -o k --output-format csv -- python $R/tools/kbench.py 1 $CFG > "$OUT/f.log" 2>&1
KB_WARM=1 timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d "$OUT/w" -o k --output-format csv -- python $R/tools/kbench.py 1 $CFG > "$OUT/w.log" 2>&1
python3 $R/tools/r5_pmc_levels_summary.py "$OUT"
