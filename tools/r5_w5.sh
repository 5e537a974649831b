#!/bin/bash
# round 5 (GPU box): five waves per SIMD for the packet kernel (variant library w5: 96 registers, 11 mesh stack levels in LDS) against the base library, same box
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s8
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forked_workers" ) > gpurun_out/s8/forked.txt 2>&1; tail -4 gpurun_out/s8/forked.txt
for v in base w5 base w5; do
  lib=$R/source_amd/lib/variants/librsx_$v.so; [ $v = base ] && lib=$R/source_amd/lib/librsx.so
  for w in c3 flat c2k; do
    RSX_LIB=$lib timeout 300 python bench.py --workload $w --no-cpu-baseline --no-pmc > gpurun_out/s8/${v}_$w.json 2> gpurun_out/s8/${v}_$w.err
    python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/s8/${v}_$w.json").read().strip().splitlines()[-1]); print("$v $w", d["ms_per_step"], "%.4g" % d["value"])
except Exception as e: print("$v $w ERR", e, open("gpurun_out/s8/${v}_$w.err").read()[-300:])
PY
  done
done
RSX_LIB=$R/source_amd/lib/variants/librsx_w5.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c3_full_size or fused_welford or frames_instanced or packet_walk" 2>&1 | tail -2
