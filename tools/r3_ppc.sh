#!/bin/bash
# tools/r3_ppc.sh — configs[1] with K passes per library call: K sweep, per-lane vs packet walk, un-overlapped kernel times
mkdir -p gpurun_out/ppc
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --workload c2 --steps 40 --warmup 5 --no-pmc --no-cpu-baseline --passes-per-call $K 2>gpurun_out/ppc/$tag.err > gpurun_out/ppc/$tag.json
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/ppc/%s.json" % tag).read().strip().splitlines()[-1]); r = d["roofline"]
    print("%-28s %8.4f ms/step %.3e rays/s trace %.4f acc %s" % (tag, d["ms_per_step"], d["value"], r["kernel_ms"], r["accumulate_kernel_ms"]))
except Exception as e:
    print(tag, "failed", e)
PY
}
for K in ${KS:-1 4 8 16 32 64}; do run k$K X=1; done
K=16
run k16_lane RSX_PACKET_MIN_SPP=0
run k16_nopipe RSX_PIPELINE=0
run k16_lane_nopipe RSX_PACKET_MIN_SPP=0 RSX_PIPELINE=0
K=8
run k8_packet RSX_PACKET_MIN_SPP=8
K=16
run k16_fused_nopipe RSX_PIPELINE=0 RSX_FUSE=1
run k16_fused_lane_nopipe RSX_PIPELINE=0 RSX_FUSE=1 RSX_PACKET_MIN_SPP=0
K=64
run k64_fused RSX_FUSE=1
run k64_unfused RSX_FUSE=0
