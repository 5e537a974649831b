#!/bin/bash
# round 4 (GPU box): bench.py's multi-rank flow with W ranks sharing the one GPU — librsx's own exchange over the transport stub
# (tests/stub_rccl, RSX_RCCL_LIB) for tile / sample / slice sharding, each verified against a one-GPU render inside the run.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r4_multirank; mkdir -p $OUT
hipcc -shared -fPIC -O2 tests/stub_rccl/rccl_stub.cpp -o $OUT/librccl_stub.so || exit 1
for spec in "2 c3 tile" "4 c3 tile" "2 c3 sample" "2 c5s slice"; do set -- $spec; W=$1; WL=$2; SH=$3
  rm -rf $OUT/stubdir; mkdir -p $OUT/stubdir
  echo "== W=$W workload=$WL sharding=$SH"
  RSX_RCCL_LIB=$OUT/librccl_stub.so RSX_STUB_DIR=$OUT/stubdir RSX_DEVICE=0 CUDA_VISIBLE_DEVICES=0 HIP_VISIBLE_DEVICES=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W \
    --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus $W --steps 3 --warmup 1 --workload $WL --sharding $SH --no-pmc --no-cpu-baseline > $OUT/w${W}_${WL}_$SH.json 2> $OUT/w${W}_${WL}_$SH.err
  echo "rc $?"; python - $OUT/w${W}_${WL}_$SH.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]); c = d["config"]
    print(d["n_gpus"], d["ms_per_step"], "%.4g" % d["value"], c["sharding"], c["collective"], "ranks", c["rccl_ranks"], "digest==1gpu", c["frame_digest_equals_single_gpu"], "merge==1gpu", c["sample_merge_equals_single_gpu"], "tiles", c["tile_bounds"])
except Exception as e:
    print("no line:", e)
PY
  tail -2 $OUT/w${W}_${WL}_$SH.err | cut -c1-300
done
