mkdir -p gpurun_out/s4
run() { name=$1; shift; env "$@" timeout 300 python bench.py --workload $WL --steps $ST --warmup 1 --no-cpu-baseline --no-pmc $EXTRA > gpurun_out/s4/$name.json 2> gpurun_out/s4/$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/s4/$name.json").read().strip().splitlines()[-1]); print("$name", d["ms_per_step"], "%.4g"%d["value"])
except Exception as e: print("$name ERR", e)
PY
}
WL=c5; ST=2; EXTRA=""
run c5_default A=1
run c5_wg2 RSX_PATH_WG=2
run c5_wg2_l4 RSX_PATH_WG=2 RSX_PATH_LANES=4
run c5_l4 RSX_PATH_LANES=4
run c5_l6 RSX_PATH_LANES=6
WL=c1; ST=6
EXTRA="--passes-per-call 1"; run c1_k1 A=1
EXTRA="--passes-per-call 2"; run c1_k2 A=1
EXTRA="--passes-per-call 4"; run c1_k4 A=1
