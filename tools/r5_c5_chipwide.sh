#!/bin/bash
# round 5 (GPU box): how busy is the vector ALU over a WHOLE configs[4] pass (512 overlapping slice launches)? The per-launch counters
# of the profile bundle give each launch's share of a chip that ~6 launches divide; this sums SQ_ACTIVE_INST_VALU over every dispatch of
# one pass (counters serialise the kernels, so the pass's wall time comes from an un-profiled run of the same command).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/c5_chipwide; rm -rf $OUT; mkdir -p $OUT
timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 1200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT/pmc -o k --output-format csv -- python $R/bench.py --child --workload c5 --steps 1 > $OUT/pmc.log 2>&1
python3 - <<PY
import csv, glob, json, collections
d = json.loads(open("$OUT/bench.json").read())
secs = d["ms_per_step"] * 1e-3
tot = collections.Counter(); per = collections.defaultdict(collections.Counter); n = collections.Counter()
for f in glob.glob("$OUT/pmc/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'rocclr' in r['Kernel_Name']: continue
        tot[r['Counter_Name']] += float(r['Counter_Value']); per[r['Kernel_Name'][:48]][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_INSTS_VALU': n[r['Kernel_Name'][:48]] += 1
simd_cycles = secs * 2.4e9 * 1024
print("configs[4] pass: %.3f s un-profiled; vector instructions %.4g, active VALU cycles x 4 = %.4g of %.4g SIMD-cycles at 2.4 GHz: chip-wide vector ALU busy %.3f, lane utilisation %.3f" % (
    secs, tot["SQ_INSTS_VALU"], 4 * tot["SQ_ACTIVE_INST_VALU"], simd_cycles, 4 * tot["SQ_ACTIVE_INST_VALU"] / simd_cycles, tot["SQ_THREAD_CYCLES_VALU"] / max(1.0, 64 * tot["SQ_ACTIVE_INST_VALU"])))
for k, c in sorted(per.items(), key=lambda kv: -kv[1]["SQ_ACTIVE_INST_VALU"])[:6]:
    print("  %-48s launches %5d  share of active VALU cycles %.3f  lanes %.3f" % (k, n[k], c["SQ_ACTIVE_INST_VALU"] / tot["SQ_ACTIVE_INST_VALU"], c["SQ_THREAD_CYCLES_VALU"] / max(1.0, 64 * c["SQ_ACTIVE_INST_VALU"])))
PY
