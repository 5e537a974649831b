#!/bin/bash
# round 5 (GPU box): where the waves of the path kernels wait — instruction fetch, LDS, scalar / vector memory levels.
# tools/r5_pmc_groups.sh [kbench config]: counters of one pass, summed per kernel name, for both forms of the path passes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-cornell}
OUT=$R/gpurun_out/pmc_groups_$CFG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for wf in 0 1; do
 i=0
 for grp in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES" \
            "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_TC_INST_REQ SQC_TC_STALL" \
            "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" \
            "SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
   i=$((i+1))
   RSX_WAVEFRONT=$wf KB_WARM=1 timeout 300 rocprofv3 --pmc $grp -d "$OUT/wf${wf}_g$i" -o k --output-format csv -- python $R/tools/kbench.py 1 $CFG > "$OUT/wf${wf}_g$i.log" 2>&1
 done
done
python3 - <<PY
import csv, glob, collections
for wf in (0, 1):
    tot = collections.defaultdict(collections.Counter)
    for f in glob.glob("$OUT/wf%d_g*/*counter_collection.csv" % wf):
        rows = list(csv.DictReader(open(f)))
        ids = sorted({int(r['Dispatch_Id']) for r in rows if 'k_accumulate' in r['Kernel_Name']})
        lo = ids[-2] if len(ids) > 1 else 0                    # the last pass
        for r in rows:
            if int(r['Dispatch_Id']) <= lo or 'rocclr' in r['Kernel_Name']: continue
            tot[r['Kernel_Name'][:40]][r['Counter_Name']] += float(r['Counter_Value'])
    print("RSX_WAVEFRONT=%d" % wf)
    for name, c in sorted(tot.items()):
        if 'accumulate' in name or 'to_queue' in name: continue
        print(" ", name)
        for k in sorted(c): print("    %-28s %.5g" % (k, c[k]))
        if c.get("SQ_WAVE_CYCLES"):
            w = c["SQ_WAVE_CYCLES"]
            print("    -> per wave-cycle: waiting for an instruction %.3f (LDS %.3f); instruction fetches in flight %.3f; LDS level %.3f, scalar-memory level %.3f, vector-memory level %.3f" % (
                c["SQ_WAIT_INST_ANY"] / w, c["SQ_WAIT_INST_LDS"] / w, c["SQ_IFETCH_LEVEL"] / w, c["SQ_INST_LEVEL_LDS"] / w, c["SQ_INST_LEVEL_SMEM"] / w, c["SQ_INST_LEVEL_VMEM"] / w))
        if c.get("SQC_ICACHE_REQ"):
            print("    -> instruction cache hit rate %.4f (misses %.4g + duplicate %.4g of %.4g requests)" % (c["SQC_ICACHE_HITS"] / c["SQC_ICACHE_REQ"], c["SQC_ICACHE_MISSES"], c["SQC_ICACHE_MISSES_DUPLICATE"], c["SQC_ICACHE_REQ"]))
PY
