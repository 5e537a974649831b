#!/bin/bash
# round 6 (GPU box): where the waves of the PACKET kernel wait — tools/r6_pmc_packet.sh [kbench config] [lib]; counters per launch of k_render_trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-c3full}
[ -n "$2" ] && export RSX_LIB=$R/source_amd/lib/variants/librsx_$2.so
OUT=$R/gpurun_out/pmc_packet_$CFG${2:+_$2}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_TC_INST_REQ SQC_TC_STALL" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" \
           "SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAVES"; do
  i=$((i+1))
  KB_WARM=1 timeout 300 rocprofv3 --pmc $grp -d "$OUT/g$i" -o k --output-format csv -- python $R/tools/kbench.py 2 $CFG > "$OUT/g$i.log" 2>&1
done
python3 - <<PY | tee "$OUT/summary.txt"
import csv, glob, collections
tot = collections.defaultdict(collections.Counter); cnt = collections.Counter()
for f in glob.glob("$OUT/g*/*counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    seen = set()
    for r in rows:
        if 'k_render_trace' not in r['Kernel_Name']: continue
        tot[r['Kernel_Name'][:60]][r['Counter_Name']] += float(r['Counter_Value'])
        seen.add((r['Kernel_Name'][:60], r['Dispatch_Id'], r['Counter_Name']))
    for (k, d, c) in seen: cnt[(k, c)] += 1
for name, c in sorted(tot.items()):
    print(name)
    for k in sorted(c):
        c[k] /= max(1, cnt[(name, k)])
        print("    %-28s %.5g  (per launch, %d launches)" % (k, c[k], cnt[(name, k)]))
    w = c.get("SQ_WAVE_CYCLES")
    if w:
        print("    -> per wave-cycle: waiting for an instruction %.3f (LDS %.3f); instruction fetches in flight %.3f; LDS level %.3f, scalar-memory level %.3f, vector-memory level %.3f" % (
            c["SQ_WAIT_INST_ANY"] / w, c["SQ_WAIT_INST_LDS"] / w, c["SQ_IFETCH_LEVEL"] / w, c["SQ_INST_LEVEL_LDS"] / w, c["SQ_INST_LEVEL_SMEM"] / w, c["SQ_INST_LEVEL_VMEM"] / w))
    if c.get("SQC_ICACHE_REQ"):
        print("    -> instruction cache hit rate %.4f (misses %.4g + duplicate %.4g of %.4g requests); scalar data cache miss rate %.4f of %.4g" % (
            c["SQC_ICACHE_HITS"] / c["SQC_ICACHE_REQ"], c["SQC_ICACHE_MISSES"], c["SQC_ICACHE_MISSES_DUPLICATE"], c["SQC_ICACHE_REQ"], c["SQC_DCACHE_MISSES"] / max(1, c["SQC_DCACHE_REQ"]), c["SQC_DCACHE_REQ"]))
    if c.get("GRBM_GUI_ACTIVE"):
        g = c["GRBM_GUI_ACTIVE"]
        print("    -> VALU busy %.3f, scalar %.3f (of SIMD cycles); VALU insts %.4g, SALU %.4g, SMEM %.4g" % (c["SQ_ACTIVE_INST_VALU"] * 4 / (g * 128), c.get("SQ_ACTIVE_INST_SCA", 0) * 4 / (g * 128), c["SQ_INSTS_VALU"], c["SQ_INSTS_SALU"], c["SQ_INSTS_SMEM"]))
PY
