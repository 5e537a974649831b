#!/bin/bash
# round 6 (GPU box): HIP-API + kernel + copy timeline of the stall reproducer (trigger: a new device scene over the same world)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6_stalls_trace; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace -d $O/t -o t --output-format csv -- python $R/tools/r6_world_stalls.py 40 cornell scene > $O/run.txt 2>&1
cat $O/run.txt | grep -v "^    call 0 " | tail -20
python3 - <<PY | tee $O/timeline.txt
import csv, glob
def load(pat):
    f = glob.glob("$O/t/**/" + pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
api, ker, cp = load("*hip_api_trace.csv"), load("*kernel_trace.csv"), load("*memory_copy_trace.csv")
print("records: %d HIP calls, %d kernels, %d copies" % (len(api), len(ker), len(cp)))
if api:
    t0 = min(int(r["Start_Timestamp"]) for r in api)
    ev = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "api", r["Function"]) for r in api]
    ev += [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "kernel", r["Kernel_Name"][:60]) for r in ker]
    ev += [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "copy", r.get("Direction", "")) for r in cp]
    ev.sort()
    print("everything that took longer than 15 ms, with the 6 events before it (start ms, duration ms, kind, name):")
    for i, e in enumerate(ev):
        if e[1] > 15e6:
            for p in ev[max(0, i - 6):i]:
                print("      %10.3f %9.3f %-6s %s" % (p[0] / 1e6, p[1] / 1e6, p[2], p[3]))
            print("  >>> %10.3f %9.3f %-6s %s" % (e[0] / 1e6, e[1] / 1e6, e[2], e[3]))
    # device gaps: idle time between consecutive kernels longer than 15 ms
    ks = sorted((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Kernel_Name"][:60]) for r in ker)
    print("gaps on the device longer than 15 ms (end of one kernel -> start of the next):")
    for a, b in zip(ks, ks[1:]):
        if b[0] - a[1] > 15e6:
            print("      %10.3f -> %10.3f  (%.1f ms)  after %s, before %s" % (a[1] / 1e6, b[0] / 1e6, (b[0] - a[1]) / 1e6, a[2], b[2]))
    # HIP calls around scene creation
    names = {}
    for r in api: names[r["Function"]] = names.get(r["Function"], 0) + 1
    print("HIP calls by name:", dict(sorted(names.items(), key=lambda kv: -kv[1])))
PY
