cd /tmp && export TMPDIR=/tmp
RSX_EAGER_BATCH=0 timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/c2tl -o k --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 64 --no-pmc --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv,os
rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/c2tl/k_kernel_trace.csv")))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
big=[i for i,r in enumerate(rows) if 'k_render_trace<false, 0, 1, 2, true>' in r['Kernel_Name']]
a=big[-4]
t0=int(rows[a]['Start_Timestamp']); prev=t0
for r in rows[a-3:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print("%-58s start %8.3f dur %6.3f gap %6.3f" % (r['Kernel_Name'][:58], (s-t0)/1e6,(e-s)/1e6,(s-prev)/1e6)); prev=e
PY
