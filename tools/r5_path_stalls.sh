count() { python - "$@" <<'PY'
import sys, subprocess, os
env = dict(os.environ)
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1); env[k] = v
out = subprocess.run([sys.executable, "tools/r5_path_batches_diag.py", sys.argv[1], "72", "0"], env=env, capture_output=True, text=True, timeout=250).stdout
for line in out.splitlines():
    ms = [float(x) for x in line.split(": ")[1].split()][1:]
    slow = [x for x in ms if x > 20]
    print(" ".join(sys.argv[2:]) or "default", line.split(" ms")[0], "calls", len(ms), "median %.1f" % sorted(ms)[len(ms)//2], "stalls", len(slow), ["%.0f" % x for x in slow])
PY
}
count 4
count 4 RSX_PATH_LPT=0
count 4 RSX_PIPELINE=1
count 4 GPU_MAX_HW_QUEUES=4
count 4 RSX_PIPELINE=2
count 1
