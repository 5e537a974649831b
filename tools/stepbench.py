#!/usr/bin/env python3
"""Wall-clock per pass of configs[1] (what bench.py's `value` is made of) for the library in $RSX_LIB / pipeline depth in $RSX_PIPELINE."""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from source_amd import api as ns, scenes
from source_amd.device import get_context
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
world = scenes.build_c2(ns, n=132)[0]
cam, pipe = scenes.c2_camera(ns, world, (1024, 1024), spp=1, bins=15)
cam.frame_sampler = ns.RectFrameSampler2D()
eng = ns.HipEngine(rng="philox", seed=20250905, timing=False)
cam.render_engine = eng
ctx = get_context()
world.build_accelerator()
k = 0
for _ in range(int(os.environ.get("RSX_WARM", "6"))):
    eng.sample_offset = k; k += 1; cam.observe()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    eng.sample_offset = k; k += 1; cam.observe()
ctx.synchronize()
dt = (time.perf_counter() - t0) / steps
try:
    tr, ac = ctx.render_history(min(steps, 64))
except Exception:
    tr, ac = [0.0], [0.0]
digest = hashlib.sha256(np.ascontiguousarray(pipe.frame.mean).tobytes()).hexdigest()[:16]
print(json.dumps({"pipeline": os.environ.get("RSX_PIPELINE", "2"), "ms_per_step": round(dt * 1e3, 4), "Mrays_per_s": round(1.048576 / dt / 1e3, 1),
                  "trace_ms": round(float(np.mean(tr)), 4), "accum_ms": round(float(np.mean(ac)), 4), "digest": digest}))

if os.environ.get("RSX_TIMELINE"):
    from source_amd import _lib
    n = steps + 6
    t = np.zeros((n, 4), dtype=np.float32)
    _lib.check(_lib.lib().rsx_render_timeline(ctx.handle, n, _lib.ptr(t)))
    ends = t[:, 3]
    print("  timeline of %d calls: span %.2f ms" % (n, ends.max()))
    prev = 0.0
    for i in range(n):
        gap = t[i, 0] - (t[i - 2, 1] if i >= 2 else 0.0)
        flag = " <== stall" if (i >= 1 and t[i, 3] - t[i - 1, 3] > 1.5) else ""
        if flag or i < 8 or i >= n - 2:
            print("  call %2d: trace %.3f..%.3f  merge %.3f..%.3f  (merge end delta %.3f)%s" % (i, *t[i], t[i, 3] - (t[i - 1, 3] if i else 0), flag))
