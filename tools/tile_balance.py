import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from source_amd import api as ns, scenes
from source_amd import distributed as D
from source_amd.device import get_context
world = scenes.build_c3(ns, n=132)[0]
cam, pipe = scenes.c3_camera(ns, world, (2048, 2048), spp=64, bins=15)
eng = ns.HipEngine(rng="philox", seed=1)
cam.render_engine = eng
ctx = get_context()
world.build_accelerator()
for N in (2, 4, 8):
    ts = []
    for r in range(N):
        cam.frame_sampler = ns.RectFrameSampler2D(rect=D.tile_rect(r, N, 2048, 2048))
        for k in range(3):
            cam.observe()
        ctx.synchronize()
        t0 = time.perf_counter()
        for k in range(5):
            cam.observe()
        ctx.synchronize()
        ts.append((time.perf_counter() - t0) / 5 * 1e3)
    print(N, ["%.2f" % t for t in ts], "max %.2f  mean %.2f  balance %.2f" % (max(ts), sum(ts) / N, sum(ts) / N / max(ts)))
print("interleaved strips")
for N, W in ((8, 128), (8, 64), (8, 32), (4, 128), (4, 64), (2, 128)):
    ts = []
    for r in range(N):
        strips = [(x0, 0, min(x0 + W, 2048), 2048) for x0 in range(r * W, 2048, N * W)]
        def step():
            for s in strips:
                cam.frame_sampler = ns.RectFrameSampler2D(rect=s)
                cam.observe()
        for k in range(3):
            step()
        ctx.synchronize()
        t0 = time.perf_counter()
        for k in range(5):
            step()
        ctx.synchronize()
        ts.append((time.perf_counter() - t0) / 5 * 1e3)
    print(N, W, ["%.2f" % t for t in ts], "max %.2f  mean %.2f  balance %.2f" % (max(ts), sum(ts) / N, sum(ts) / N / max(ts)))
print("cost-balanced contiguous tiles (bench.py --tiles balanced)")
bw = 64
cost = []
for x0 in range(0, 2048, bw):
    cam.frame_sampler = ns.RectFrameSampler2D(rect=(x0, 0, x0 + bw, 2048))
    cam.observe(); ctx.synchronize()
    t0 = time.perf_counter()
    for k in range(4): cam.observe()
    ctx.synchronize()
    cost.append((time.perf_counter() - t0) / 4)
for N in (2, 4, 8):
    b = D.balanced_bounds(cost, bw, 2048, N)
    ts = []
    for r in range(N):
        cam.frame_sampler = ns.RectFrameSampler2D(rect=D.tile_rect(r, N, 2048, 2048, b))
        for k in range(3): cam.observe()
        ctx.synchronize()
        t0 = time.perf_counter()
        for k in range(5): cam.observe()
        ctx.synchronize()
        ts.append((time.perf_counter() - t0) / 5 * 1e3)
    print(N, b, ["%.2f" % t for t in ts], "max %.2f  mean %.2f  balance %.2f" % (max(ts), sum(ts) / N, sum(ts) / N / max(ts)))
print("tiles rebalanced by measured tile times (distributed.rebalance_bounds)")
for N in (4, 8):
    b = [(2048 * r) // N for r in range(N)] + [2048]
    for it in range(5):
        ts = []
        for r in range(N):
            cam.frame_sampler = ns.RectFrameSampler2D(rect=D.tile_rect(r, N, 2048, 2048, b))
            cam.observe(); ctx.synchronize()
            t0 = time.perf_counter()
            for k in range(3): cam.observe()
            ctx.synchronize()
            ts.append((time.perf_counter() - t0) / 3 * 1e3)
        print(N, it, b, "max %.2f  mean %.2f  balance %.2f" % (max(ts), sum(ts) / N, sum(ts) / N / max(ts)))
        b = D.rebalance_bounds(b, ts, 2048)
