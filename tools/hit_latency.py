import sys, time
sys.path.insert(0, '/root/repo')
from source_amd import api as ns, scenes
world = scenes.build_c2(ns, n=132)[0]
world.build_accelerator()
ray = ns.Ray(ns.Point3D(0, 0.16, -0.4), ns.Vector3D(0, -0.2, 1).normalise())
for _ in range(20): world.hit(ray)
t0 = time.perf_counter()
for _ in range(300): hit = world.hit(ray)
dt = (time.perf_counter() - t0) / 300
print("World.hit(ray) single-ray latency: %.1f us (hit t=%s)" % (dt * 1e6, hit.ray_distance if hit else None))
