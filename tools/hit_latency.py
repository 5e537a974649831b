"""Latency of one World.hit(ray) / World.contains(point) from Python on the host walk (csrc/rsx_hostwalk.cpp) — no device needed:
python tools/hit_latency.py   (kept output: profiles/r06_host_latency.txt)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from source_amd import api as ns, scenes
from source_amd._flatten import FlatScene
from source_amd.device import HostScene

cases = {"configs[1] mesh world (69k triangles)": (lambda: scenes.build_c2(ns, n=132)[0], (0, 0.16, -0.4), (0, -0.2, 1)),
         "demos/csg.py world (configs[3])": (lambda: scenes.build_csg_demo(ns)[0], (0.3, 0.2, -6.0), (-0.35, 0.3, 1)),
         "demos/prism.py world (configs[4])": (lambda: scenes.build_prism(ns)[0], (0.1, 0.05, -2.0), (0.0, 0.0, 1))}
for name, (build, o, d) in cases.items():
    world = build()
    host = HostScene(world.flatten())
    ray = ns.Ray(ns.Point3D(*o), ns.Vector3D(*d).normalise())
    dd = ray.direction
    for _ in range(200): host.hit_one(o[0], o[1], o[2], dd.x, dd.y, dd.z, float("inf"))
    n = 20000
    t0 = time.perf_counter()
    for _ in range(n): r = host.hit_one(o[0], o[1], o[2], dd.x, dd.y, dd.z, float("inf"))
    dt = (time.perf_counter() - t0) / n
    t1 = time.perf_counter()
    for _ in range(2000): host.contains_batch([o])
    dc = (time.perf_counter() - t1) / 2000
    print("%-40s rsx_hit_host_one through ctypes: %.2f us per ray (hit: prim %s, t %s); contains (numpy call): %.1f us" %
          (name, dt * 1e6, None if r is None else r[0], None if r is None else "%.6f" % r[1], dc * 1e6))
