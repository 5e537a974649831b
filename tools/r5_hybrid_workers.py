#!/usr/bin/env python3
"""Worker processes of the host-callback path (hybrid.run_block) on the GPU box: the `user` scene of tools/host_material_rate.py after a
`host` run in the same process (the parent's heap is what the forks inherit), with run_block's statistics per piece.
usage: python tools/r5_hybrid_workers.py [pixels] [spp] [order, e.g. host,user,user]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from source_amd import api as ns, scenes                    # noqa: E402
from source_amd.device import get_context                   # noqa: E402
from source_amd.optical import hybrid                       # noqa: E402
from source_amd.optical.material import hemisphere_cosine_pdf   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SPP = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ORDER = (sys.argv[3] if len(sys.argv) > 3 else "host,user,user").split(",")


class MyLambert(ns.Lambert):
    def evaluate_shading(self, world, ray, s_in, s_out, w_refl, w_trans, back_face, w2s, s2w, intersection):
        pdf = hemisphere_cosine_pdf(s_out)
        if pdf == 0.0:
            return ray.new_spectrum()
        spectrum = ray.spawn_daughter(w_refl, s_out.transform(s2w)).trace(world)
        spectrum.mul_array(self.reflectivity.sample(spectrum.min_wavelength, spectrum.max_wavelength, spectrum.bins))
        spectrum.mul_scalar(pdf)
        return spectrum


def rss():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"):
            return line.split()[1] + " kB"


print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| cores", hybrid.usable_cores())
for kind in ORDER:
    world, prims = scenes.build_cornell(ns)
    if kind.startswith("user"):
        for p in prims:
            if isinstance(p.material, ns.Lambert):
                p.material = MyLambert(p.material.reflectivity)
    cam, pipe = scenes.cornell_camera(ns, world, (N, N), spp=SPP, bins=15)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.render_engine = ns.HipEngine(rng="philox", seed=5, host_materials=(kind == "host"), host_workers=1 if kind == "user1" else None)
    world.build_accelerator()
    del hybrid.last_stats[:]
    t0 = time.perf_counter()
    cam.observe()
    get_context().synchronize()
    dt = time.perf_counter() - t0
    print("%-6s %dx%d x %d: %.3f s, %.4g primary rays/s, rss %s" % (kind, N, N, SPP, dt, N * N * SPP / dt, rss()), flush=True)
    for st in hybrid.last_stats:
        print("   ", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items() if k not in ("worker_wait_s",)})
