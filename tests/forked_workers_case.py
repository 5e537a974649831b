"""The body of tests/test_gpu_parity.py::test_python_materials_in_forked_workers_with_the_device_in_the_loop, run as a process of its own
(see there). Prints "forked workers OK" when every comparison held."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from source_amd import api as ns, scenes                    # noqa: E402
from source_amd.optical import hybrid                       # noqa: E402
from source_amd.optical.material import hemisphere_cosine_pdf   # noqa: E402


def eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


class MyLambert(ns.Lambert):
    def evaluate_shading(self, world, ray, s_in, s_out, w_refl, w_trans, back_face, w2s, s2w, intersection):
        pdf = hemisphere_cosine_pdf(s_out)
        if pdf == 0.0:
            return ray.new_spectrum()
        spectrum = ray.spawn_daughter(w_refl, s_out.transform(s2w)).trace(world)
        spectrum.mul_array(self.reflectivity.sample(spectrum.min_wavelength, spectrum.max_wavelength, spectrum.bins))
        spectrum.mul_scalar(pdf)
        return spectrum


hybrid.MIN_RAYS_PER_WORKER = 256


def render(user, **engine):
    world, prims = scenes.build_cornell(ns)
    if user:
        for p in prims:
            if isinstance(p.material, ns.Lambert):
                p.material = MyLambert(p.material.reflectivity)
    cam, pipe = scenes.cornell_camera(ns, world, (40, 36), 2, 5)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.render_engine = ns.HipEngine(rng="philox", seed=17, **engine)
    del hybrid.last_stats[:]
    cam.observe()
    return pipe.frame.mean.copy(), pipe.frame.variance.copy(), cam.stats["rays"], [st["workers"] for st in hybrid.last_stats]


device = render(False)
workers = render(True, host_workers=4)
alone = render(True, host_workers=1)
per_node = render(True, host_workers=3, per_node_materials=True)
assert workers[3] == [4] and alone[3] == [] and per_node[3] == [3], (workers[3], alone[3], per_node[3])
for other in (workers, alone, per_node):
    assert eq(device[0], other[0]) and eq(device[1], other[1]) and device[2] == other[2]
again = render(False)
assert eq(device[0], again[0]) and (device[0] > 0).mean() > 0.3
print("forked workers OK")
