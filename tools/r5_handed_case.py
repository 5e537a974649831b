"""round 5 diagnostic (GPU box): the scenes of test_handed_on_paths_give_the_same_frames one after the other in ONE process: tools/r5_handed_case.py glass lambert cornell prism"""
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from source_amd import api as ns, scenes
tab = dict(glass=(scenes.build_glass, scenes.glass_camera, dict(pixels=(96, 80), spp=4)),
           lambert=(scenes.build_lambert, scenes.lambert_camera, dict(pixels=(64, 48), spp=4)),
           cornell=(scenes.build_cornell, scenes.cornell_camera, dict(pixels=(96, 96), spp=4)),
           prism=(scenes.build_prism, scenes.prism_camera, dict(pixels=(96, 64), spp=2, bins=8, spectral_rays=8)))
for which in sys.argv[1:]:
    build, camera, kw = tab[which]
    world = build(ns)[0]
    cam, pipe = camera(ns, world, **kw)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.render_engine = ns.HipEngine(rng="philox", seed=5)
    cam.observe()
    print(which, "first ok", flush=True)
    cam.observe()
    print(which, "ok", pipe.frame.mean.max(), cam.stats["rays"], flush=True)
