#!/usr/bin/env python3
"""Runs README.md's path-tracing example (smaller frame) on the GPU box: python tools/readme_example_paths.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from source_amd import api as rs, scenes  # noqa: E402

world, _ = scenes.build_cornell(rs)
rgb = rs.RGBPipeline2D()
cam, _ = scenes.cornell_camera(rs, world, (96, 96), spp=16, bins=15, pipelines=[rgb])
cam.frame_sampler = rs.RGBAdaptiveSampler2D(rgb, ratio=10, fraction=0.2, min_samples=64, cutoff=0.05)
cam.render_engine = rs.MulticoreEngine()
passes = 0
while not cam.render_complete and passes < 40:
    cam.observe()
    passes += 1
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "cornell.png")
os.makedirs(os.path.dirname(out), exist_ok=True)
rgb.save(out)
print("passes", passes, "complete", cam.render_complete, "samples min/max", int(rgb.xyz_frame.samples.min()), int(rgb.xyz_frame.samples.max()), "->", out)
