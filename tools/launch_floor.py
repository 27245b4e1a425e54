#!/usr/bin/env python3
"""What a launch of the sphere kernel costs with next to nothing to do: config 2's world, small frames and depth 1."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import abi, lib  # noqa: E402
from tests.conftest import CAM_C2  # noqa: E402

scene = abi.load_scene(os.path.join(ROOT, "tests", "golden", "scene_c2.rtsc"))
cam = lib.camera(*CAM_C2)
ctx = lib.Context(0)
ctx.upload(scene)
ctx.render(cam, abi.render_params(1200, 800, 100, 50, seed=1))
for (w, h, spp, depth) in ((64, 64, 1, 1), (64, 64, 1, 50), (1200, 800, 1, 1), (1200, 800, 1, 2), (1200, 800, 1, 50), (1200, 800, 4, 1), (1200, 800, 4, 50),
                           (1200, 800, 63, 1), (1200, 800, 63, 3), (1200, 800, 63, 8), (1200, 800, 63, 50)):
    best = None
    for rep in range(4):
        _, st = ctx.render(cam, abi.render_params(w, h, spp, depth, seed=1))
        best = st if best is None or st["kernel_ms"] < best["kernel_ms"] else best
    print(f"{w}x{h}x{spp} depth {depth:2d}: kernel {best['kernel_ms']:7.3f} ms  rays {best['rays']:10d}  {best['rays'] / best['kernel_ms'] / 1e6:6.2f} Grays/s", flush=True)
ctx.close()
