#!/usr/bin/env python3
"""Kernel time of config 2 (1200x800, depth 50) at small sample counts: what a launch costs before its samples do."""
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
ctx.render(cam, abi.render_params(1200, 800, 250, 50, seed=1))
for spp in (1, 2, 4, 8, 16, 31, 32, 48, 63, 64, 96, 125):
    best = None
    for rep in range(3):
        _, st = ctx.render(cam, abi.render_params(1200, 800, spp, 50, seed=1))
        best = st["kernel_ms"] if best is None else min(best, st["kernel_ms"])
    print(f"{spp:4d} spp: kernel {best:7.3f} ms  per sample {best / spp:7.4f} ms  {1200 * 800 * spp / best / 1e3:7.1f} Msamples/s", flush=True)
ctx.close()
