#!/usr/bin/env python3
"""Kernel time of configs 1 and 5 at their own sizes (config 1: three spheres, 400x225x100; config 5: the Cornell box of
"The Rest Of Your Life" with light sampling, 600x600x1000)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import abi, lib  # noqa: E402
from tests.conftest import CAM_C1, gold  # noqa: E402
ctx = lib.Context(0)
for name, scene, cam, (W, H, spp), flags in (
        ("config 1", gold("scene_c1.rtsc"), lib.camera(*CAM_C1), (400, 225, 100), 0),
        ("config 5", gold("scene_b3_cornell.rtsc"), lib.camera((278, 278, -800), (278, 278, 0), (0, 1, 0), 40.0, 1.0, 0.0, 10.0, 0.0, 1.0),
         (600, 600, 1000), abi.RT_RENDER_IMPORTANCE)):
    ctx.upload(abi.load_scene(scene))
    best = None
    for rep in range(3):
        _, st = ctx.render(cam, abi.render_params(W, H, spp, 50, seed=1, flags=flags))
        if best is None or st["kernel_ms"] < best["kernel_ms"]:
            best = st
    ms = best["kernel_ms"]
    print(f"{name}: {W}x{H}x{spp}: kernel {ms:.2f} ms, {W * H * spp / ms / 1e3:.0f} Msamples/s, {best['rays'] / ms / 1e3:.0f} Mrays/s")
ctx.close()
