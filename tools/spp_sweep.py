#!/usr/bin/env python3
"""Kernel time of config 2 (1200x800, depth 50) at the sample counts a rank of 1, 2, 4 and 8 GPUs renders of the 500-spp
frame, and the straight line through them (time per sample, constant per launch)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import abi, lib  # noqa: E402
from tests.conftest import CAM_C2  # noqa: E402

scene = abi.load_scene(os.path.join(ROOT, "tests", "golden", "scene_c2.rtsc"))
cam = lib.camera(*CAM_C2)
ctx = lib.Context(0)
ctx.upload(scene)
ctx.render(cam, abi.render_params(1200, 800, 250, 50, seed=1))  # warm-up: clocks, code objects, staging buffers
rows = []
for spp in (62, 63, 125, 250, 500):
    best = None
    for rep in range(3):
        _, st = ctx.render(cam, abi.render_params(1200, 800, spp, 50, seed=1))
        best = st["kernel_ms"] if best is None else min(best, st["kernel_ms"])
    rows.append((spp, best))
    print(f"{spp:4d} spp: kernel {best:7.2f} ms  {1200 * 800 * spp / best / 1e3:7.1f} Msamples/s", flush=True)
ctx.close()
x, y = np.array([r[0] for r in rows], float), np.array([r[1] for r in rows], float)
slope, const = np.polyfit(x, y, 1)
print(f"fit: {slope:.4f} ms per sample per pixel + {const:.2f} ms per launch")
for n in (1, 2, 4, 8):
    per_rank = -(-500 // n)
    t = slope * per_rank + const
    print(f"{n} GPU(s): {per_rank} spp per rank -> {t:6.2f} ms of kernel per frame, {500 * 960000 / t / 1e6:6.2f} Gsamples/s before the all-reduce "
          f"({(slope * 500 + const) / t:.2f}x)")
