#!/usr/bin/env python3
"""Kernel time of the config-2 frame against samples per pixel and depth: the intercept of the line is the
time the launch spends outside the steady state (start-up and the tail of long paths)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import abi, lib  # noqa: E402

scene = abi.load_scene(os.path.join(ROOT, "tests", "golden", "scene_c2.rtsc"))
cam = lib.camera((12, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.1, 10.0)
W, H = 1200, 800
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev)
ctx = lib.Context(0, s.cuda_stream)
ctx.upload(scene)
buf = torch.zeros((H, W, 3), dtype=torch.int64, device=dev)
spps = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,2,4,8,16,32,63,125,250".split(","))]
depths = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "50,8".split(","))]
for depth in depths:
    rows = []
    for spp in spps:
        best = 1e9
        for rep in range(3):
            ctx.render_device(cam, abi.render_params(W, H, spp, depth, seed=1 + rep), buf.data_ptr())
            ctx.synchronize()
            st = ctx.take_stats()
            best = min(best, st["kernel_ms"] / st["kernel_launches"])
        rows.append((spp, best))
        print(f"depth {depth:3d} spp {spp:4d}  kernel {best:9.3f} ms  {best / spp:8.4f} ms/spp", flush=True)
    x = np.array([r[0] for r in rows[-4:]], float)
    y = np.array([r[1] for r in rows[-4:]], float)
    a, b = np.polyfit(x, y, 1)
    print(f"depth {depth}: fit over the last four points: {a:.4f} ms/spp + {b:.3f} ms", flush=True)
ctx.close()
