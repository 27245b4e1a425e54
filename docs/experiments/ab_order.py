#!/usr/bin/env python3
"""Same-process comparison of the work queue in frame order and in tile order (dearest tiles first): two contexts on one
stream, launches alternating between them, event-timed kernel time (the ordered launches include the tile pass)."""
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
ctxs = {}
for setting in ("0", "1"):
    os.environ["RT_PIXEL_ORDER"] = setting
    ctxs[setting] = lib.Context(0, s.cuda_stream)
    ctxs[setting].upload(scene)
buf = torch.zeros((H, W, 3), dtype=torch.int64, device=dev)
spps = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "63,500".split(","))]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for spp in spps:
    t = {"0": [], "1": []}
    for rep in range(reps + 1):
        for setting, ctx in ctxs.items():
            ctx.render_device(cam, abi.render_params(W, H, spp, 50, seed=1 + rep), buf.data_ptr())
            ctx.synchronize()
            st = ctx.take_stats()
            if rep > 0:
                t[setting].append(st["kernel_ms"])
    a, b = np.array(t["0"]), np.array(t["1"])
    print(f"spp {spp:4d}: frame order {a.mean():8.3f} +- {a.std():.3f} ms   tile order {b.mean():8.3f} +- {b.std():.3f} ms   "
          f"difference {b.mean() - a.mean():+.3f} ms ({100 * (b.mean() / a.mean() - 1):+.2f} %)", flush=True)
