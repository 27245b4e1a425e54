#!/usr/bin/env python3
"""Kernel time of config-2 launches enqueued back to back on one stream (no host wait between them) against launches
the host waits for one by one: what part of a short launch's constant is the GPU idling between launches."""
import os
import sys

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
spps = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "63,500".split(","))]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for spp in spps:
    for mode in ("waited", "back to back"):
        for rep in range(2):
            for k in range(n):
                ctx.render_device(cam, abi.render_params(W, H, spp, 50, seed=1 + k), buf.data_ptr())
                if mode == "waited":
                    ctx.synchronize()
            ctx.synchronize()
            st = ctx.take_stats()
        print(f"spp {spp:4d} {mode:13s}: {st['kernel_ms'] / st['kernel_launches']:8.3f} ms per launch ({st['kernel_launches']} launches)", flush=True)
ctx.close()
