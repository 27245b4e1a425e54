#!/usr/bin/env python3
"""Host-side cost of rt_render around its kernel: wall time of the call against the event-timed kernel, first call
(allocations, code-object load) and later calls, at the headline frame size.  Usage: tools/call_overhead.py [spp]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import abi, lib  # noqa: E402
from tests.conftest import CAM_C2, gold  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 500
scene = abi.load_scene(gold("scene_c2.rtsc"))
cam = lib.camera(*CAM_C2)
with lib.Context(0) as ctx:
    ctx.upload(scene)
    for k in range(4):
        t0 = time.perf_counter()
        _, st = ctx.render(cam, abi.render_params(1200, 800, spp, 50, seed=1 + k))
        dt = (time.perf_counter() - t0) * 1e3
        print(f"call {k}: {dt:.1f} ms, kernel {st['kernel_ms']:.1f} ms, host side {dt - st['kernel_ms']:.1f} ms", flush=True)
