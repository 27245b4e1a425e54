#!/usr/bin/env python3
"""One BASELINE configuration at its own frame size on one GPU (the command tools/profile_config.sh runs under
rocprofv3): `tools/bench_config.py <c1|c2|c3|c4|c5> [spp] [launches]`.  Prints one JSON line: frame, event-timed
kernel time per launch, samples and rays PER LAUNCH -- what bench.py's `configs` block scales the committed counters by."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CONFIGS, GOLD  # noqa: E402
from raytracing_amd import abi, lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
cfg = CONFIGS[name]
spp = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "-" else cfg["spp"]
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 2
W, H = cfg["width"], cfg["height"]
flags = abi.RT_RENDER_IMPORTANCE if cfg["flags"] == "importance" else 0
with lib.Context(0) as ctx:
    ctx.upload(abi.load_scene(os.path.join(GOLD, cfg["scene"])))
    cam = cfg["camera"](lib)
    best = None
    for rep in range(launches):
        _, st = ctx.render(cam, abi.render_params(W, H, spp, 50, seed=1, flags=flags))
        if best is None or st["kernel_ms"] < best["kernel_ms"]:
            best = st
print(json.dumps({"config_name": name, "workload": cfg["workload"], "width": W, "height": H, "spp": spp, "kernel": cfg["kernel"],
                  "kernel_ms": best["kernel_ms"], "samples": best["samples"], "rays": best["rays"],
                  "msamples_per_s": best["samples"] / best["kernel_ms"] / 1e3, "mrays_per_s": best["rays"] / best["kernel_ms"] / 1e3}))
