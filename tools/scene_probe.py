#!/usr/bin/env python3
"""Traversal counters of the scene-graph kernel (library built with -DRT_SCENE_PROBE): stack entries by node kind,
loop trips per wave-level trace, lane fill.  Small frames: the counters are device-wide atomics."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RT_LIB_PATH", os.path.join(ROOT, "tools", "libscprobe.so"))
from raytracing_amd import abi, lib  # noqa: E402
G = os.path.join(ROOT, "tests", "golden")
kat = json.load(open(os.path.join(G, "kat_b2.json")))
ctx = lib.Context(0)
for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["bouncing", "final", "cornell_box"]):
    flags = 0
    if name in ("c4", "c5"):  # a BASELINE configuration of bench.py (c5: light sampled by importance)
        from bench import CONFIGS
        cfg = CONFIGS[name]
        sc = abi.load_scene(os.path.join(G, cfg["scene"]))
        cam = cfg["camera"](lib)
        flags = abi.RT_RENDER_IMPORTANCE if cfg["flags"] == "importance" else 0
    else:
        k = kat[name]
        sc = abi.load_scene(os.path.join(G, f"scene_b2_{name}.rtsc"))
        cam = lib.camera(k["lookfrom"], k["lookat"], (0, 1, 0), k["vfov"], k["aspect"], k["aperture"], 10.0, 0.0, 1.0)
    ctx.upload(sc)
    ctx.take_stats()
    _, st = ctx.render(cam, abi.render_params(400, 400, 8, 50, seed=1, flags=flags))
    print(f"== {name}: rays {st['rays']} samples {st['samples']}", flush=True)
    sys.stderr.flush()
ctx.close()
