#!/usr/bin/env python3
"""Kernel time of the book-2 worlds (bouncing spheres, final scene, Cornell boxes, Perlin spheres) at full frame size and a
few samples per pixel.  RT_LIB_PATH selects the library under test."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import abi, lib  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
kat = json.load(open(os.path.join(G, "kat_b2.json")))
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["bouncing", "final", "cornell_box", "cornell_smoke", "two_perlin"]
SIZE = {"bouncing": (1200, 675), "final": (800, 800)}
ctx = lib.Context(0)
for name in names:
    k = kat[name]
    path = os.path.join(G, f"scene_b2_{name}.rtsc")
    sc = abi.load_scene(path if os.path.exists(path) else path + ".gz")
    cam = lib.camera(k["lookfrom"], k["lookat"], (0, 1, 0), k["vfov"], k["aspect"], k["aperture"], 10.0, 0.0, 1.0)
    ctx.upload(sc)
    W, H = SIZE.get(name, (600, 600))
    best = None
    for rep in range(2):
        _, st = ctx.render(cam, abi.render_params(W, H, spp, 50, seed=1))
        if best is None or st["kernel_ms"] < best["kernel_ms"]:
            best = st
    ms = best["kernel_ms"]
    print(f"{name:14s} {W}x{H}x{spp}: kernel {ms:8.1f} ms  {W * H * spp / ms / 1e3:8.1f} Msamples/s  "
          f"{best['rays'] / ms / 1e3:8.1f} Mrays/s  rays/sample {best['rays'] / best['samples']:.2f}", flush=True)
ctx.close()
