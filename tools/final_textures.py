#!/usr/bin/env python3
"""Kernel time of the Next-Week final scene with its Perlin / image textures replaced by solid colours: what texture
evaluation (executed by the whole wave when one lane needs it) costs."""
import copy, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import abi, lib  # noqa: E402
G = os.path.join(ROOT, "tests", "golden")
k = json.load(open(os.path.join(G, "kat_b2.json")))["final"]
sc = abi.load_scene(os.path.join(G, "scene_b2_final.rtsc"))
cam = lib.camera(k["lookfrom"], k["lookat"], (0, 1, 0), k["vfov"], k["aspect"], k["aperture"], 10.0, 0.0, 1.0)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ctx = lib.Context(0)
def run(label, kinds):
    s = copy.deepcopy(sc)
    for t in s.textures:
        if int(t["kind"]) in kinds:
            t["kind"] = abi.RT_TEX_SOLID
            t["p"][:3] = 0.5
    ctx.upload(s)
    best = None
    for _ in range(3):
        _, st = ctx.render(cam, abi.render_params(800, 800, spp, 50, seed=1))
        best = st if best is None or st["kernel_ms"] < best["kernel_ms"] else best
    print(f"{label:36s} {best['kernel_ms']:7.2f} ms  rays/sample {best['rays'] / best['samples']:.2f}", flush=True)
run("as is", ())
run("Perlin -> solid", (abi.RT_TEX_NOISE,))
run("image -> solid", (abi.RT_TEX_IMAGE,))
run("both -> solid", (abi.RT_TEX_NOISE, abi.RT_TEX_IMAGE))
ctx.close()
