#!/usr/bin/env python3
"""Kernel time of the Next-Week final scene with single top-level objects left out: what each part of the world costs."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import abi, lib  # noqa: E402
G = os.path.join(ROOT, "tests", "golden")
k = json.load(open(os.path.join(G, "kat_b2.json")))["final"]
sc = abi.load_scene(os.path.join(G, "scene_b2_final.rtsc"))
cam = lib.camera(k["lookfrom"], k["lookat"], (0, 1, 0), k["vfov"], k["aspect"], k["aperture"], 10.0, 0.0, 1.0)
root = sc.nodes[sc.root]
kids = sc.children[root["child0"]: root["child0"] + root["child1"]].copy()
names = {v: n for n, v in vars(abi).items() if n.startswith("RT_NODE_")}
ctx = lib.Context(0)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
def run(label, keep):
    import copy
    s = copy.deepcopy(sc)
    s.children[root["child0"]: root["child0"] + len(keep)] = keep
    s.nodes[s.root]["child1"] = len(keep)
    ctx.upload(s)
    best = None
    for _ in range(2):
        _, st = ctx.render(cam, abi.render_params(800, 800, spp, 50, seed=1))
        best = st if best is None or st["kernel_ms"] < best["kernel_ms"] else best
    print(f"{label:40s} {best['kernel_ms']:7.1f} ms  rays/sample {best['rays'] / best['samples']:.2f}", flush=True)
run("whole world", kids)
for i, c in enumerate(kids):
    n = sc.nodes[c]
    run(f"without child {i} ({names.get(int(n['kind']), n['kind'])})", np.delete(kids, i))
ctx.close()
