#!/usr/bin/env python3
"""Where a random world of tests/test_fuzz_worlds.py differs from the oracle: pixels, first-hit objects, materials."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import abi, lib  # noqa: E402
from tests import oracle_binding as ob  # noqa: E402
from tests import test_fuzz_worlds as t  # noqa: E402
seed = int(sys.argv[1])
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 50
scene = t.random_world(1000 + seed)
if len(sys.argv) > 4:  # keep only these top-level objects
    keep = [int(x) for x in sys.argv[4].split(",")]
    r0 = scene.nodes[scene.root]
    kids = scene.children[r0["child0"]: r0["child0"] + r0["child1"]].copy()
    scene.children[r0["child0"]: r0["child0"] + len(keep)] = kids[keep]
    scene.nodes[scene.root]["child1"] = len(keep)
mk = lambda f: f((7, 3.5, 8), (0, 0.7, 0), (0, 1, 0), 32.0, 4.0 / 3.0, 0.05, 10.0, 0.0, 1.0)
cam, ocam = mk(lib.camera), mk(ob.camera)
p = abi.render_params(56, 42, spp, depth, seed=seed)
_, fixed, ost = ob.render(scene, ocam, p, rng=ob.RNG_PHILOX, assoc=ob.ASSOC_FORWARD, has_time=True, threads=16)
want = fixed.astype(np.float64) / 2.0 ** abi.RT_ACCUM_FRAC_BITS
root = scene.nodes[scene.root]
tops = scene.children[root["child0"]: root["child0"] + root["child1"]]
print("top-level kinds:", [int(scene.nodes[c]["kind"]) for c in tops])
for set_min in (2, 1000):
    os.environ["RT_SUBTREE_SET_MIN"] = str(set_min)
    c = lib.Context(0)
    c.upload(scene)
    got, st = c.render(cam, p)
    ids = c.first_hit(cam, 56, 42)
    c.close()
    bad = np.argwhere(np.any(got != want, axis=2))
    print(f"set_min {set_min}: rays {st['rays']} vs {ost['rays']}, differing pixels {len(bad)}")
    for y, x in bad[:12]:
        print("   pixel", (int(y), int(x)), "first-hit top object", int(ids[y, x]), "got", got[y, x], "want", want[y, x])
