#!/usr/bin/env python3
"""Where a scaled random world (tests/test_fuzz_worlds.py::test_random_world_at_other_scales) differs from the oracle.
Usage: tools/scale_probe.py <seed> <scale>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import abi, lib  # noqa: E402
from tests import oracle_binding as ob  # noqa: E402
from tests import test_fuzz_worlds as t  # noqa: E402
seed, scale = int(sys.argv[1]), float(sys.argv[2])
scene = t.scale_world(t.random_world(1000 + seed), scale)
mk = lambda f: f((7 * scale, 3.5 * scale, 8 * scale), (0, 0.7 * scale, 0), (0, 1, 0), 32.0, 4.0 / 3.0, 0.05 * scale, 10.0 * scale, 0.0, 1.0)
cam, ocam = mk(lib.camera), mk(ob.camera)
p = abi.render_params(48, 36, 3, 50, seed=seed)
_, fixed, ost = ob.render(scene, ocam, p, rng=ob.RNG_PHILOX, assoc=ob.ASSOC_FORWARD, has_time=True, threads=16)
want = fixed.astype(np.float64) / 2.0 ** abi.RT_ACCUM_FRAC_BITS
root = scene.nodes[scene.root]
tops = scene.children[root["child0"]: root["child0"] + root["child1"]]
print("top-level kinds:", [int(scene.nodes[c]["kind"]) for c in tops])
print("texture kinds:", [int(x) for x in scene.textures["kind"]], "material kinds:", [int(x) for x in scene.materials["kind"]])
for set_min, exact in ((2, False), (1000, False), (2, True)):
    os.environ["RT_SUBTREE_SET_MIN"] = str(set_min)
    os.environ["RT_CLUSTER_MIN"] = "4"
    c = lib.Context(0)
    c.upload(scene)
    c.set_exact_only(exact)
    got, st = c.render(cam, p)
    ids = c.first_hit(cam, 48, 36)
    c.close()
    bad = np.argwhere(np.any(got != want, axis=2))
    print(f"set_min {set_min} exact_only {exact}: rays {st['rays']} vs {ost['rays']}, differing pixels {len(bad)}, "
          f"largest difference {np.abs(got - want).max() * 2.0 ** 36:.0f} units")
    for y, x in bad[:6]:
        print("   pixel", (int(y), int(x)), "first-hit node", int(ids[y, x]), "kind", int(scene.nodes[ids[y, x]]["kind"]) if ids[y, x] >= 0 else None,
              "got", got[y, x], "want", want[y, x])
