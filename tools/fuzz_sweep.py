#!/usr/bin/env python3
"""One-off sweep of the random-world generators of tests/test_fuzz_worlds.py over many seeds (GPU against the oracle,
bit for bit): scene graphs, sphere worlds, importance sampling, large leaf sets and book-1 flat lists.  Usage: tools/fuzz_sweep.py <first seed> <count>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import abi, lib  # noqa: E402
from tests import oracle_binding as ob  # noqa: E402
from tests import test_fuzz_worlds as t  # noqa: E402
first, count = int(sys.argv[1]), int(sys.argv[2])
FRAC = 2.0 ** abi.RT_ACCUM_FRAC_BITS
bad = 0
for seed in range(first, first + count):
    os.environ["RT_CLUSTER_MIN"] = "8" if seed % 2 else "64"  # odd seeds: two-level screens on small sets too
    try:
        t.check_flat_list_case(seed)
    except AssertionError as e:
        bad += 1
        print(f"MISMATCH seed {seed} flat list: {str(e)[:300]}", flush=True)
    for gen, name, flags_list in ((t.random_world, "graph", (0,)), (t.random_sphere_world, "spheres", (0,)),
                                  (t.random_is_world, "importance", (abi.RT_RENDER_IMPORTANCE, 0)),
                                  (t.random_big_set_world, "big sets", (0,))):
        if name == "big sets" and seed % 4 != 0:  # the large worlds cost the oracle more: one seed in four
            continue
        scene = gen(seed)
        sc = (1.0, 1.0, 1.0, 1.0e-4, 1.0e4, 1.0e7)[(seed // 2) % 6]  # half of the worlds at another scale (noise greyed)
        if sc != 1.0:
            scene = t.scale_world(scene, sc)
        t0, t1 = ((0.0, 1.0), (0.25, 0.5), (0.0, 0.0), (-0.5, 1.5))[seed % 4] if name == "spheres" else (0.0, 1.0)
        # now and then a lens that makes the rays overflow -- not for worlds with an image texture: a NaN texture
        # coordinate indexes the image out of bounds in the reference (and in the oracle, which then crashes)
        huge_lens = seed % 11 == 3 and not np.any(scene.textures["kind"] == abi.RT_TEX_IMAGE)
        mk = lambda f: f(((7 + seed % 3) * sc, 3.5 * sc, (8 - seed % 2) * sc), (0, 0.7 * sc, 0), (0, 1, 0), 32.0 + seed % 7, 4.0 / 3.0,
                        1.0e300 if huge_lens else 0.05 * (seed % 2) * sc, 10.0 * sc, t0, t1)
        cam, ocam = mk(lib.camera), mk(ob.camera)
        for flags in flags_list:
            p = abi.render_params(48, 36, 3, 50, seed=seed, flags=flags)
            _, fixed, ost = ob.render(scene, ocam, p, rng=ob.RNG_PHILOX, assoc=ob.ASSOC_FORWARD, has_time=True, threads=16)
            want = fixed.astype(np.float64) / FRAC
            oids = ob.first_hit(scene, ocam, 60, 45)
            for set_min in ((64, 100000) if name == "big sets" else (2, 1000)):
                os.environ["RT_SUBTREE_SET_MIN"] = str(set_min)
                c = lib.Context(0)
                try:
                    c.upload(scene)
                    got, st = c.render(cam, p)
                    ids = c.first_hit(cam, 60, 45)
                finally:
                    c.close()
                if st["rays"] != ost["rays"] or not np.array_equal(got, want) or not np.array_equal(ids, oids):
                    bad += 1
                    print(f"MISMATCH seed {seed} {name} scale {sc:g} flags {flags} set_min {set_min}: rays {st['rays']} vs {ost['rays']}, "
                          f"pixels {int(np.any(got != want, axis=2).sum())} (largest difference {np.abs(got - want).max() * FRAC:.0f} units of 2^-36), "
                          f"ids {int((ids != oids).sum())}", flush=True)
    if (seed - first) % 20 == 19:
        print(f"... {seed - first + 1} seeds done, {bad} mismatches", flush=True)
print(f"done: {count} seeds, {bad} mismatches")
