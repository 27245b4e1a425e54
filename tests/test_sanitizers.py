"""The CPU side under AddressSanitizer + UndefinedBehaviorSanitizer (host code only, no GPU): the oracle rendering every
kind of world in both random-number modes, and the C++ host layer (recording classes, scene builders, flattener)
rolling the reference's scenes.  Any sanitizer report aborts the program (-fno-sanitize-recover), which fails the test;
the results must also be the unsanitized ones."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from raytracing_amd import abi
from tests import oracle_binding as ob
from tests.conftest import ROOT, gold

SCENES = ["scene_c1.rtsc", "scene_c2.rtsc", "scene_b2_bouncing.rtsc", "scene_b2_two_perlin.rtsc", "scene_b2_cornell_smoke.rtsc",
          "scene_b2_final.rtsc", "scene_b2_simple_light.rtsc", "scene_b3_cornell.rtsc"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def _make(directory):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, directory), "sanitize"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.fail(f"make -C {directory} sanitize failed:\n{r.stdout}\n{r.stderr}")


def _fnv(h, data: bytes):
    for b in data:
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def test_oracle_under_asan_and_ubsan():
    _make("oracle")
    exe = os.path.join(ROOT, "oracle", "_san", "sanitize_check")
    paths = [gold(s) for s in SCENES]
    r = subprocess.run([exe] + paths, capture_output=True, text=True, env=ENV)
    assert r.returncode == 0, r.stdout + r.stderr
    got = dict(line.split() for line in r.stdout.strip().splitlines())
    # the same renders through the unsanitized library, hashed the same way
    W, H, SPP = 24, 16, 3
    for path in paths[:3] + paths[-1:]:
        scene = abi.load_scene(path)
        big = any(k in path for k in ("cornell", "final", "b3"))
        cam = ob.camera((278, 278, -800) if big else (13, 2.5, 4), (278, 278, 0) if big else (0, 0.5, 0), (0, 1, 0),
                        40.0 if big else 25.0, W / H, 0.05, 10.0, 0.0, 1.0)
        p = abi.render_params(W, H, SPP, 50, seed=7, sample_begin=2)
        h = 0xCBF29CE484222325
        for rng in (ob.RNG_RAND, ob.RNG_PHILOX):
            ob.lib().orc_srand(1)
            _, fixed, st = ob.render(scene, cam, p, rng=rng, assoc=ob.ASSOC_REFERENCE, has_time=True, threads=2)
            h = _fnv(h, fixed.tobytes())
            h = _fnv(h, np.array([st["rays"], st["nan_samples"]], dtype=np.uint64).tobytes())
        if scene.lights_plus1:
            pi = abi.render_params(W, H, SPP, 50, seed=7, sample_begin=2, flags=abi.RT_RENDER_IMPORTANCE)
            _, fixed, _ = ob.render(scene, cam, pi, rng=ob.RNG_PHILOX, assoc=ob.ASSOC_REFERENCE, has_time=True, threads=2)
            h = _fnv(h, fixed.tobytes())
        h = _fnv(h, ob.first_hit(scene, cam, W, H).astype(np.int32).tobytes())
        assert got[path] == f"{h:016x}", path


def test_host_layer_under_asan_and_ubsan(tmp_path):
    _make("apps")
    san = os.path.join(ROOT, "apps", "_san")
    cases = [("rt_one_weekend", ["--scene", "final"], "scene_c2.rtsc"), ("rt_one_weekend", ["--scene", "three"], "scene_c1.rtsc")]
    cases += [("rt_next_week", ["--scene", s], f"scene_b2_{s}.rtsc")
              for s in ("bouncing", "two_perlin", "simple_light", "cornell_box", "cornell_smoke", "final")]
    cases += [("rt_rest_of_life", [], "scene_b3_cornell.rtsc")]
    for app, args, golden in cases:
        out = tmp_path / golden
        r = subprocess.run([os.path.join(san, app)] + args + ["--no-render", "--dump-scene", str(out)], capture_output=True,
                           text=True, env=ENV)
        assert r.returncode == 0, (app, args, r.stderr[-2000:])
        with open(gold(golden), "rb") as fp:
            assert out.read_bytes() == fp.read(), (app, args)
    # bad options end in a usage error, not in a sanitizer report
    r = subprocess.run([os.path.join(san, "rt_one_weekend"), "--scene", "nonsense", "--no-render"], capture_output=True, env=ENV)
    assert r.returncode == 2
