"""The C++ host layer (raytracing_amd/host/*.hpp + apps/): reference-shaped scene builders must
roll the reference's scenes byte for byte under g++ AND under clang (the reference itself does
not: its draws sit in argument lists), and the app must drive the GPU through the C ABI."""
import hashlib
import os
import shutil
import subprocess

import numpy as np
import pytest

from raytracing_amd import abi, lib
from tests.conftest import CAM_C1, ROOT, gold

APPS = os.path.join(ROOT, "apps")


def build_app(tmp_path, compiler, app="rt_one_weekend"):
    exe = str(tmp_path / f"{app}_{os.path.basename(compiler)}")
    subprocess.run([compiler, "-std=c++14", "-O2", "-ffp-contract=off", f"-I{ROOT}/include", "-o", exe,
                    os.path.join(APPS, app + ".cpp"), f"-L{ROOT}/raytracing_amd", "-lraytracing_amd",
                    f"-Wl,-rpath,{ROOT}/raytracing_amd"], check=True, capture_output=True)
    return exe


@pytest.mark.parametrize("compiler", ["g++", "/opt/rocm/lib/llvm/bin/clang++"])
def test_scene_builders_roll_the_reference_scenes(tmp_path, compiler):
    if shutil.which(compiler) is None:
        pytest.skip(f"{compiler} not installed")
    exe = build_app(tmp_path, compiler)
    for scene, golden in (("final", "scene_c2.rtsc"), ("three", "scene_c1.rtsc")):
        out = tmp_path / f"{scene}.rtsc"
        subprocess.run([exe, "--scene", scene, "--no-render", "--dump-scene", str(out)], check=True)
        with open(gold(golden), "rb") as fp:
            assert out.read_bytes() == fp.read(), f"{scene} scene differs from the reference dump"


@pytest.mark.parametrize("compiler", ["g++", "/opt/rocm/lib/llvm/bin/clang++"])
def test_book2_scene_builders_roll_the_reference_scenes(tmp_path, compiler):
    """All eight scenes of B2/RayTracing.cpp:279-351 -- including the BVH the reference's constructor builds
    (random split axes, std::sort), the Perlin tables and rotate_y's boxes -- byte for byte."""
    if shutil.which(compiler) is None:
        pytest.skip(f"{compiler} not installed")
    exe = build_app(tmp_path, compiler, "rt_next_week")
    for scene in ("bouncing", "two_spheres", "two_perlin", "earth", "simple_light", "cornell_box", "cornell_smoke", "final"):
        out = tmp_path / f"{scene}.rtsc"
        subprocess.run([exe, "--scene", scene, "--no-render", "--dump-scene", str(out)], check=True, capture_output=True)
        with open(gold(f"scene_b2_{scene}.rtsc"), "rb") as fp:
            assert out.read_bytes() == fp.read(), f"{scene} differs from the reference dump"


@pytest.mark.gpu
def test_next_week_app_renders_through_the_c_abi(tmp_path):
    import json
    exe = build_app(tmp_path, "g++", "rt_next_week")
    r = subprocess.run([exe, "--scene", "cornell_box", "--width", "120", "--spp", "7", "--seed", "3"], check=True,
                       capture_output=True)
    with open(gold("kat_b2.json")) as fp:
        k = json.load(fp)["cornell_box"]
    with lib.Context(0) as ctx:
        ctx.upload(abi.load_scene(gold("scene_b2_cornell_box.rtsc")))
        cam = lib.camera(k["lookfrom"], k["lookat"], (0, 1, 0), k["vfov"], k["aspect"], k["aperture"], 10.0, 0.0, 1.0)
        sums, _ = ctx.render(cam, abi.render_params(120, 120, 7, 50, seed=3))
    want = tmp_path / "want.ppm"
    lib.write_ppm(str(want), lib.write_color(sums, 7))
    assert r.stdout == want.read_bytes()


@pytest.mark.gpu
def test_app_renders_through_the_c_abi(tmp_path):
    """apps/rt_one_weekend (C++ host -> C ABI -> GPU -> P3) equals the same render driven from Python."""
    exe = build_app(tmp_path, "g++")
    r = subprocess.run([exe, "--scene", "three", "--width", "160", "--spp", "9", "--seed", "5"], check=True,
                       capture_output=True)
    with lib.Context(0) as ctx:
        ctx.upload(abi.load_scene(gold("scene_c1.rtsc")))
        sums, _ = ctx.render(lib.camera(*CAM_C1), abi.render_params(160, 90, 9, 50, seed=5))
    rgb = lib.write_color(sums, 9)
    want = tmp_path / "want.ppm"
    lib.write_ppm(str(want), rgb)
    assert hashlib.sha256(r.stdout).hexdigest() == hashlib.sha256(want.read_bytes()).hexdigest()
    assert r.stdout.startswith(b"P3\n160 90\n255\n")
