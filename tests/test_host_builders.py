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


def build_app(tmp_path, compiler):
    exe = str(tmp_path / f"rt_one_weekend_{os.path.basename(compiler)}")
    subprocess.run([compiler, "-std=c++14", "-O2", "-ffp-contract=off", f"-I{ROOT}/include", "-o", exe,
                    os.path.join(APPS, "rt_one_weekend.cpp"), f"-L{ROOT}/raytracing_amd", "-lraytracing_amd",
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
