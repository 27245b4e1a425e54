import json
import math
import os
import sys

import numpy as np
import pytest
try:  # first, before anything loads libraytracing_amd.so: PyTorch ships its own HIP runtime (same soname as the
    # system's), and a process must settle on one copy before either library touches the GPU; some tests hand torch
    # tensors to the library.  Without PyTorch the CPU tests still run (raytracing_amd.lib guards the import likewise).
    import torch  # noqa: F401
except ImportError:
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from raytracing_amd import abi  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def gold(name):
    return os.path.join(GOLD, name)


@pytest.fixture(scope="session")
def kat_b1():
    with open(gold("kat_b1.json")) as fp:
        return json.load(fp)


@pytest.fixture(scope="session")
def scene_c1():
    return abi.load_scene(gold("scene_c1.rtsc"))


@pytest.fixture(scope="session")
def scene_c2():
    return abi.load_scene(gold("scene_c2.rtsc"))


# camera arguments of the two book-1 configurations: (lookfrom, lookat, vup, vfov, aspect, aperture, focus)
C1_LOOKFROM, C1_LOOKAT = (3, 3, 2), (0, 0, -1)
CAM_C1 = (C1_LOOKFROM, C1_LOOKAT, (0, 1, 0), 20.0, 16.0 / 9.0, 2.0,
          math.sqrt(sum((a - b) ** 2 for a, b in zip(C1_LOOKFROM, C1_LOOKAT))))
CAM_C2 = ((12, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 3.0 / 2.0, 0.1, 10.0)


def make_sphere_scene(spheres, materials, sky_mode=abi.RT_SKY_GRADIENT, background=(0, 0, 0)):
    """Build a flat-list scene.  spheres: [(cx, cy, cz, r, material_index)];
    materials: [("lambertian", (r,g,b)) | ("metal", (r,g,b), fuzz) | ("dielectric", ir)]"""
    n = len(spheres)
    nodes = np.zeros(n + 1, dtype=abi.NODE_DTYPE)
    nodes[0]["kind"], nodes[0]["material"], nodes[0]["child0"], nodes[0]["child1"] = abi.RT_NODE_LIST, -1, 0, n
    for i, (cx, cy, cz, r, m) in enumerate(spheres):
        nd = nodes[i + 1]
        nd["kind"], nd["material"], nd["child0"], nd["child1"] = abi.RT_NODE_SPHERE, m, -1, -1
        nd["p"][:4] = (cx, cy, cz, r)
    children = np.arange(1, n + 1, dtype=np.int32)
    mats = np.zeros(len(materials), dtype=abi.MATERIAL_DTYPE)
    texs = []
    for i, m in enumerate(materials):
        if m[0] == "lambertian":
            t = np.zeros((), dtype=abi.TEXTURE_DTYPE)
            t["kind"], t["a"], t["b"] = abi.RT_TEX_SOLID, -1, -1
            t["p"][:3] = m[1]
            texs.append(t)
            mats[i]["kind"], mats[i]["texture"] = abi.RT_MAT_LAMBERTIAN, len(texs) - 1
            mats[i]["albedo"] = m[1]
        elif m[0] == "metal":
            mats[i]["kind"], mats[i]["texture"] = abi.RT_MAT_METAL, -1
            mats[i]["albedo"] = m[1]
            mats[i]["param"] = min(m[2], 1.0)
        elif m[0] == "dielectric":
            mats[i]["kind"], mats[i]["texture"], mats[i]["param"] = abi.RT_MAT_DIELECTRIC, -1, m[1]
        else:
            raise ValueError(m)
    textures = np.array(texs, dtype=abi.TEXTURE_DTYPE) if texs else np.zeros(0, dtype=abi.TEXTURE_DTYPE)
    return abi.Scene(0, nodes, children, mats, textures, np.zeros(0, dtype=abi.PERLIN_DTYPE), [], sky_mode,
                     tuple(background))


# Worlds whose arithmetic calls log / sin / cos / acos / atan2 (media, noise and image textures, sphere uv, book 3's
# light sampling): the device's libm is within one ulp of glibc's (two for atan2) but not identical, and a last-bit
# difference in a hit distance or a texture coordinate moves a sample by ~1e-16 of its value.  Samples are rounded to
# 2^-36 before they are summed, so kernel and oracle still agree unless a sample sits within that distance of a rounding
# boundary: measured about 40 one-unit differences in a thousand million samples.  The contract for such worlds is:
# at most LIBM_MAX_VALUES values per render off by at most LIBM_MAX_UNITS units of 2^-36 (ray counts are asserted
# exactly where the test has them).  Every other world -- configurations 1 to 3 among them -- is held to exact equality.
LIBM_MAX_UNITS, LIBM_MAX_VALUES = 2, 4


def scene_calls_libm(scene, flags=0):
    """Whether rendering this world evaluates log / sin / cos / acos / atan2 on the device: a constant medium draws its
    free-flight distance with log (B2/constant_medium.h:50), an isotropic material comes with one, a noise texture
    takes a sin (B2/texture.h:56), an image texture reads the sphere's uv, which are an acos and an atan2
    (B2/sphere.h:25-38), and book 3's light and cosine sampling (RT_RENDER_IMPORTANCE) calls sin and cos.  Nothing else
    does: a checker texture needs only the sign of its sines, rotate_y's sine and cosine are worked out on the host."""
    if flags & abi.RT_RENDER_IMPORTANCE:
        return True
    if np.any(scene.nodes["kind"] == abi.RT_NODE_MEDIUM) or np.any(scene.materials["kind"] == abi.RT_MAT_ISOTROPIC):
        return True
    kinds = scene.textures["kind"] if len(scene.textures) else np.zeros(0, dtype=np.int32)
    return bool(np.any((kinds == abi.RT_TEX_NOISE) | (kinds == abi.RT_TEX_IMAGE)))


def assert_sums_match(got, want, libm_world, what=None):
    """got, want: double colour sums [H, W, 3] of the kernel and of the oracle."""
    if not libm_world:
        assert np.array_equal(got, want), (what, np.abs(got - want).max())
        return
    frac = 2.0 ** abi.RT_ACCUM_FRAC_BITS
    units = np.abs(np.rint(got * frac) - np.rint(want * frac))
    assert units.max() <= LIBM_MAX_UNITS and np.count_nonzero(units) <= LIBM_MAX_VALUES, \
        (what, units.max(), np.count_nonzero(units))
