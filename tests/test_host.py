"""CPU-side checks of the product: the C-ABI library loads and exports what the header
declares, its host helpers restate the reference exactly, scene files round-trip, and
without a GPU it fails loudly (no CPU fallback)."""
import ctypes as C
import hashlib
import os
import re

import numpy as np
import pytest

from raytracing_amd import abi, lib
from tests import oracle_binding as ob
from tests.conftest import CAM_C1, CAM_C2, ROOT, gold


def test_library_exports_every_declared_symbol():
    with open(os.path.join(ROOT, "include", "raytracing_amd.h")) as fp:
        header = fp.read()
    body = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", body))
    assert declared, "no prototypes found in the header"
    L = lib.load()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} is declared in raytracing_amd.h but not exported"
    assert declared == set(lib.EXPORTED_SYMBOLS)


def test_struct_layouts_match_header_sizes():
    assert C.sizeof(abi.rt_node) == 112 and C.sizeof(abi.rt_material) == 40
    assert C.sizeof(abi.rt_texture) == 48 and C.sizeof(abi.rt_perlin) == 9216
    assert C.sizeof(abi.rt_camera_desc) == 8 * 24 and C.sizeof(abi.rt_render_params) == 32


@pytest.mark.parametrize("args", [CAM_C1, CAM_C2, ((478, 278, -600), (278, 278, 0), (0, 1, 0), 40.0, 1.0, 0.0, 10.0)])
def test_camera_init_matches_oracle_bitwise(args):
    a = lib.camera(*args, 0.0, 1.0)
    b = ob.camera(*args, 0.0, 1.0)
    assert bytes(a) == bytes(b)


def test_write_color_and_ppm_reproduce_reference_output(tmp_path, kat_b1):
    """Fed the reference's colour sums, the library writes the reference's P3 file byte for byte."""
    ref = np.load(gold("c1_400x225x100_ref.npz"))
    # radiance is float32(sum/100): recover the 8-bit image through the library from exact sums instead
    cam = ob.camera(*CAM_C1)
    ob.lib().orc_srand(1)
    acc, _, _ = ob.render(abi.load_scene(gold("scene_c1.rtsc")), cam, abi.render_params(400, 225, 100, 50),
                          rng=ob.RNG_RAND, assoc=ob.ASSOC_REFERENCE, want_fixed=False)
    rgb = lib.write_color(acc, 100)
    assert np.array_equal(rgb, ref["rgb"])
    out = tmp_path / "c1.ppm"
    lib.write_ppm(str(out), rgb)
    assert hashlib.sha256(out.read_bytes()).hexdigest() == kat_b1["c1_400x225x100"]["ppm_sha256"]


def test_write_color_edge_values():
    a = np.array([[[0.0, 1.0, 0.998], [4.0, 1e-9, np.nan], [-1.0, 0.25, 0.9999]]])
    got = lib.write_color(a, 1)
    assert got.tolist() == [[[0, 255, 255], [255, 0, 0], [0, 128, 255]]]


def test_scene_file_round_trip(tmp_path, scene_c2):
    p = tmp_path / "x.rtsc"
    abi.save_scene(p, scene_c2)
    with open(gold("scene_c2.rtsc"), "rb") as fp:
        assert p.read_bytes() == fp.read()
    s = abi.load_scene(p)
    assert np.array_equal(s.nodes, scene_c2.nodes) and np.array_equal(s.materials, scene_c2.materials)


def test_no_gpu_means_loud_failure():
    L = lib.load()
    if L.rt_get_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(lib.RtError) as e:
        lib.Context(0)
    assert e.value.code == abi.RT_ERR_NO_DEVICE


def test_error_strings():
    L = lib.load()
    assert L.rt_error_string(0) == b"ok"
    for code in range(1, 8):
        assert L.rt_error_string(code) not in (b"ok", b"unknown error code")
    assert L.rt_error_string(99) == b"unknown error code"


def test_checker_sign_rule_matches_libm():
    """The kernels take the sign of sin(10 x) from k = rint(x / pi), r = fma(-k, pi_hi, x) when |r| > 1e-8 and |x| < 1e8
    (rt_render_common.h: sin_sign) instead of evaluating sin().  The rule restated with exact rational arithmetic for
    the fused multiply-add, against the sign of libm's sin: random arguments, and arguments a few ulps to a few
    1e-8 away from multiples of pi, where the rule is closest to its limit."""
    import math
    from fractions import Fraction
    PI_HI = 3.141592653589793116
    INV_PI = 0.318309886183790671537767526745

    def rule(x):
        k = float(np.rint(x * INV_PI))
        r = float(Fraction(x) - Fraction(k) * Fraction(PI_HI))  # one rounding, like fma
        if abs(x) < 1e8 and abs(r) > 1e-8:
            return 1 if ((int(k) & 1) != 0) == (r < 0) else -1
        return None  # the kernels call sin() here

    rng = np.random.default_rng(5)
    xs = list(rng.uniform(-2.0e4, 2.0e4, 4000)) + list(rng.uniform(-9.9e7, 9.9e7, 2000))
    for k in list(range(-40, 41)) + [1000, 31831, 3183098, 31830988]:
        base = k * math.pi
        for d in (0.0, 1.1e-8, -1.1e-8, 3e-8, -3e-8, 1e-6, -1e-6, 1.5, -1.5, 1.5707963, -1.5707963):
            xs.append(base + d)
    decided = 0
    for x in xs:
        got = rule(float(x))
        if got is None:
            continue
        decided += 1
        s = math.sin(float(x))
        assert s != 0.0 and got == (1 if s > 0 else -1), (x, got, s)
    assert decided > 0.95 * len(xs)
