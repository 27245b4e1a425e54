"""ctypes binding of oracle/liboracle.so -- the CPU parity checker (test infrastructure).

Only tests, the smoke check and the cpu_baseline leg of bench.py import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from raytracing_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

RNG_RAND, RNG_PHILOX = 0, 1
ASSOC_REFERENCE, ASSOC_FORWARD = 0, 1

_lib = None


def build() -> None:
    subprocess.run(["make", "-C", ORACLE_DIR, "liboracle.so"], check=True, capture_output=True)


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(ORACLE_DIR, "rt_oracle.c")
        if not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_render.restype = C.c_int
        L.orc_render.argtypes = [C.POINTER(abi.rt_scene_desc), C.POINTER(abi.rt_camera_desc),
                                 C.POINTER(abi.rt_render_params), C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_render_rows.restype = C.c_int
        L.orc_render_rows.argtypes = [C.POINTER(abi.rt_scene_desc), C.POINTER(abi.rt_camera_desc),
                                      C.POINTER(abi.rt_render_params), C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_first_hit.restype = C.c_int
        L.orc_first_hit.argtypes = [C.POINTER(abi.rt_scene_desc), C.POINTER(abi.rt_camera_desc), C.c_int, C.c_int,
                                    C.c_void_p]
        L.orc_camera_init.restype = None
        L.orc_camera_init.argtypes = [C.POINTER(abi.rt_camera_desc), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double)] + [C.c_double] * 6
        L.orc_philox4x32_10.restype = None
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_write_color.restype = None
        L.orc_write_color.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_srand.argtypes = [C.c_uint]
        L.orc_rand.restype = C.c_int
        _lib = L
    return _lib


def camera(lookfrom, lookat, vup, vfov, aspect, aperture, focus, time0=0.0, time1=0.0) -> abi.rt_camera_desc:
    cam = abi.rt_camera_desc()
    v = lambda a: (C.c_double * 3)(*[float(x) for x in a])
    lib().orc_camera_init(C.byref(cam), v(lookfrom), v(lookat), v(vup), vfov, aspect, aperture, focus, time0, time1)
    return cam


def render(scene: abi.Scene, cam, params, rng=RNG_PHILOX, assoc=ASSOC_FORWARD, has_time=False, threads=1,
           want_fixed=True):
    """Returns (double sums [H,W,3], fixed-point sums [H,W,3] or None, stats dict)."""
    d = scene.desc()
    H, W = params.height, params.width
    accum = np.zeros((H, W, 3), dtype=np.float64)
    fixed = np.zeros((H, W, 3), dtype=np.int64) if want_fixed else None
    stats = np.zeros(3, dtype=np.uint64)
    rc = lib().orc_render(C.byref(d), C.byref(cam), C.byref(params), rng, assoc, int(has_time), threads,
                          accum.ctypes.data, fixed.ctypes.data if want_fixed else None, stats.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"orc_render failed: {rc}")
    return accum, fixed, {"rays": int(stats[0]), "nan_samples": int(stats[1]), "draws": int(stats[2])}


def render_rows(scene: abi.Scene, cam, params, row_begin, row_end, has_time=False):
    """Fixed-point sums (Philox draws, forward association) of image rows [row_begin, row_end) of the
    full frame described by `params`; rows of a frame are independent of each other."""
    d = scene.desc()
    fixed = np.zeros((row_end - row_begin, params.width, 3), dtype=np.int64)
    rc = lib().orc_render_rows(C.byref(d), C.byref(cam), C.byref(params), row_begin, row_end, int(has_time),
                               fixed.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"orc_render_rows failed: {rc}")
    return fixed


def first_hit(scene: abi.Scene, cam, W, H):
    d = scene.desc()
    ids = np.zeros((H, W), dtype=np.int32)
    rc = lib().orc_first_hit(C.byref(d), C.byref(cam), W, H, ids.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"orc_first_hit failed: {rc}")
    return ids


def write_color(accum, spp):
    H, W, _ = accum.shape
    out = np.zeros((H, W, 3), dtype=np.uint8)
    lib().orc_write_color(np.ascontiguousarray(accum).ctypes.data, W, H, spp, out.ctypes.data)
    return out


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return tuple(int(x) for x in o)
