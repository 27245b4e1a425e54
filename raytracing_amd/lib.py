"""ctypes binding of libraytracing_amd.so (the HIP library behind include/raytracing_amd.h).

There is no CPU fallback: if the shared library is missing, or no GPU is visible when a
context is created, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_LIB_PATH selects another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("RT_LIB_PATH") or os.path.join(_HERE, "libraytracing_amd.so")

_lib = None


class RtError(RuntimeError):
    def __init__(self, code, what, detail=""):
        super().__init__(f"{what}: error {code} ({detail})")
        self.code = code


def load():
    """Load the HIP library (built in-tree by `make -C raytracing_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C raytracing_amd/csrc` "
                          "(there is no CPU fallback)")
    # PyTorch ships its own copy of the HIP runtime under the system's soname.  A process that hands torch tensors to
    # this library (tests, bench.py) must use ONE copy: if torch is installed it is imported first, so that the
    # library's libamdhip64 dependency resolves to the copy torch has already loaded (loaded the other way round,
    # torch finds no GPU).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    ctx_p = C.c_void_p
    L.rt_get_device_count.restype = C.c_int
    L.rt_context_create.restype = C.c_int
    L.rt_context_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(ctx_p)]
    L.rt_context_destroy.restype = None
    L.rt_context_destroy.argtypes = [ctx_p]
    L.rt_scene_upload.restype = C.c_int
    L.rt_scene_upload.argtypes = [ctx_p, C.POINTER(abi.rt_scene_desc)]
    L.rt_render.restype = C.c_int
    L.rt_render.argtypes = [ctx_p, C.POINTER(abi.rt_camera_desc), C.POINTER(abi.rt_render_params), C.c_void_p,
                            C.POINTER(abi.rt_stats)]
    L.rt_render_device.restype = C.c_int
    L.rt_render_device.argtypes = [ctx_p, C.POINTER(abi.rt_camera_desc), C.POINTER(abi.rt_render_params), C.c_void_p]
    L.rt_context_synchronize.restype = C.c_int
    L.rt_context_synchronize.argtypes = [ctx_p]
    L.rt_render_check.restype = C.c_int
    L.rt_render_check.argtypes = [ctx_p, C.POINTER(abi.rt_camera_desc), C.POINTER(abi.rt_render_params)]
    L.rt_render_noalloc.restype = C.c_int
    L.rt_render_noalloc.argtypes = [ctx_p, C.POINTER(abi.rt_camera_desc), C.POINTER(abi.rt_render_params), C.c_void_p]
    L.rt_read_sums.restype = C.c_int
    L.rt_read_sums.argtypes = [ctx_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.rt_context_set_exact_only.restype = C.c_int
    L.rt_context_set_exact_only.argtypes = [ctx_p, C.c_int]
    L.rt_context_set_force_graph.restype = C.c_int
    L.rt_context_set_force_graph.argtypes = [ctx_p, C.c_int]
    L.rt_context_kernels_for.restype = C.c_int
    L.rt_context_kernels_for.argtypes = [ctx_p, C.POINTER(abi.rt_camera_desc)]
    L.rt_context_take_stats.restype = C.c_int
    L.rt_context_take_stats.argtypes = [ctx_p, C.POINTER(abi.rt_stats)]
    L.rt_first_hit.restype = C.c_int
    L.rt_first_hit.argtypes = [ctx_p, C.POINTER(abi.rt_camera_desc), C.c_int, C.c_int, C.c_void_p]
    grp_p = C.c_void_p
    L.rt_split_samples.restype = None
    L.rt_split_samples.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.rt_group_create.restype = C.c_int
    L.rt_group_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(grp_p)]
    L.rt_comm_id_create.restype = C.c_int
    L.rt_comm_id_create.argtypes = [C.POINTER(abi.rt_comm_id)]
    L.rt_group_create_rank.restype = C.c_int
    L.rt_group_create_rank.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(abi.rt_comm_id), C.c_int, C.POINTER(grp_p)]
    L.rt_group_destroy.restype = None
    L.rt_group_destroy.argtypes = [grp_p]
    for name in ("rt_group_size", "rt_group_local_count", "rt_group_frame_slots"):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [grp_p]
    L.rt_group_context.restype = ctx_p
    L.rt_group_context.argtypes = [grp_p, C.c_int, C.c_int]
    L.rt_group_scene_upload.restype = C.c_int
    L.rt_group_scene_upload.argtypes = [grp_p, C.POINTER(abi.rt_scene_desc)]
    L.rt_group_render.restype = C.c_int
    L.rt_group_render.argtypes = [grp_p, C.POINTER(abi.rt_camera_desc), C.POINTER(abi.rt_render_params), C.c_void_p,
                                  C.POINTER(abi.rt_stats)]
    L.rt_group_render_async.restype = C.c_int
    L.rt_group_render_async.argtypes = [grp_p, C.POINTER(abi.rt_camera_desc), C.POINTER(abi.rt_render_params), C.c_int]
    L.rt_group_wait.restype = C.c_int
    L.rt_group_wait.argtypes = [grp_p, C.c_int]
    L.rt_group_reserve.restype = C.c_int
    L.rt_group_reserve.argtypes = [grp_p, C.c_int, C.c_int]
    L.rt_group_set_fail_next.restype = C.c_int
    L.rt_group_set_fail_next.argtypes = [grp_p, C.c_int]
    L.rt_group_read.restype = C.c_int
    L.rt_group_read.argtypes = [grp_p, C.c_int, C.c_void_p]
    L.rt_group_device_sums.restype = C.c_void_p
    L.rt_group_device_sums.argtypes = [grp_p, C.c_int, C.c_int]
    L.rt_group_take_stats.restype = C.c_int
    L.rt_group_take_stats.argtypes = [grp_p, C.POINTER(abi.rt_stats)]
    L.rt_group_last_error_detail.restype = C.c_char_p
    L.rt_group_last_error_detail.argtypes = [grp_p]
    L.rt_camera_init.restype = None
    L.rt_camera_init.argtypes = [C.POINTER(abi.rt_camera_desc), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                 C.POINTER(C.c_double)] + [C.c_double] * 6
    L.rt_write_color.restype = None
    L.rt_write_color.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.rt_write_ppm.restype = C.c_int
    L.rt_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
    L.rt_error_string.restype = C.c_char_p
    L.rt_error_string.argtypes = [C.c_int]
    L.rt_last_error_detail.restype = C.c_char_p
    L.rt_last_error_detail.argtypes = [ctx_p]
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    "rt_get_device_count", "rt_context_create", "rt_context_destroy", "rt_scene_upload", "rt_render",
    "rt_render_device", "rt_context_synchronize", "rt_render_check", "rt_context_set_exact_only", "rt_context_set_force_graph",
    "rt_context_take_stats", "rt_context_kernels_for",
    "rt_first_hit", "rt_camera_init", "rt_write_color", "rt_write_ppm", "rt_error_string", "rt_last_error_detail",
    "rt_render_noalloc", "rt_read_sums", "rt_split_samples", "rt_group_create", "rt_comm_id_create", "rt_group_create_rank", "rt_group_destroy",
    "rt_group_size", "rt_group_local_count", "rt_group_frame_slots", "rt_group_context", "rt_group_scene_upload",
    "rt_group_render", "rt_group_render_async", "rt_group_wait", "rt_group_read", "rt_group_device_sums",
    "rt_group_take_stats", "rt_group_last_error_detail", "rt_group_reserve", "rt_group_set_fail_next",
]


def camera(lookfrom, lookat, vup, vfov, aspect, aperture, focus, time0=0.0, time1=0.0) -> abi.rt_camera_desc:
    """camera::camera of the reference, computed on the host by the library."""
    cam = abi.rt_camera_desc()
    v = lambda a: (C.c_double * 3)(*[float(x) for x in a])
    load().rt_camera_init(C.byref(cam), v(lookfrom), v(lookat), v(vup), vfov, aspect, aperture, focus, time0, time1)
    return cam


def write_color(accum, spp) -> np.ndarray:
    H, W, _ = accum.shape
    out = np.zeros((H, W, 3), dtype=np.uint8)
    a = np.ascontiguousarray(accum, dtype=np.float64)
    load().rt_write_color(a.ctypes.data, W, H, int(spp), out.ctypes.data)
    return out


def write_ppm(path, rgb) -> None:
    H, W, _ = rgb.shape
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    rc = load().rt_write_ppm(os.fsencode(path), a.ctypes.data, W, H)
    if rc != 0:
        raise RtError(rc, "rt_write_ppm")


def stats_dict(st: abi.rt_stats) -> dict:
    return {"samples": int(st.samples), "rays": int(st.rays), "nan_samples": int(st.nan_samples),
            "overflow_rays": int(st.overflow_rays), "kernel_launches": int(st.kernel_launches),
            "kernel_ms": float(st.kernel_ms)}


def split_samples(spp: int, index: int, n: int):
    """rt_split_samples: samples [begin, begin + count) of member `index` of `n` (host arithmetic, no GPU needed)."""
    b, c = C.c_int(), C.c_int()
    load().rt_split_samples(int(spp), int(index), int(n), C.byref(b), C.byref(c))
    return b.value, c.value


def comm_id_create() -> bytes:
    """rt_comm_id_create: the 128 bytes rank 0 hands to every rank of a one-process-per-GPU group."""
    cid = abi.rt_comm_id()
    L = load()
    rc = L.rt_comm_id_create(C.byref(cid))
    if rc != 0:
        raise RtError(rc, "rt_comm_id_create", L.rt_group_last_error_detail(None).decode())
    return bytes(C.string_at(C.addressof(cid), abi.RT_COMM_ID_BYTES))


class Context:
    """One GPU, one scene at a time; mirrors the rt_context_* entry points."""

    def __init__(self, device: int = 0, stream=None):
        self._L = load()
        self._h = C.c_void_p()
        rc = self._L.rt_context_create(int(device), stream, C.byref(self._h))
        if rc != 0:
            raise RtError(rc, "rt_context_create", self._L.rt_last_error_detail(None).decode())
        self._scene = None

    def _check(self, rc, what):
        if rc != 0:
            raise RtError(rc, what, self._L.rt_last_error_detail(self._h).decode())

    def close(self):
        if self._h:
            self._L.rt_context_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, scene: abi.Scene):
        d = scene.desc()
        self._check(self._L.rt_scene_upload(self._h, C.byref(d)), "rt_scene_upload")
        self._scene = scene

    def set_exact_only(self, on: bool):
        self._check(self._L.rt_context_set_exact_only(self._h, int(bool(on))), "rt_context_set_exact_only")

    def set_force_graph(self, on: bool):
        """Takes effect at the next upload()."""
        self._check(self._L.rt_context_set_force_graph(self._h, int(bool(on))), "rt_context_set_force_graph")

    def kernels_for(self, cam) -> int:
        """abi.RT_KERNELS_* : the kernel family that renders the uploaded world for this camera."""
        return int(self._L.rt_context_kernels_for(self._h, C.byref(cam)))

    def render(self, cam, params):
        """-> (double colour sums [H,W,3], stats dict)"""
        out = np.zeros((params.height, params.width, 3), dtype=np.float64)
        st = abi.rt_stats()
        self._check(self._L.rt_render(self._h, C.byref(cam), C.byref(params), out.ctypes.data, C.byref(st)),
                    "rt_render")
        return out, stats_dict(st)

    def render_device(self, cam, params, d_accum_ptr: int):
        """Enqueue a render that ADDS fixed-point sums into device memory at `d_accum_ptr`.  One writer at a time: calls
        that add to the same buffer must be ordered (one stream, or synchronised) -- from 32 samples per pixel
        on, a pixel's sums reach memory by a plain load, add and store, not by atomics."""
        self._check(self._L.rt_render_device(self._h, C.byref(cam), C.byref(params), C.c_void_p(d_accum_ptr)),
                    "rt_render_device")

    def render_noalloc(self, cam, params, d_out_ptr: int):
        """rt_render_noalloc: enqueue a render that leaves DOUBLE sums in device memory at `d_out_ptr`.  All contexts of
        one GPU share that GPU's scratch frame: calls that overlap in time on one GPU (different streams) race."""
        self._check(self._L.rt_render_noalloc(self._h, C.byref(cam), C.byref(params), C.c_void_p(d_out_ptr)),
                    "rt_render_noalloc")

    def synchronize(self):
        self._check(self._L.rt_context_synchronize(self._h), "rt_context_synchronize")

    def render_check(self, cam, params) -> int:
        """rt_render_check: the code rt_render_device would return for these arguments (0 = accepted); enqueues nothing."""
        return int(self._L.rt_render_check(self._h, C.byref(cam), C.byref(params)))

    def read_sums(self, d_accum_ptr: int, width, height) -> np.ndarray:
        """rt_read_sums: device fixed-point sums -> host doubles [H,W,3] (waits for the context's stream)."""
        out = np.zeros((height, width, 3), dtype=np.float64)
        self._check(self._L.rt_read_sums(self._h, C.c_void_p(d_accum_ptr), int(width), int(height), out.ctypes.data),
                    "rt_read_sums")
        return out

    def take_stats(self) -> dict:
        st = abi.rt_stats()
        self._check(self._L.rt_context_take_stats(self._h, C.byref(st)), "rt_context_take_stats")
        return stats_dict(st)

    def first_hit(self, cam, width, height) -> np.ndarray:
        ids = np.zeros((height, width), dtype=np.int32)
        self._check(self._L.rt_first_hit(self._h, C.byref(cam), int(width), int(height), ids.ctypes.data),
                    "rt_first_hit")
        return ids


class _BorrowedContext(Context):
    """A context owned by a group: same methods, never destroyed from here."""

    def __init__(self, handle):
        self._L = load()
        self._h = C.c_void_p(handle)
        self._scene = None

    def close(self):
        self._h = C.c_void_p()


class Group:
    """Several GPUs rendering one frame; mirrors the rt_group_* entry points.

    Group(devices=[0, 1, ...])                      one process driving all of them (ncclCommInitAll)
    Group(device=d, rank=r, world=n, comm_id=b)     one process per GPU (ncclCommInitRank); `b` from comm_id_create()
                                                    on rank 0, handed to every rank by the caller
    """

    def __init__(self, devices=None, *, device=None, rank=None, world=None, comm_id=None, frame_slots=1):
        self._L = load()
        self._h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int * len(devices))(*[int(d) for d in devices])
            rc = self._L.rt_group_create(arr, len(devices), int(frame_slots), C.byref(self._h))
            what = "rt_group_create"
        else:
            cid = abi.rt_comm_id()
            C.memmove(C.addressof(cid), comm_id, abi.RT_COMM_ID_BYTES)
            rc = self._L.rt_group_create_rank(int(device), int(rank), int(world), C.byref(cid), int(frame_slots),
                                              C.byref(self._h))
            what = "rt_group_create_rank"
        if rc != 0:
            raise RtError(rc, what, self._L.rt_group_last_error_detail(None).decode())
        self._scene = None

    def _check(self, rc, what):
        if rc != 0:
            raise RtError(rc, what, self._L.rt_group_last_error_detail(self._h).decode())

    def close(self):
        if self._h:
            self._L.rt_group_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self) -> int:
        return int(self._L.rt_group_size(self._h))

    @property
    def local_count(self) -> int:
        return int(self._L.rt_group_local_count(self._h))

    @property
    def frame_slots(self) -> int:
        return int(self._L.rt_group_frame_slots(self._h))

    def context(self, local_index=0, slot=0) -> Context:
        h = self._L.rt_group_context(self._h, int(local_index), int(slot))
        if not h:
            raise IndexError("no such member / slot")
        return _BorrowedContext(h)

    def upload(self, scene: abi.Scene):
        d = scene.desc()
        self._check(self._L.rt_group_scene_upload(self._h, C.byref(d)), "rt_group_scene_upload")
        self._scene = scene

    def render(self, cam, params):
        """-> (double colour sums [H,W,3] of the whole frame, stats dict of the local members)"""
        out = np.zeros((params.height, params.width, 3), dtype=np.float64)
        st = abi.rt_stats()
        self._check(self._L.rt_group_render(self._h, C.byref(cam), C.byref(params), out.ctypes.data, C.byref(st)),
                    "rt_group_render")
        return out, stats_dict(st)

    def render_async(self, cam, params, slot=0):
        self._check(self._L.rt_group_render_async(self._h, C.byref(cam), C.byref(params), int(slot)),
                    "rt_group_render_async")

    def wait(self, slot=0):
        self._check(self._L.rt_group_wait(self._h, int(slot)), "rt_group_wait")

    def reserve(self, width, height):
        """rt_group_reserve: allocate every slot's buffers for frames of this size now."""
        self._check(self._L.rt_group_reserve(self._h, int(width), int(height)), "rt_group_reserve")

    def set_fail_next(self, local_index=0):
        """Validation: the next render_async treats this local member's launch as failed."""
        self._check(self._L.rt_group_set_fail_next(self._h, int(local_index)), "rt_group_set_fail_next")

    def read(self, width, height, slot=0) -> np.ndarray:
        out = np.zeros((height, width, 3), dtype=np.float64)
        self._check(self._L.rt_group_read(self._h, int(slot), out.ctypes.data), "rt_group_read")
        return out

    def device_sums(self, local_index=0, slot=0) -> int:
        return int(self._L.rt_group_device_sums(self._h, int(local_index), int(slot)) or 0)

    def take_stats(self) -> dict:
        st = abi.rt_stats()
        self._check(self._L.rt_group_take_stats(self._h, C.byref(st)), "rt_group_take_stats")
        return stats_dict(st)
