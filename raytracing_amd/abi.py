"""ctypes mirror of include/raytracing_amd.h (struct layouts, enums, scene files).

This module only describes data; it loads no library.  `raytracing_amd.lib` binds the
entry points of the HIP library to these types.
"""
from __future__ import annotations

import ctypes as C
import struct
from dataclasses import dataclass, field

import numpy as np

RT_ABI_VERSION = 1
RT_ACCUM_FRAC_BITS = 36
RT_ACCUM_SAMPLE_LIMIT = 67108864.0

RT_OK = 0
RT_ERR_INVALID_ARGUMENT = 1
RT_ERR_UNSUPPORTED_SCENE = 2
RT_ERR_NO_DEVICE = 3
RT_ERR_HIP = 4
RT_ERR_NO_SCENE = 5
RT_ERR_OUT_OF_MEMORY = 6
RT_ERR_LIMIT = 7
RT_ERR_COMM = 8
RT_COMM_ID_BYTES = 128

(RT_NODE_SPHERE, RT_NODE_MOVING_SPHERE, RT_NODE_XY_RECT, RT_NODE_XZ_RECT, RT_NODE_YZ_RECT, RT_NODE_BOX,
 RT_NODE_LIST, RT_NODE_BVH, RT_NODE_TRANSLATE, RT_NODE_ROTATE_Y, RT_NODE_MEDIUM, RT_NODE_FLIP_FACE) = range(1, 13)
RT_RENDER_EXACT_ONLY, RT_RENDER_IMPORTANCE = 1, 2
RT_KERNELS_NONE, RT_KERNELS_SPHERE_LIST, RT_KERNELS_SCENE_GRAPH = 0, 1, 2
(RT_MAT_LAMBERTIAN, RT_MAT_METAL, RT_MAT_DIELECTRIC, RT_MAT_DIFFUSE_LIGHT, RT_MAT_ISOTROPIC) = range(1, 6)
(RT_TEX_SOLID, RT_TEX_CHECKER, RT_TEX_NOISE, RT_TEX_IMAGE) = range(1, 5)
RT_SKY_GRADIENT, RT_SKY_BACKGROUND = 0, 1


class rt_node(C.Structure):
    _fields_ = [("kind", C.c_int32), ("material", C.c_int32), ("child0", C.c_int32), ("child1", C.c_int32),
                ("p", C.c_double * 12)]


class rt_material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("texture", C.c_int32), ("albedo", C.c_double * 3), ("param", C.c_double)]


class rt_texture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("reserved", C.c_int32),
                ("p", C.c_double * 4)]


class rt_perlin(C.Structure):
    _fields_ = [("ranvec", (C.c_double * 3) * 256), ("perm_x", C.c_int32 * 256), ("perm_y", C.c_int32 * 256),
                ("perm_z", C.c_int32 * 256)]


class rt_image(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", C.POINTER(C.c_uint8))]


class rt_scene_desc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("root", C.c_int32), ("n_nodes", C.c_int32), ("n_children", C.c_int32),
                ("n_materials", C.c_int32), ("n_textures", C.c_int32), ("n_perlins", C.c_int32),
                ("n_images", C.c_int32),
                ("nodes", C.POINTER(rt_node)), ("children", C.POINTER(C.c_int32)),
                ("materials", C.POINTER(rt_material)), ("textures", C.POINTER(rt_texture)),
                ("perlins", C.POINTER(rt_perlin)), ("images", C.POINTER(rt_image)),
                ("sky_mode", C.c_int32), ("lights_plus1", C.c_int32), ("background", C.c_double * 3)]


class rt_camera_desc(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("lower_left_corner", C.c_double * 3), ("horizontal", C.c_double * 3),
                ("vertical", C.c_double * 3), ("u", C.c_double * 3), ("v", C.c_double * 3), ("w", C.c_double * 3),
                ("lens_radius", C.c_double), ("time0", C.c_double), ("time1", C.c_double)]


class rt_render_params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("sample_begin", C.c_int32),
                ("sample_count", C.c_int32), ("max_depth", C.c_int32), ("flags", C.c_int32), ("seed", C.c_uint64)]


class rt_comm_id(C.Structure):
    _fields_ = [("bytes", C.c_char * RT_COMM_ID_BYTES)]


class rt_stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("nan_samples", C.c_uint64),
                ("overflow_rays", C.c_uint64), ("kernel_launches", C.c_uint64), ("kernel_ms", C.c_double)]


NODE_DTYPE = np.dtype([("kind", "<i4"), ("material", "<i4"), ("child0", "<i4"), ("child1", "<i4"),
                       ("p", "<f8", (12,))])
MATERIAL_DTYPE = np.dtype([("kind", "<i4"), ("texture", "<i4"), ("albedo", "<f8", (3,)), ("param", "<f8")])
TEXTURE_DTYPE = np.dtype([("kind", "<i4"), ("a", "<i4"), ("b", "<i4"), ("reserved", "<i4"), ("p", "<f8", (4,))])
PERLIN_DTYPE = np.dtype([("ranvec", "<f8", (256, 3)), ("perm_x", "<i4", (256,)), ("perm_y", "<i4", (256,)),
                         ("perm_z", "<i4", (256,))])

assert NODE_DTYPE.itemsize == C.sizeof(rt_node) == 112
assert MATERIAL_DTYPE.itemsize == C.sizeof(rt_material) == 40
assert TEXTURE_DTYPE.itemsize == C.sizeof(rt_texture) == 48
assert PERLIN_DTYPE.itemsize == C.sizeof(rt_perlin) == 9216


@dataclass
class Scene:
    """A scene graph held in numpy arrays with the exact memory layout of the C structs."""

    root: int
    nodes: np.ndarray
    children: np.ndarray
    materials: np.ndarray
    textures: np.ndarray
    perlins: np.ndarray
    images: list = field(default_factory=list)  # list of uint8 arrays shaped (h, w, 3)
    sky_mode: int = RT_SKY_GRADIENT
    background: tuple = (0.0, 0.0, 0.0)
    lights_plus1: int = 0  # 1 + index of the list node holding the light shapes (importance sampling), 0 = none

    def desc(self) -> rt_scene_desc:
        """Build an rt_scene_desc that points into this object's arrays (keep `self` alive)."""
        d = rt_scene_desc()
        d.abi_version = RT_ABI_VERSION
        d.root = int(self.root)
        d.n_nodes = len(self.nodes)
        d.n_children = len(self.children)
        d.n_materials = len(self.materials)
        d.n_textures = len(self.textures)
        d.n_perlins = len(self.perlins)
        d.n_images = len(self.images)
        for name, arr, ty in (("nodes", self.nodes, rt_node), ("children", self.children, C.c_int32),
                              ("materials", self.materials, rt_material), ("textures", self.textures, rt_texture),
                              ("perlins", self.perlins, rt_perlin)):
            assert arr.flags["C_CONTIGUOUS"]
            setattr(d, name, C.cast(arr.ctypes.data, C.POINTER(ty)) if len(arr) else None)
        self._image_structs = (rt_image * max(1, len(self.images)))()
        for k, im in enumerate(self.images):
            assert im.dtype == np.uint8 and im.flags["C_CONTIGUOUS"]
            self._image_structs[k].height, self._image_structs[k].width = im.shape[0], im.shape[1]
            self._image_structs[k].rgb = C.cast(im.ctypes.data, C.POINTER(C.c_uint8)) if im.size else None
        d.images = C.cast(self._image_structs, C.POINTER(rt_image)) if self.images else None
        d.sky_mode = int(self.sky_mode)
        d.lights_plus1 = int(self.lights_plus1)
        d.background[:] = [float(x) for x in self.background]
        return d


def load_scene(path) -> Scene:
    """Read an RTSCENE1 file (layout: include/rt_scene_io.h)."""
    if str(path).endswith(".gz"):
        import gzip
        with gzip.open(path, "rb") as fp:
            blob = fp.read()
    else:
        with open(path, "rb") as fp:
            blob = fp.read()
    if blob[:8] != b"RTSCENE1":
        raise ValueError(f"{path}: not an RTSCENE1 file")
    root, n_nodes, n_children, n_mat, n_tex, n_perlin, n_img, sky = struct.unpack_from("<8i", blob, 8)
    off = 8 + 32
    background = struct.unpack_from("<3d", blob, off)
    off += 24

    def take(dtype, n):
        nonlocal off
        a = np.frombuffer(blob, dtype=dtype, count=n, offset=off).copy()
        off += a.nbytes
        return a

    nodes = take(NODE_DTYPE, n_nodes)
    children = take(np.dtype("<i4"), n_children)
    materials = take(MATERIAL_DTYPE, n_mat)
    textures = take(TEXTURE_DTYPE, n_tex)
    perlins = take(PERLIN_DTYPE, n_perlin)
    images = []
    for _ in range(n_img):
        w, h = struct.unpack_from("<2i", blob, off)
        off += 8
        images.append(take(np.uint8, 3 * w * h).reshape(h, w, 3))
    if off != len(blob):
        raise ValueError(f"{path}: {len(blob) - off} trailing bytes")
    return Scene(root, nodes, children, materials, textures, perlins, images, sky & 0xFF, background, (sky & 0xFFFFFFFF) >> 8)


def save_scene(path, s: Scene) -> None:
    with open(path, "wb") as fp:
        fp.write(b"RTSCENE1")
        fp.write(struct.pack("<8i", s.root, len(s.nodes), len(s.children), len(s.materials), len(s.textures),
                             len(s.perlins), len(s.images), (s.sky_mode & 0xFF) | (s.lights_plus1 << 8)))
        fp.write(struct.pack("<3d", *s.background))
        for a in (s.nodes, s.children, s.materials, s.textures, s.perlins):
            fp.write(a.tobytes())
        for im in s.images:
            fp.write(struct.pack("<2i", im.shape[1], im.shape[0]))
            fp.write(im.tobytes())


def render_params(width, height, spp, max_depth=50, seed=1, sample_begin=0, flags=0) -> rt_render_params:
    p = rt_render_params()
    p.width, p.height = int(width), int(height)
    p.sample_begin, p.sample_count = int(sample_begin), int(spp)
    p.max_depth, p.flags, p.seed = int(max_depth), int(flags), int(seed)
    return p
