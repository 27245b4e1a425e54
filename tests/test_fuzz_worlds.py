"""Randomly generated worlds through the C ABI against the oracle, bit for bit: every node kind of books 1 and 2 in
random arrangements (top-level primitives, checker / Perlin / image textures, instance chains, boxes, BVH subtrees of spheres and boxes -- evaluated as
leaf sets when RT_SUBTREE_SET_MIN allows --, lists inside the graph, media), random materials and both sky modes."""
import math

import numpy as np
import pytest

from raytracing_amd import abi, lib
from tests import oracle_binding as ob
from tests.conftest import assert_sums_match, gold, scene_calls_libm

pytestmark = pytest.mark.gpu
FRAC = 2.0 ** abi.RT_ACCUM_FRAC_BITS


class WorldBuilder:
    def __init__(self, rng):
        self.rng = rng
        self.nodes = []
        self.children = []
        self.tex = [(abi.RT_TEX_SOLID, -1, -1, (0.8, 0.8, 0.8, 0.0))]
        self.mat = []
        # one Perlin table (B2/perlin.h:10-21: unit vectors and three permutations) and one small image
        self.perlin = np.zeros(1, abi.PERLIN_DTYPE)
        v = rng.uniform(-1, 1, (256, 3))
        self.perlin["ranvec"][0] = v / np.linalg.norm(v, axis=1, keepdims=True)
        for name in ("perm_x", "perm_y", "perm_z"):
            self.perlin[name][0] = rng.permutation(256)
        self.image = rng.integers(0, 256, (9, 16, 3), dtype=np.uint8)

    # ---- materials
    def solid(self, rgb):
        self.tex.append((abi.RT_TEX_SOLID, -1, -1, (*rgb, 0.0)))
        return len(self.tex) - 1

    def material(self, allow_light=True):
        r = self.rng
        kind = r.choice(["lambert", "checker", "noise", "image", "metal", "glass"] + (["light"] if allow_light else []))
        if kind == "lambert":
            self.mat.append((abi.RT_MAT_LAMBERTIAN, self.solid(r.uniform(0.1, 0.9, 3)), (0, 0, 0), 0.0))
        elif kind == "checker":
            a, b = self.solid(r.uniform(0.1, 0.9, 3)), self.solid(r.uniform(0.1, 0.9, 3))
            self.tex.append((abi.RT_TEX_CHECKER, a, b, (0, 0, 0, 0)))
            self.mat.append((abi.RT_MAT_LAMBERTIAN, len(self.tex) - 1, (0, 0, 0), 0.0))
        elif kind == "noise":
            self.tex.append((abi.RT_TEX_NOISE, 0, -1, (float(r.uniform(1, 6)), 0, 0, 0)))
            self.mat.append((abi.RT_MAT_LAMBERTIAN, len(self.tex) - 1, (0, 0, 0), 0.0))
        elif kind == "image":
            self.tex.append((abi.RT_TEX_IMAGE, 0, -1, (0, 0, 0, 0)))
            self.mat.append((abi.RT_MAT_LAMBERTIAN, len(self.tex) - 1, (0, 0, 0), 0.0))
        elif kind == "metal":
            self.mat.append((abi.RT_MAT_METAL, -1, tuple(r.uniform(0.5, 1.0, 3)), float(r.uniform(0, 0.5))))
        elif kind == "glass":
            self.mat.append((abi.RT_MAT_DIELECTRIC, -1, (0, 0, 0), 1.5))
        else:
            self.mat.append((abi.RT_MAT_DIFFUSE_LIGHT, self.solid(r.uniform(1.0, 4.0, 3)), (0, 0, 0), 0.0))
        return len(self.mat) - 1

    # ---- nodes; every method returns (node id, (lo, hi) bounding box in the node's own frame)
    def node(self, kind, material=-1, child0=0, child1=0, p=()):
        q = np.zeros(12)
        q[:len(p)] = p
        self.nodes.append((kind, material, child0, child1, q))
        return len(self.nodes) - 1

    def sphere(self, c, r):
        c = np.asarray(c, float)
        return self.node(abi.RT_NODE_SPHERE, self.material(), p=(*c, r)), (c - r, c + r)

    def moving_sphere(self, c0, c1, r):
        c0, c1 = np.asarray(c0, float), np.asarray(c1, float)
        i = self.node(abi.RT_NODE_MOVING_SPHERE, self.material(), p=(*c0, *c1, 0.0, 1.0, r))
        a0 = c0 + ((0.0 - 0.0) / (1.0 - 0.0)) * (c1 - c0)
        a1 = c0 + ((1.0 - 0.0) / (1.0 - 0.0)) * (c1 - c0)
        return i, (np.minimum(a0 - r, a1 - r), np.maximum(a0 + r, a1 + r))

    def box(self, lo, hi):
        (x0, y0, z0), (x1, y1, z1) = lo, hi
        m = self.material(allow_light=False)
        sides = [(abi.RT_NODE_XY_RECT, (x0, x1, y0, y1, z1)), (abi.RT_NODE_XY_RECT, (x0, x1, y0, y1, z0)),
                 (abi.RT_NODE_XZ_RECT, (x0, x1, z0, z1, y1)), (abi.RT_NODE_XZ_RECT, (x0, x1, z0, z1, y0)),
                 (abi.RT_NODE_YZ_RECT, (y0, y1, z0, z1, x1)), (abi.RT_NODE_YZ_RECT, (y0, y1, z0, z1, x0))]
        ids = [self.node(k, m, p=p) for k, p in sides]
        first = len(self.children)
        self.children.extend(ids)
        return self.node(abi.RT_NODE_BOX, -1, first, 6, p=(x0, y0, z0, x1, y1, z1)), (np.array(lo, float), np.array(hi, float))

    def rect(self):
        r = self.rng
        kind = r.choice([abi.RT_NODE_XY_RECT, abi.RT_NODE_XZ_RECT, abi.RT_NODE_YZ_RECT])
        a0, b0 = r.uniform(-3, 1, 2)
        a1, b1 = a0 + r.uniform(0.5, 3), b0 + r.uniform(0.5, 3)
        k = r.uniform(-3, 4)
        i = self.node(int(kind), self.material(), p=(a0, a1, b0, b1, k))
        return i, (np.full(3, -10.0), np.full(3, 10.0))

    def random_leaf(self, boxes=True, moving=True):
        r = self.rng
        c = r.uniform(-3, 3, 3) * (1, 0.3, 1) + (0, 0.6, 0)
        what = r.choice(["sphere", "sphere", "box" if boxes else "sphere", "moving" if moving else "sphere"])
        if what == "box":
            lo = c - r.uniform(0.1, 0.5, 3)
            return self.box(tuple(lo), tuple(lo + r.uniform(0.3, 1.0, 3)))
        if what == "moving":
            return self.moving_sphere(c, c + r.uniform(-0.4, 0.4, 3), float(r.uniform(0.15, 0.5)))
        return self.sphere(c, float(r.uniform(0.15, 0.6)))

    def bvh(self, leaves):
        """A tree over `leaves` = [(id, box)], split at random points; boxes as bvh_node's constructor computes them."""
        if len(leaves) == 1:
            return leaves[0]
        cut = int(self.rng.integers(1, len(leaves)))
        (l, lb), (r, rb) = self.bvh(leaves[:cut]), self.bvh(leaves[cut:])
        lo, hi = np.minimum(lb[0], rb[0]), np.maximum(lb[1], rb[1])
        return self.node(abi.RT_NODE_BVH, -1, l, r, p=(*lo, *hi)), (lo, hi)

    def instance(self, child):
        r = self.rng
        i, box = child
        if r.random() < 0.6:
            th = math.radians(float(r.uniform(-60, 60)))
            i = self.node(abi.RT_NODE_ROTATE_Y, -1, i, 0, p=(math.sin(th), math.cos(th)))
        off = r.uniform(-1.5, 1.5, 3) * (1, 0.2, 1)
        return self.node(abi.RT_NODE_TRANSLATE, -1, i, 0, p=tuple(off)), (np.full(3, -12.0), np.full(3, 12.0))

    def medium(self):
        r = self.rng
        boundary = self.sphere(r.uniform(-2, 2, 3) * (1, 0.2, 1) + (0, 0.8, 0), float(r.uniform(0.5, 1.2))) if r.random() < 0.5 \
            else self.box((-1.0, 0.0, -1.0), (0.5, 1.2, 0.6))
        b = boundary if r.random() < 0.5 else self.instance(boundary)
        self.mat.append((abi.RT_MAT_ISOTROPIC, self.solid(r.uniform(0.2, 1.0, 3)), (0, 0, 0), 0.0))
        return self.node(abi.RT_NODE_MEDIUM, len(self.mat) - 1, b[0], 0, p=(-1.0 / float(r.uniform(0.3, 2.0)),)), b[1]

    def finish(self, top, sky):
        first = len(self.children)
        self.children.extend(i for i, _ in top)
        root = self.node(abi.RT_NODE_LIST, -1, first, len(top))
        nodes = np.zeros(len(self.nodes), abi.NODE_DTYPE)
        for i, (kind, m, c0, c1, p) in enumerate(self.nodes):
            nodes[i] = (kind, m, c0, c1, p)
        mats = np.zeros(len(self.mat), abi.MATERIAL_DTYPE)
        for i, (kind, t, alb, prm) in enumerate(self.mat):
            mats[i] = (kind, t, alb, prm)
        texs = np.zeros(len(self.tex), abi.TEXTURE_DTYPE)
        for i, (kind, a, b, p) in enumerate(self.tex):
            texs[i] = (kind, a, b, 0, p)
        return abi.Scene(root, nodes, np.array(self.children, dtype=np.int32), mats, texs, self.perlin,
                         images=[np.ascontiguousarray(self.image)], sky_mode=sky, background=(0.6, 0.7, 0.9))


def random_world(seed):
    rng = np.random.default_rng(seed)
    w = WorldBuilder(rng)
    top = [w.sphere((0, -1000, 0), 1000.0)]
    for _ in range(int(rng.integers(4, 9))):
        what = rng.choice(["leaf", "rect", "instance", "bvh_static", "bvh_any", "bvh_instanced", "list", "medium"])
        if what == "leaf":
            top.append(w.random_leaf())
        elif what == "rect":
            top.append(w.rect())
        elif what == "instance":
            top.append(w.instance(w.random_leaf(moving=False)))
        elif what == "bvh_static":  # a leaf set when RT_SUBTREE_SET_MIN allows
            top.append(w.bvh([w.random_leaf(moving=False) for _ in range(int(rng.integers(2, 12)))]))
        elif what == "bvh_any":     # moving spheres keep the tree
            top.append(w.bvh([w.random_leaf() for _ in range(int(rng.integers(2, 8)))]))
        elif what == "bvh_instanced":
            top.append(w.instance(w.bvh([w.random_leaf(moving=False) for _ in range(int(rng.integers(2, 10)))])))
        elif what == "list":
            kids = [w.random_leaf() for _ in range(int(rng.integers(1, 4)))]
            first = len(w.children)
            w.children.extend(i for i, _ in kids)
            top.append((w.node(abi.RT_NODE_LIST, -1, first, len(kids)), None))
        else:
            top.append(w.medium())
    return w.finish(top, int(rng.choice([abi.RT_SKY_GRADIENT, abi.RT_SKY_BACKGROUND])))


@pytest.mark.parametrize("seed", range(32))
def test_random_world_matches_oracle(seed, monkeypatch):
    scene = random_world(1000 + seed)
    mk = lambda f: f((7, 3.5, 8), (0, 0.7, 0), (0, 1, 0), 32.0, 4.0 / 3.0, 0.05, 10.0, 0.0, 1.0)
    cam, ocam = mk(lib.camera), mk(ob.camera)
    p = abi.render_params(56, 42, 4, 50, seed=seed)
    _, fixed, ost = ob.render(scene, ocam, p, rng=ob.RNG_PHILOX, assoc=ob.ASSOC_FORWARD, has_time=True, threads=16)
    want = fixed.astype(np.float64) / FRAC
    want_ids = ob.first_hit(scene, ocam, 72, 54)
    for set_min in (2, 1000):  # BVH subtrees of static leaves as leaf sets / walked as trees
        monkeypatch.setenv("RT_SUBTREE_SET_MIN", str(set_min))
        c = lib.Context(0)
        try:
            c.upload(scene)
            got, st = c.render(cam, p)
            ids = c.first_hit(cam, 72, 54)
            c.set_exact_only(True)  # leaf sets without their float screens
            exact, _ = c.render(cam, p)
        finally:
            c.close()
        assert np.array_equal(got, exact), (seed, set_min)
        assert st["rays"] == ost["rays"] and st["nan_samples"] == ost["nan_samples"], (seed, set_min)
        assert_sums_match(got, want, scene_calls_libm(scene), (seed, set_min))
        assert np.array_equal(ids, want_ids), (seed, set_min)


def test_rays_that_overflow_walk_a_mixed_subtree_as_the_reference_does(monkeypatch):
    """A lens radius of 1e300 gives every ray an origin and a direction of that size: |d|^2 overflows, a sphere's
    discriminant is inf - inf = NaN and the sphere "hit" at t = NaN, while a box side is still met at a finite t -- which
    leaf ends up as the hit then hangs on the order of the tests and on the boxes of the bvh_nodes in between.  A leaf set
    that holds spheres AND boxes cannot tell from its two flat lists, so such rays walk the subtree itself (world 40516 of
    tools/fuzz_sweep.py, found by it: every pixel differed)."""
    scene = random_world(40516)
    mk = lambda f: f((8, 3.5, 8), (0, 0.7, 0), (0, 1, 0), 32.0, 4.0 / 3.0, 1.0e300, 10.0, 0.0, 1.0)
    cam, ocam = mk(lib.camera), mk(ob.camera)
    p = abi.render_params(48, 36, 3, 50, seed=40516)
    _, fixed, ost = ob.render(scene, ocam, p, rng=ob.RNG_PHILOX, assoc=ob.ASSOC_FORWARD, has_time=True, threads=16)
    want = fixed.astype(np.float64) / FRAC
    for set_min in (2, 1000):
        monkeypatch.setenv("RT_SUBTREE_SET_MIN", str(set_min))
        with lib.Context(0) as c:
            c.upload(scene)
            got, st = c.render(cam, p)
        assert st["rays"] == ost["rays"], set_min
        assert_sums_match(got, want, scene_calls_libm(scene), set_min)


def random_sphere_world(seed):
    """Spheres and moving spheres only, in random nestings of lists and BVH nodes: runs on the sphere-list kernels."""
    rng = np.random.default_rng(seed)
    w = WorldBuilder(rng)
    w.material = lambda allow_light=True, _m=w.material: _pick_sphere_world_material(w, _m)
    n = int(rng.choice([1, 2, 31, 33, 64, 130, 520]))
    leaves = []
    for _ in range(n):
        c = rng.uniform(-6, 6, 3) * (1, 0.1, 1) + (0, 0.4, 0)
        if rng.random() < 0.4:
            t0 = float(rng.choice([0.0, -1.0]))
            leaf = w.moving_sphere(c, c + rng.uniform(-0.5, 0.5, 3), float(rng.uniform(0.1, 0.45)))
            i = leaf[0]
            kind, m, c0, c1, p = w.nodes[i]
            p[6], p[7] = t0, t0 + float(rng.choice([1.0, 3.0]))
            c0v, c1v, r = p[:3], p[3:6], p[8]
            a0 = c0v + ((p[6] - p[6]) / (p[7] - p[6])) * (c1v - c0v)
            a1 = c0v + ((p[7] - p[6]) / (p[7] - p[6])) * (c1v - c0v)
            leaf = (i, (np.minimum(a0 - r, a1 - r), np.maximum(a0 + r, a1 + r)))
        else:
            leaf = w.sphere(c, float(rng.uniform(0.1, 0.5)))
        leaves.append(leaf)
    top = [w.sphere((0, -1000, 0), 1000.0)]
    while leaves:
        take = int(rng.integers(1, len(leaves) + 1))
        group, leaves = leaves[:take], leaves[take:]
        how = rng.choice(["bvh", "list", "each"])
        if how == "bvh":
            top.append(w.bvh(group))
        elif how == "list":
            first = len(w.children)
            w.children.extend(i for i, _ in group)
            top.append((w.node(abi.RT_NODE_LIST, -1, first, len(group)), None))
        else:
            top.extend(group)
    return w.finish(top, int(rng.choice([abi.RT_SKY_GRADIENT, abi.RT_SKY_BACKGROUND])))


def _pick_sphere_world_material(w, general):
    while True:
        m = general(False)
        kind, tex = w.mat[m][0], w.mat[m][1]
        if kind != abi.RT_MAT_LAMBERTIAN or w.tex[tex][0] in (abi.RT_TEX_SOLID, abi.RT_TEX_CHECKER):
            return m
        w.mat.pop()


@pytest.mark.parametrize("seed", range(16))
def test_random_sphere_world_matches_oracle(seed):
    scene = random_sphere_world(2000 + seed)
    rng = np.random.default_rng(seed)
    t0, t1 = [(0.0, 1.0), (0.25, 0.5), (0.0, 0.0), (-0.5, 1.5)][seed % 4]  # the last one leaves some sweeps: graph walk
    aperture = float(rng.choice([0.0, 0.1]))
    mk = lambda f: f((9, 3.0, 9), (0, 0.3, 0), (0, 1, 0), 40.0, 4.0 / 3.0, aperture, 12.0, t0, t1)
    cam, ocam = mk(lib.camera), mk(ob.camera)
    p = abi.render_params(56, 42, 4, 50, seed=seed)
    _, fixed, ost = ob.render(scene, ocam, p, rng=ob.RNG_PHILOX, assoc=ob.ASSOC_FORWARD, has_time=True, threads=16)
    c = lib.Context(0)
    try:
        c.upload(scene)
        got, st = c.render(cam, p)
        ids = c.first_hit(cam, 72, 54)
        c.set_exact_only(True)
        exact, _ = c.render(cam, p)
    finally:
        c.close()
    assert st["rays"] == ost["rays"], seed
    assert_sums_match(got, fixed.astype(np.float64) / FRAC, False, (seed, t0, t1))   # spheres only: exact
    assert np.array_equal(got, exact)
    assert np.array_equal(ids, ob.first_hit(scene, ocam, 72, 54)), seed


def random_is_world(seed):
    """A world for RT_RENDER_IMPORTANCE (ray_color of book 3): no media, a lamp behind a flip_face wrapper, and a
    light list (outside the world graph) of an xz_rect like the lamp and, sometimes, a glass sphere of the world."""
    rng = np.random.default_rng(seed)
    w = WorldBuilder(rng)
    top = [w.sphere((0, -1000, 0), 1000.0)]
    for _ in range(int(rng.integers(3, 8))):
        what = rng.choice(["leaf", "rect", "instance", "bvh_static", "bvh_instanced"])
        if what == "leaf":
            top.append(w.random_leaf(moving=False))
        elif what == "rect":
            top.append(w.rect())
        elif what == "instance":
            top.append(w.instance(w.random_leaf(moving=False)))
        elif what == "bvh_static":
            top.append(w.bvh([w.random_leaf(moving=False) for _ in range(int(rng.integers(2, 10)))]))
        else:
            top.append(w.instance(w.bvh([w.random_leaf(moving=False) for _ in range(int(rng.integers(2, 8)))])))
    # the lamp: an xz_rect emitter seen through flip_face (it emits from its front face only: downwards after the flip)
    x0, z0 = rng.uniform(-2, 0, 2)
    lamp_p = (x0, x0 + float(rng.uniform(1, 2.5)), z0, z0 + float(rng.uniform(1, 2.5)), float(rng.uniform(3.0, 4.5)))
    w.mat.append((abi.RT_MAT_DIFFUSE_LIGHT, w.solid((8.0, 8.0, 8.0)), (0, 0, 0), 0.0))
    lamp = w.node(abi.RT_NODE_XZ_RECT, len(w.mat) - 1, p=lamp_p)
    top.append((w.node(abi.RT_NODE_FLIP_FACE, -1, lamp, 0), None))
    shapes = [w.node(abi.RT_NODE_XZ_RECT, -1, p=lamp_p)]
    if rng.random() < 0.5:
        c, r = rng.uniform(-2, 2, 3) * (1, 0.1, 1) + (0, 0.9, 0), float(rng.uniform(0.3, 0.7))
        w.mat.append((abi.RT_MAT_DIELECTRIC, -1, (0, 0, 0), 1.5))
        top.append((w.node(abi.RT_NODE_SPHERE, len(w.mat) - 1, p=(*c, r)), None))
        shapes.append(w.node(abi.RT_NODE_SPHERE, -1, p=(*c, r)))
    first = len(w.children)
    w.children.extend(shapes)
    lights = w.node(abi.RT_NODE_LIST, -1, first, len(shapes))
    scene = w.finish(top, abi.RT_SKY_BACKGROUND)
    scene.background = (0.0, 0.0, 0.0) if rng.random() < 0.5 else (0.2, 0.25, 0.3)
    scene.lights_plus1 = lights + 1
    return scene


@pytest.mark.parametrize("seed", range(12))
def test_random_world_importance_sampling_matches_oracle(seed, monkeypatch):
    """Book 3's ray_color (parity with the reference unpinned: it has no such code) -- GPU against the oracle."""
    scene = random_is_world(3000 + seed)
    mk = lambda f: f((7, 3.5, 8), (0, 0.7, 0), (0, 1, 0), 32.0, 4.0 / 3.0, 0.0, 10.0, 0.0, 1.0)
    cam, ocam = mk(lib.camera), mk(ob.camera)
    for flags in (abi.RT_RENDER_IMPORTANCE, 0):
        p = abi.render_params(56, 42, 4, 50, seed=seed, flags=flags)
        _, fixed, ost = ob.render(scene, ocam, p, rng=ob.RNG_PHILOX, assoc=ob.ASSOC_FORWARD, has_time=True, threads=16)
        for set_min in (2, 1000):
            monkeypatch.setenv("RT_SUBTREE_SET_MIN", str(set_min))
            c = lib.Context(0)
            try:
                c.upload(scene)
                got, st = c.render(cam, p)
            finally:
                c.close()
            assert st["rays"] == ost["rays"] and st["nan_samples"] == ost["nan_samples"], (seed, flags, set_min)
            assert_sums_match(got, fixed.astype(np.float64) / FRAC, scene_calls_libm(scene, flags), (seed, flags, set_min))


def random_big_set_world(seed):
    """Large leaf sets: a BVH of 70-700 static spheres and boxes (several screening groups and chunks, odd counts),
    by itself or instanced, next to a hollow glass sphere (negative radius), nested instances and a medium whose
    boundary is a list; the camera of the test sometimes sits inside the cloud."""
    rng = np.random.default_rng(seed)
    w = WorldBuilder(rng)
    top = [w.sphere((0, -1000, 0), 1000.0)]
    n = int(rng.choice([70, 129, 257, 513, 700]))
    kind = rng.choice(["spheres", "boxes", "mixed"])
    leaves = []
    for _ in range(n):
        c = rng.uniform(-4, 4, 3) * (1, 0.35, 1) + (0, 1.6, 0)
        if kind == "boxes" or (kind == "mixed" and rng.random() < 0.4):
            lo = c - rng.uniform(0.05, 0.25, 3)
            leaves.append(w.box(tuple(lo), tuple(lo + rng.uniform(0.1, 0.5, 3))))
        else:
            leaves.append(w.sphere(c, float(rng.uniform(0.05, 0.3))))
    cloud = w.bvh(leaves)
    top.append(w.instance(cloud) if rng.random() < 0.5 else cloud)
    # a hollow glass ball: B1's trick of a second sphere with a negative radius inside the first
    c = rng.uniform(-2, 2, 3) * (1, 0, 1) + (0, 0.7, 0)
    w.mat.append((abi.RT_MAT_DIELECTRIC, -1, (0, 0, 0), 1.5))
    glass = len(w.mat) - 1
    top.append((w.node(abi.RT_NODE_SPHERE, glass, p=(*c, 0.7)), None))
    top.append((w.node(abi.RT_NODE_SPHERE, glass, p=(*c, -0.6)), None))
    # instances nested two deep around a leaf: up to four translate / rotate_y levels, the library's limit
    top.append(w.instance(w.instance(w.random_leaf(moving=False))))
    # a medium bounded by a list of two spheres
    kids = [w.sphere(rng.uniform(-3, 3, 3) * (1, 0.1, 1) + (0, 0.9, 0), float(rng.uniform(0.5, 1.0))) for _ in range(2)]
    first = len(w.children)
    w.children.extend(i for i, _ in kids)
    boundary = w.node(abi.RT_NODE_LIST, -1, first, 2)
    w.mat.append((abi.RT_MAT_ISOTROPIC, w.solid(rng.uniform(0.2, 1.0, 3)), (0, 0, 0), 0.0))
    top.append((w.node(abi.RT_NODE_MEDIUM, len(w.mat) - 1, boundary, 0, p=(-1.0 / float(rng.uniform(0.5, 3.0)),)), None))
    return w.finish(top, int(rng.choice([abi.RT_SKY_GRADIENT, abi.RT_SKY_BACKGROUND])))


@pytest.mark.parametrize("seed", range(8))
def test_random_big_set_world_matches_oracle(seed, monkeypatch):
    scene = random_big_set_world(4000 + seed)
    lookfrom = (7, 3.5, 8) if seed % 2 == 0 else (0.3, 1.5, 0.2)  # outside / inside the cloud
    mk = lambda f: f(lookfrom, (0, 1.2, 0) if seed % 2 == 0 else (2, 1.4, 1), (0, 1, 0), 40.0, 4.0 / 3.0, 0.0, 10.0, 0.0, 1.0)
    cam, ocam = mk(lib.camera), mk(ob.camera)
    p = abi.render_params(56, 42, 4, 50, seed=seed)
    _, fixed, ost = ob.render(scene, ocam, p, rng=ob.RNG_PHILOX, assoc=ob.ASSOC_FORWARD, has_time=True, threads=16)
    want_ids = ob.first_hit(scene, ocam, 72, 54)
    for set_min in (64, 100000):
        monkeypatch.setenv("RT_SUBTREE_SET_MIN", str(set_min))
        c = lib.Context(0)
        try:
            c.upload(scene)
            got, st = c.render(cam, p)
            ids = c.first_hit(cam, 72, 54)
            c.set_exact_only(True)
            exact, _ = c.render(cam, p)
        finally:
            c.close()
        assert st["rays"] == ost["rays"], (seed, set_min)
        assert_sums_match(got, fixed.astype(np.float64) / FRAC, scene_calls_libm(scene), (seed, set_min))
        assert np.array_equal(got, exact) and np.array_equal(ids, want_ids), (seed, set_min)


def random_flat_list(seed):
    """A book-1 world for the sphere-list kernels: 1 to ~1100 spheres (one or several screening chunks, odd counts),
    overlapping, some hollow (negative radius), some huge, every material."""
    from tests.conftest import make_sphere_scene
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 5, 31, 32, 33, 97, 256, 487, 511, 513, 1025, 1100]))
    mats = [("lambertian", tuple(rng.uniform(0.1, 0.9, 3))), ("metal", tuple(rng.uniform(0.5, 1.0, 3)), float(rng.uniform(0, 1.2))),
            ("dielectric", 1.5), ("lambertian", tuple(rng.uniform(0.1, 0.9, 3))), ("metal", (0.9, 0.9, 0.9), 0.0), ("dielectric", 2.4)]
    spheres = []
    if rng.random() < 0.8:
        spheres.append((0.0, -1000.0, 0.0, 1000.0, 0))
    while len(spheres) < n:
        c = rng.uniform(-8, 8, 3) * (1, 0.15, 1) + (0, 0.5, 0)
        r = float(rng.choice([0.05, 0.2, 0.2, 0.4, 1.0, 3.0]) * rng.uniform(0.8, 1.2))
        m = int(rng.integers(0, len(mats)))
        spheres.append((*c, r, m))
        if mats[m][0] == "dielectric" and rng.random() < 0.3 and len(spheres) < n:
            spheres.append((*c, -0.9 * r, m))  # hollow glass: B1R's trick
    # one world in five at another scale: 1e12 is beyond what the two-level screen's margins cover, 1e17 beyond the float
    # screen altogether
    scale = float(rng.choice([1.0] * 8 + [1.0e-6, 1.0e12] if n > 600 else [1.0] * 8 + [1.0e-6, 1.0e5, 1.0e12, 1.0e17]))
    spheres = [(x * scale, y * scale, z * scale, r * scale, m) for (x, y, z, r, m) in spheres[:n]]
    return make_sphere_scene(spheres, mats, int(rng.choice([abi.RT_SKY_GRADIENT, abi.RT_SKY_BACKGROUND])), (0.3, 0.4, 0.5)), rng, scale


def flat_list_case(seed):
    """World, camera arguments and render parameters of one random book-1 case: odd image sizes, 1 to 130 samples per
    pixel (from 100 on the pixel-by-pixel pass and the accumulation cache), sample offsets, shallow and deep paths,
    cameras that may sit inside a sphere."""
    scene, rng, scale = random_flat_list(seed)
    W, H = int(rng.integers(2, 70)), int(rng.integers(2, 50))
    spp = int(rng.choice([1, 3, 8, 17, 104, 130]))
    if spp > 100:
        W, H = min(W, 40), min(H, 24)
    depth = int(rng.choice([1, 5, 50]))
    begin = int(rng.choice([0, 0, 7, 1000]))
    lookfrom = tuple(rng.uniform(-9, 9, 3) * (1, 0.2, 1) + (0, 1.5, 0))
    aperture = float(rng.choice([0.0, 0.1, 0.6]))
    vfov = float(rng.choice([20.0, 60.0, 110.0]))
    lens = 1.0e300 if rng.random() < 0.04 else aperture * scale  # a lens that makes the rays overflow
    cam_args = (tuple(x * scale for x in lookfrom), (0, 0.5 * scale, 0), (0, 1, 0), vfov, W / H, lens, 6.0 * scale)
    return scene, cam_args, abi.render_params(W, H, spp, depth, seed=seed * 7919 + 1, sample_begin=begin)


def check_flat_list_case(seed):
    scene, cam_args, p = flat_list_case(seed)
    W, H = p.width, p.height
    cam, ocam = lib.camera(*cam_args), ob.camera(*cam_args)
    _, fixed, ost = ob.render(scene, ocam, p, rng=ob.RNG_PHILOX, assoc=ob.ASSOC_FORWARD, threads=16)
    c = lib.Context(0)
    try:
        c.upload(scene)
        assert c.kernels_for(cam) == abi.RT_KERNELS_SPHERE_LIST
        got, st = c.render(cam, p)
        ids = c.first_hit(cam, W, H)
        c.set_exact_only(True)
        exact, se = c.render(cam, p)
    finally:
        c.close()
    what = (seed, W, H, p.sample_count, p.max_depth, p.sample_begin)
    assert st["rays"] == ost["rays"] == se["rays"] and st["samples"] == W * H * p.sample_count, what
    assert_sums_match(got, fixed.astype(np.float64) / FRAC, False, what)   # a book-1 list: exact
    assert np.array_equal(got, exact), what
    assert np.array_equal(ids, ob.first_hit(scene, ocam, W, H)), what


@pytest.mark.parametrize("seed", range(16))
def test_random_flat_list_matches_oracle(seed):
    """The headline kernels on random lists, against the oracle and against their own exact-only mode."""
    check_flat_list_case(6000 + seed)


def scale_world(scene, s):
    """The same world with every length multiplied by s.  Noise textures become grey: their sine takes a world coordinate
    times a fixed frequency, so on a world 1e4 times larger a last-bit difference of the device libm somewhere up the
    path (a medium's log) no longer disappears in the accumulation format but comes out as 1e-9 of a colour."""
    for t in scene.textures:
        if int(t["kind"]) == abi.RT_TEX_NOISE:
            t["kind"] = abi.RT_TEX_SOLID
            t["p"][:3] = 0.5
    lengths = {abi.RT_NODE_SPHERE: [0, 1, 2, 3], abi.RT_NODE_MOVING_SPHERE: [0, 1, 2, 3, 4, 5, 8],
               abi.RT_NODE_XY_RECT: [0, 1, 2, 3, 4], abi.RT_NODE_XZ_RECT: [0, 1, 2, 3, 4], abi.RT_NODE_YZ_RECT: [0, 1, 2, 3, 4],
               abi.RT_NODE_BOX: [0, 1, 2, 3, 4, 5], abi.RT_NODE_BVH: [0, 1, 2, 3, 4, 5], abi.RT_NODE_TRANSLATE: [0, 1, 2],
               abi.RT_NODE_ROTATE_Y: [3, 4, 5, 6, 7, 8], abi.RT_NODE_MEDIUM: [0]}
    for n in scene.nodes:
        for k in lengths.get(int(n["kind"]), []):
            n["p"][k] *= s
    return scene


@pytest.mark.parametrize("family,seed,scale", [("graph", s, sc) for s in range(6) for sc in (1.0e-4, 1.0e4, 1.0e7)] +
                         [(f, s, sc) for f in ("spheres", "big sets") for s in range(3) for sc in (1.0e-4, 1.0e4, 1.0e7)])
def test_random_world_at_other_scales(family, seed, scale, monkeypatch):
    """Random worlds shrunk and blown up: the margins of the screens follow the world's extent; beyond 8e6 the box sets
    give way to the tree walk, beyond 1e6 the two-level screens to the one-level ones."""
    make = {"graph": random_world, "spheres": random_sphere_world, "big sets": random_big_set_world}[family]
    scene = scale_world(make(1000 + seed), scale)
    mk = lambda f: f((7 * scale, 3.5 * scale, 8 * scale), (0, 0.7 * scale, 0), (0, 1, 0), 32.0, 4.0 / 3.0, 0.05 * scale, 10.0 * scale, 0.0, 1.0)
    cam, ocam = mk(lib.camera), mk(ob.camera)
    p = abi.render_params(48, 36, 3, 50, seed=seed)
    _, fixed, ost = ob.render(scene, ocam, p, rng=ob.RNG_PHILOX, assoc=ob.ASSOC_FORWARD, has_time=True, threads=16)
    for set_min in (2, 1000):
        monkeypatch.setenv("RT_SUBTREE_SET_MIN", str(set_min))
        monkeypatch.setenv("RT_CLUSTER_MIN", "4")
        c = lib.Context(0)
        try:
            c.upload(scene)
            got, st = c.render(cam, p)
            ids = c.first_hit(cam, 60, 45)
        finally:
            c.close()
        assert st["rays"] == ost["rays"], (seed, scale, set_min)
        assert_sums_match(got, fixed.astype(np.float64) / FRAC, scene_calls_libm(scene), (seed, scale, set_min))
        assert np.array_equal(ids, ob.first_hit(scene, ocam, 60, 45)), (seed, scale, set_min)


@pytest.mark.parametrize("scale", [1.0, 3.0e7, 1.0e10])
def test_perlin_noise_beyond_the_int_range(scale):
    """Perlin noise converts floor(coordinate) to int; on a world of 1e8 and more that leaves the range of int, where the
    reference's x86-64 build gets INT_MIN from the conversion (the turbulence doubles the coordinate six times, so
    3e7 overflows in its last octaves only).  Spheres only: nothing up the path calls the libm."""
    tex = np.zeros(2, abi.TEXTURE_DTYPE)
    tex["kind"] = abi.RT_TEX_NOISE
    tex["a"] = 0
    tex["p"][0][0], tex["p"][1][0] = 4.0, 0.3
    mat = np.zeros(2, abi.MATERIAL_DTYPE)
    mat["kind"] = abi.RT_MAT_LAMBERTIAN
    mat["texture"] = [0, 1]
    nodes = np.zeros(4, abi.NODE_DTYPE)
    for i, (c, r, m) in enumerate([((0, -1000, 0), 1000, 0), ((0, 2, 0), 2, 1), ((4, 1, 1), 1, 0)]):
        nodes[i]["kind"], nodes[i]["material"] = abi.RT_NODE_SPHERE, m
        nodes[i]["p"][:4] = (*c, r)
    nodes[3]["kind"], nodes[3]["child0"], nodes[3]["child1"], nodes[3]["material"] = abi.RT_NODE_LIST, 0, 3, -1
    base = abi.load_scene(gold("scene_b2_two_perlin.rtsc"))
    scene = abi.Scene(root=3, nodes=nodes, children=np.arange(3, dtype=np.int32), materials=mat, textures=tex,
                      perlins=base.perlins[:1].copy(), sky_mode=abi.RT_SKY_GRADIENT)
    for n in scene.nodes[:3]:
        n["p"][:4] *= scale
    mk = lambda f: f((13 * scale, 2 * scale, 3 * scale), (0, 0, 0), (0, 1, 0), 30.0, 1.5, 0.0, 10.0 * scale, 0.0, 1.0)
    cam, ocam = mk(lib.camera), mk(ob.camera)
    p = abi.render_params(60, 40, 4, 50, seed=3)
    _, fixed, ost = ob.render(scene, ocam, p, rng=ob.RNG_PHILOX, assoc=ob.ASSOC_FORWARD, has_time=True, threads=16)
    c = lib.Context(0)
    try:
        c.upload(scene)
        c.set_force_graph(True)
        got, st = c.render(cam, p)
    finally:
        c.close()
    assert st["rays"] == ost["rays"]
    assert_sums_match(got, fixed.astype(np.float64) / FRAC, scene_calls_libm(scene), None)
