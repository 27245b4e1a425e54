"""Book-2 worlds that hold nothing but spheres (config 3's bouncing spheres under a BVH) are rendered by the
sphere-list kernels as the list of their leaves.  The results must be those of the graph walk, i.e. of the oracle,
bit for bit; and worlds or cameras for which the list could differ from the reference's tree walk (a BVH box that
does not enclose its leaves, a shutter interval outside a moving sphere's [time0, time1]) must take the graph walk."""
import copy

import numpy as np
import pytest

from raytracing_amd import abi, lib
from tests import oracle_binding as ob
from tests.test_oracle_golden_b2 import camera_b2, load_b2

pytestmark = pytest.mark.gpu
FRAC = 2.0 ** abi.RT_ACCUM_FRAC_BITS


@pytest.fixture(scope="module")
def kat_b2():
    import json
    from tests.conftest import gold
    with open(gold("kat_b2.json")) as fp:
        return json.load(fp)


@pytest.fixture()
def ctx():
    c = lib.Context(0)
    yield c
    c.close()


def oracle_sums(scene, cam, p):
    _, fixed, st = ob.render(scene, cam, p, rng=ob.RNG_PHILOX, assoc=ob.ASSOC_FORWARD, has_time=True, threads=16)
    return fixed.astype(np.float64) / FRAC, st


def nested_world():
    """root list [ground (checker), list [bvh(moving, bvh(metal sphere, moving)), glass sphere], moving sphere]."""
    tex = np.zeros(4, abi.TEXTURE_DTYPE)
    tex["kind"] = [abi.RT_TEX_SOLID, abi.RT_TEX_CHECKER, abi.RT_TEX_SOLID, abi.RT_TEX_SOLID]
    tex["a"][1], tex["b"][1] = 2, 3
    tex["p"][0][:3] = (0.7, 0.3, 0.2)
    tex["p"][2][:3] = (0.2, 0.3, 0.1)
    tex["p"][3][:3] = (0.9, 0.9, 0.9)
    mat = np.zeros(4, abi.MATERIAL_DTYPE)
    mat["kind"] = [abi.RT_MAT_LAMBERTIAN, abi.RT_MAT_LAMBERTIAN, abi.RT_MAT_METAL, abi.RT_MAT_DIELECTRIC]
    mat["texture"] = [0, 1, -1, -1]
    mat["albedo"][2] = (0.8, 0.7, 0.6)
    mat["param"] = [0, 0, 0.3, 1.5]
    nodes = np.zeros(10, abi.NODE_DTYPE)

    def sphere(i, c, r, m):
        nodes[i]["kind"], nodes[i]["material"] = abi.RT_NODE_SPHERE, m
        nodes[i]["p"][:4] = (*c, r)

    def moving(i, c0, c1, t0, t1, r, m):
        nodes[i]["kind"], nodes[i]["material"] = abi.RT_NODE_MOVING_SPHERE, m
        nodes[i]["p"][:9] = (*c0, *c1, t0, t1, r)

    def box(i):
        n = nodes[i]
        if n["kind"] == abi.RT_NODE_SPHERE:
            c, r = n["p"][:3], n["p"][3]
            return c - r, c + r
        if n["kind"] == abi.RT_NODE_MOVING_SPHERE:
            c0, c1, t0, t1, r = n["p"][:3], n["p"][3:6], n["p"][6], n["p"][7], n["p"][8]
            a0 = c0 + ((t0 - t0) / (t1 - t0)) * (c1 - c0)
            a1 = c0 + ((t1 - t0) / (t1 - t0)) * (c1 - c0)
            return np.minimum(a0 - r, a1 - r), np.maximum(a0 + r, a1 + r)
        lo_l, hi_l = box(n["child0"])
        lo_r, hi_r = box(n["child1"])
        return np.minimum(lo_l, lo_r), np.maximum(hi_l, hi_r)

    def bvh(i, left, right):
        nodes[i]["kind"], nodes[i]["material"] = abi.RT_NODE_BVH, -1
        nodes[i]["child0"], nodes[i]["child1"] = left, right
        lo, hi = box(i)
        nodes[i]["p"][:6] = (*lo, *hi)

    children = np.array([1, 2, 3, 4, 5], dtype=np.int32)
    nodes[0]["kind"], nodes[0]["material"], nodes[0]["child0"], nodes[0]["child1"] = abi.RT_NODE_LIST, -1, 0, 3
    sphere(1, (0, -1000, 0), 1000, 1)
    nodes[2]["kind"], nodes[2]["material"], nodes[2]["child0"], nodes[2]["child1"] = abi.RT_NODE_LIST, -1, 3, 2
    moving(3, (-2.5, 0.6, 0.5), (-2.5, 1.4, 0.2), 0.0, 1.0, 0.6, 0)
    sphere(5, (0, 1, 0), 1.0, 3)
    moving(6, (2.5, 0.7, -0.5), (3.2, 0.9, -0.5), -1.0, 3.0, 0.7, 0)
    sphere(8, (1.2, 0.5, 2.0), 0.5, 2)
    moving(9, (-1.0, 0.4, 2.2), (-1.0, 0.4, 3.0), 0.0, 2.0, 0.4, 2)
    bvh(7, 8, 9)
    bvh(4, 6, 7)
    return abi.Scene(0, nodes, children, mat, tex, np.zeros(0, abi.PERLIN_DTYPE))


def cam_for(t0, t1, mk=lib.camera):
    return mk((6, 3, 7), (0, 0.8, 0), (0, 1, 0), 35.0, 4.0 / 3.0, 0.1, 9.0, t0, t1)


def check_against_oracle(ctx, scene, t0, t1, family):
    cam = cam_for(t0, t1)
    assert ctx.kernels_for(cam) == family
    p = abi.render_params(64, 48, 5, 50, seed=77)
    got, st = ctx.render(cam, p)
    want, ost = oracle_sums(scene, cam_for(t0, t1, ob.camera), p)
    assert st["rays"] == ost["rays"] and st["nan_samples"] == 0
    assert np.array_equal(got, want), np.abs(got - want).max()
    assert np.array_equal(ctx.first_hit(cam, 80, 60), ob.first_hit(scene, cam_for(t0, t1, ob.camera), 80, 60))


def test_nested_sphere_world_matches_oracle(ctx):
    scene = nested_world()
    ctx.upload(scene)
    check_against_oracle(ctx, scene, 0.0, 1.0, abi.RT_KERNELS_SPHERE_LIST)
    check_against_oracle(ctx, scene, 0.25, 0.75, abi.RT_KERNELS_SPHERE_LIST)
    check_against_oracle(ctx, scene, 0.5, 0.5, abi.RT_KERNELS_SPHERE_LIST)
    ids = ctx.first_hit(cam_for(0.0, 1.0), 80, 60)
    assert set(np.unique(ids)) == {-1, 0, 1, 2}  # indices within the root list, whatever the nesting below


def test_shutter_outside_a_sweep_takes_the_graph_walk(ctx):
    """Sphere 3 moves over [0, 1]: its screening bound does not cover a ray at time 1.4."""
    scene = nested_world()
    ctx.upload(scene)
    check_against_oracle(ctx, scene, 0.0, 1.5, abi.RT_KERNELS_SCENE_GRAPH)
    check_against_oracle(ctx, scene, -0.5, 0.5, abi.RT_KERNELS_SCENE_GRAPH)
    check_against_oracle(ctx, scene, 1.0, 0.0, abi.RT_KERNELS_SPHERE_LIST)  # reversed interval, same times


def test_bvh_box_that_does_not_enclose_its_leaves_takes_the_graph_walk(ctx):
    """The reference culls by the boxes it is given; a list of the leaves would not.  Such worlds keep the tree."""
    scene = nested_world()
    scene.nodes[7]["p"][4] -= 0.3  # top of the inner box now cuts through both spheres
    ctx.upload(scene)
    check_against_oracle(ctx, scene, 0.0, 1.0, abi.RT_KERNELS_SCENE_GRAPH)
    wide = nested_world()
    wide.nodes[4]["p"][:3] -= 1.0  # a box larger than needed is fine
    ctx.upload(wide)
    check_against_oracle(ctx, wide, 0.0, 1.0, abi.RT_KERNELS_SPHERE_LIST)


def test_other_primitives_or_materials_take_the_graph_walk(ctx):
    scene = nested_world()
    scene.materials["kind"][2] = abi.RT_MAT_DIFFUSE_LIGHT
    scene.materials["texture"][2] = 3
    ctx.upload(scene)
    check_against_oracle(ctx, scene, 0.0, 1.0, abi.RT_KERNELS_SCENE_GRAPH)


def test_sphere_count_limit(monkeypatch):
    monkeypatch.setenv("RT_SPHERE_SET_MAX", "6")  # read when a context is created
    c = lib.Context(0)
    try:
        scene = nested_world()
        c.upload(scene)
        check_against_oracle(c, scene, 0.0, 1.0, abi.RT_KERNELS_SPHERE_LIST)
    finally:
        c.close()
    monkeypatch.setenv("RT_SPHERE_SET_MAX", "5")
    c = lib.Context(0)
    try:
        c.upload(scene)
        check_against_oracle(c, scene, 0.0, 1.0, abi.RT_KERNELS_SCENE_GRAPH)
    finally:
        c.close()


@pytest.mark.parametrize("name", ["bouncing", "two_spheres"])
def test_reference_sphere_worlds_list_walk_equals_graph_walk(ctx, kat_b2, name):
    k = kat_b2[name]
    scene = load_b2(name)
    cam = camera_b2(k, lib.camera)
    p = abi.render_params(96, 54, 6, 50, seed=5)
    ctx.upload(scene)
    assert ctx.kernels_for(cam) == abi.RT_KERNELS_SPHERE_LIST
    a, sa = ctx.render(cam, p)
    ids_a = ctx.first_hit(cam, 160, 90)
    ctx.set_exact_only(True)
    e, se = ctx.render(cam, p)
    ctx.set_exact_only(False)
    ctx.set_force_graph(True)
    ctx.upload(scene)
    assert ctx.kernels_for(cam) == abi.RT_KERNELS_SCENE_GRAPH
    b, sb = ctx.render(cam, p)
    ids_b = ctx.first_hit(cam, 160, 90)
    ctx.set_force_graph(False)
    assert np.array_equal(a, b) and np.array_equal(a, e) and sa["rays"] == sb["rays"] == se["rays"]
    assert np.array_equal(ids_a, ids_b)


def test_bouncing_spheres_full_size_properties(ctx, kat_b2):
    """Config 3 at its full 1200x675: determinism, split additivity, the pixel-by-pixel pass (>= 100 spp) against the
    frame-by-frame numbering, three rows against the oracle."""
    k = kat_b2["bouncing"]
    scene = load_b2("bouncing")
    cam = camera_b2(k, lib.camera)
    ctx.upload(scene)
    W, H = 1200, 675
    a, st = ctx.render(cam, abi.render_params(W, H, 4, 50, seed=1))
    assert st["samples"] == W * H * 4 and st["nan_samples"] == 0 and st["overflow_rays"] == 0
    b, _ = ctx.render(cam, abi.render_params(W, H, 4, 50, seed=1))
    assert np.array_equal(a, b)
    lo, _ = ctx.render(cam, abi.render_params(W, H, 1, 50, seed=1))
    hi, _ = ctx.render(cam, abi.render_params(W, H, 3, 50, seed=1, sample_begin=1))
    assert np.array_equal(np.rint(lo * FRAC).astype(np.int64) + np.rint(hi * FRAC).astype(np.int64),
                          np.rint(a * FRAC).astype(np.int64))
    p = abi.render_params(W, H, 4, 50, seed=1)
    for row in (3, 337, 670):
        assert np.array_equal(a[row], ob.render_rows(scene, cam, p, row, row + 1, has_time=True)[0] / FRAC)
    # 104 spp in one call (pixel pass, accumulation cache) == 100 + 4 in two calls (4 spp: frame-by-frame numbering)
    small = abi.render_params(240, 135, 104, 50, seed=9)
    whole, _ = ctx.render(cam, small)
    p1, _ = ctx.render(cam, abi.render_params(240, 135, 100, 50, seed=9))
    p2, _ = ctx.render(cam, abi.render_params(240, 135, 4, 50, seed=9, sample_begin=100))
    assert np.array_equal(np.rint(p1 * FRAC).astype(np.int64) + np.rint(p2 * FRAC).astype(np.int64),
                          np.rint(whole * FRAC).astype(np.int64))


# ---------------------------------------------------------------- sphere subtrees inside a general graph
def instanced_cluster_world():
    """root list [ground, light rect, translate(rotate_y(bvh of 5 static spheres)), bvh of 2 spheres, glass sphere]:
    not a sphere world (the rectangle), but both BVH subtrees hold only static spheres."""
    import math
    tex = np.zeros(3, abi.TEXTURE_DTYPE)
    tex["kind"] = abi.RT_TEX_SOLID
    tex["p"][0][:3] = (0.5, 0.6, 0.4)
    tex["p"][1][:3] = (7.0, 7.0, 7.0)
    tex["p"][2][:3] = (0.73, 0.73, 0.73)
    mat = np.zeros(5, abi.MATERIAL_DTYPE)
    mat["kind"] = [abi.RT_MAT_LAMBERTIAN, abi.RT_MAT_DIFFUSE_LIGHT, abi.RT_MAT_LAMBERTIAN, abi.RT_MAT_METAL, abi.RT_MAT_DIELECTRIC]
    mat["texture"] = [0, 1, 2, -1, -1]
    mat["albedo"][3] = (0.8, 0.8, 0.9)
    mat["param"] = [0, 0, 0, 0.1, 1.5]
    nodes = np.zeros(20, abi.NODE_DTYPE)
    nodes["material"] = -1

    def sphere(i, c, r, m):
        nodes[i]["kind"], nodes[i]["material"] = abi.RT_NODE_SPHERE, m
        nodes[i]["p"][:4] = (*c, r)

    def box(i):
        n = nodes[i]
        if n["kind"] == abi.RT_NODE_SPHERE:
            c, r = n["p"][:3], n["p"][3]
            return c - r, c + r
        lo_l, hi_l = box(n["child0"])
        lo_r, hi_r = box(n["child1"])
        return np.minimum(lo_l, lo_r), np.maximum(hi_l, hi_r)

    def bvh(i, left, right):
        nodes[i]["kind"] = abi.RT_NODE_BVH
        nodes[i]["child0"], nodes[i]["child1"] = left, right
        lo, hi = box(i)
        nodes[i]["p"][:6] = (*lo, *hi)

    children = np.array([1, 2, 3, 14, 17], dtype=np.int32)
    nodes[0]["kind"], nodes[0]["child0"], nodes[0]["child1"] = abi.RT_NODE_LIST, 0, 5
    sphere(1, (0, -1000, 0), 1000, 0)
    nodes[2]["kind"], nodes[2]["material"] = abi.RT_NODE_XZ_RECT, 1
    nodes[2]["p"][:5] = (-2, 2, -2, 2, 6)
    # the cluster, in its own frame around the origin
    sphere(6, (-0.8, 0.5, 0.0), 0.5, 2)
    sphere(7, (0.4, 0.4, 0.6), 0.4, 3)
    sphere(8, (0.3, 0.45, -0.7), 0.45, 2)
    sphere(9, (0.0, 1.3, 0.0), 0.4, 4)
    sphere(10, (1.3, 0.3, 0.1), 0.3, 2)
    bvh(11, 6, 7)
    bvh(12, 8, 9)
    bvh(13, 12, 10)
    bvh(5, 11, 13)
    theta = math.radians(25.0)
    nodes[4]["kind"], nodes[4]["child0"] = abi.RT_NODE_ROTATE_Y, 5
    nodes[4]["p"][:2] = (math.sin(theta), math.cos(theta))
    nodes[3]["kind"], nodes[3]["child0"] = abi.RT_NODE_TRANSLATE, 4
    nodes[3]["p"][:3] = (-1.5, 0.0, 1.0)
    sphere(15, (2.2, 0.5, -1.0), 0.5, 3)
    sphere(16, (2.6, 0.4, 0.4), 0.4, 2)
    bvh(14, 15, 16)
    sphere(17, (0.8, 0.6, 2.6), 0.6, 4)
    return abi.Scene(0, nodes[:18].copy(), children, mat, tex, np.zeros(0, abi.PERLIN_DTYPE),
                     sky_mode=abi.RT_SKY_BACKGROUND, background=(0.05, 0.05, 0.08))


def render_with_subtree_sets(monkeypatch, scene, cam, p, set_min, fh=(80, 60)):
    monkeypatch.setenv("RT_SUBTREE_SET_MIN", str(set_min))  # read when a context is created
    c = lib.Context(0)
    try:
        c.upload(scene)
        assert c.kernels_for(cam) == abi.RT_KERNELS_SCENE_GRAPH
        img, st = c.render(cam, p)
        ids = c.first_hit(cam, *fh)
    finally:
        c.close()
    return img, st, ids


def test_sphere_subtrees_inside_a_graph_match_oracle(monkeypatch):
    scene = instanced_cluster_world()
    cam, ocam = cam_for(0.0, 1.0), cam_for(0.0, 1.0, ob.camera)
    p = abi.render_params(64, 48, 6, 50, seed=31)
    want, ost = oracle_sums(scene, ocam, p)
    want_ids = ob.first_hit(scene, ocam, 80, 60)
    assert {2, 3, 4} <= set(np.unique(want_ids))  # both clusters are in view
    for set_min in (2, 5, 1000):  # both subtrees as sets; the five-sphere one only; none
        got, st, ids = render_with_subtree_sets(monkeypatch, scene, cam, p, set_min)
        assert st["rays"] == ost["rays"], set_min
        assert np.array_equal(got, want), (set_min, np.abs(got - want).max())
        assert np.array_equal(ids, want_ids), set_min


def test_subtree_with_a_short_box_keeps_its_tree(monkeypatch):
    scene = instanced_cluster_world()
    scene.nodes[13]["p"][4] -= 0.5  # cuts the top sphere: the reference culls there, a list of the leaves would not
    cam, ocam = cam_for(0.0, 1.0), cam_for(0.0, 1.0, ob.camera)
    p = abi.render_params(64, 48, 6, 50, seed=31)
    want, ost = oracle_sums(scene, ocam, p)
    got, st, ids = render_with_subtree_sets(monkeypatch, scene, cam, p, 2)
    assert st["rays"] == ost["rays"] and np.array_equal(got, want)
    assert np.array_equal(ids, ob.first_hit(scene, ocam, 80, 60))


def test_final_scene_sphere_cube_as_a_set(monkeypatch, kat_b2):
    """Config 4: the cube of 1000 spheres under translate(rotate_y(bvh)) is one sphere set; same bits without."""
    k = kat_b2["final"]
    scene = load_b2("final")
    cam = camera_b2(k, lib.camera)
    p = abi.render_params(200, 200, 8, 50, seed=4)
    a, sa, ids_a = render_with_subtree_sets(monkeypatch, scene, cam, p, 64, fh=(200, 200))
    b, sb, ids_b = render_with_subtree_sets(monkeypatch, scene, cam, p, 100000, fh=(200, 200))
    assert sa["rays"] == sb["rays"] and np.array_equal(a, b) and np.array_equal(ids_a, ids_b)
    rows = ob.render_rows(scene, camera_b2(k), p, 60, 62, has_time=True) / FRAC
    assert np.array_equal(a[60:62], rows)
    # validation switch: the same sets with their float screens bypassed (every leaf tested in double)
    c = lib.Context(0)
    try:
        c.upload(scene)
        c.set_exact_only(True)
        e, se = c.render(cam, p)
        ids_e = c.first_hit(cam, 200, 200)
    finally:
        c.close()
    assert se["rays"] == sa["rays"] and np.array_equal(e, a) and np.array_equal(ids_e, ids_a)


def box_field_world(scale=1.0):
    """root list [ground sphere, bvh over 5 boxes of unequal heights, metal sphere]; lengths multiplied by `scale`."""
    tex = np.zeros(2, abi.TEXTURE_DTYPE)
    tex["kind"] = abi.RT_TEX_SOLID
    tex["p"][0][:3] = (0.48, 0.83, 0.53)
    tex["p"][1][:3] = (0.6, 0.5, 0.4)
    mat = np.zeros(3, abi.MATERIAL_DTYPE)
    mat["kind"] = [abi.RT_MAT_LAMBERTIAN, abi.RT_MAT_LAMBERTIAN, abi.RT_MAT_METAL]
    mat["texture"] = [0, 1, -1]
    mat["albedo"][2] = (0.8, 0.8, 0.9)
    mat["param"] = [0, 0, 0.05]
    corners = [((-2.0, 0.0, -1.0), (-1.0, 0.9, 0.0)), ((-1.0, 0.0, -1.0), (0.0, 0.4, 0.0)), ((0.0, 0.0, -1.0), (1.0, 1.3, 0.0)),
               ((-1.0, 0.0, 0.0), (0.0, 0.7, 1.0)), ((0.0, 0.0, 0.0), (1.0, 0.2, 1.0))]
    n_boxes = len(corners)
    first_box = 3
    first_side = first_box + n_boxes
    first_bvh = first_side + 6 * n_boxes
    nodes = np.zeros(first_bvh + n_boxes - 1, abi.NODE_DTYPE)
    nodes["material"] = -1
    children = [1, 2, first_bvh]
    nodes[0]["kind"], nodes[0]["child0"], nodes[0]["child1"] = abi.RT_NODE_LIST, 0, 3
    nodes[1]["kind"], nodes[1]["material"] = abi.RT_NODE_SPHERE, 1
    nodes[1]["p"][:4] = (0, -1000 * scale, 0, 1000 * scale)
    nodes[2]["kind"], nodes[2]["material"] = abi.RT_NODE_SPHERE, 2
    nodes[2]["p"][:4] = (1.8 * scale, 0.6 * scale, 1.2 * scale, 0.6 * scale)
    for b, (lo, hi) in enumerate(corners):
        (x0, y0, z0), (x1, y1, z1) = np.array(lo) * scale, np.array(hi) * scale
        i = first_box + b
        nodes[i]["kind"], nodes[i]["child0"], nodes[i]["child1"] = abi.RT_NODE_BOX, len(children), 6
        nodes[i]["p"][:6] = (x0, y0, z0, x1, y1, z1)
        sides = [(abi.RT_NODE_XY_RECT, (x0, x1, y0, y1, z1)), (abi.RT_NODE_XY_RECT, (x0, x1, y0, y1, z0)),
                 (abi.RT_NODE_XZ_RECT, (x0, x1, z0, z1, y1)), (abi.RT_NODE_XZ_RECT, (x0, x1, z0, z1, y0)),
                 (abi.RT_NODE_YZ_RECT, (y0, y1, z0, z1, x1)), (abi.RT_NODE_YZ_RECT, (y0, y1, z0, z1, x0))]
        for s, (kind, p) in enumerate(sides):
            j = first_side + 6 * b + s
            nodes[j]["kind"], nodes[j]["material"] = kind, 0
            nodes[j]["p"][:5] = p
            children.append(j)

    def box(i):
        n = nodes[i]
        if n["kind"] == abi.RT_NODE_BOX:
            return n["p"][:3].copy(), n["p"][3:6].copy()
        lo_l, hi_l = box(n["child0"])
        lo_r, hi_r = box(n["child1"])
        return np.minimum(lo_l, lo_r), np.maximum(hi_l, hi_r)

    # a chain-shaped tree: bvh_k = (box_k, bvh_{k+1}), the last one holds two boxes
    for k in range(n_boxes - 2, -1, -1):
        i = first_bvh + k
        nodes[i]["kind"] = abi.RT_NODE_BVH
        nodes[i]["child0"] = first_box + k
        nodes[i]["child1"] = first_bvh + k + 1 if k < n_boxes - 2 else first_box + k + 1
        lo, hi = box(i)
        nodes[i]["p"][:6] = (*lo, *hi)
    return abi.Scene(0, nodes, np.array(children, dtype=np.int32), mat, tex, np.zeros(0, abi.PERLIN_DTYPE))


@pytest.mark.parametrize("scale,lookfrom,vfov", [(1.0, (4.0, 3.0, 5.0), 30.0), (1.0e5, (4.0e5, 3.0e5, 5.0e5), 30.0),
                                                 (2.0e5, (4.0e7, 3.0e7, 5.0e7), 1.0)])
def test_box_subtrees_inside_a_graph_match_oracle(monkeypatch, scale, lookfrom, vfov):
    """The slab screen of the boxes against the oracle; in the last case the camera lies outside the range the
    float bound covers (|origin| > 8e6), so primary rays take the all-boxes-in-double path."""
    scene = box_field_world(scale)
    mk = lambda f: f(lookfrom, (0, 0.4 * scale, 0), (0, 1, 0), vfov, 4.0 / 3.0, 0.0, 1.0, 0.0, 1.0)
    cam, ocam = mk(lib.camera), mk(ob.camera)
    p = abi.render_params(64, 48, 6, 50, seed=8)
    want, ost = oracle_sums(scene, ocam, p)
    want_ids = ob.first_hit(scene, ocam, 80, 60)
    assert 2 in np.unique(want_ids)
    for set_min in (2, 1000):
        got, st, ids = render_with_subtree_sets(monkeypatch, scene, cam, p, set_min)
        assert st["rays"] == ost["rays"], set_min
        assert np.array_equal(got, want), (set_min, np.abs(got - want).max())
        assert np.array_equal(ids, want_ids), set_min


def test_negative_radius_under_a_bvh_keeps_the_tree(ctx):
    """sphere::bounding_box of a negative radius is inside out, so the reference's tree culls hits a list would find:
    such worlds are walked as trees, and match the oracle's tree walk."""
    scene = nested_world()
    scene.nodes[8]["p"][3] = -0.5  # the metal sphere under bvh 7 (the node boxes still enclose it: |r| is unchanged)
    ctx.upload(scene)
    check_against_oracle(ctx, scene, 0.0, 1.0, abi.RT_KERNELS_SCENE_GRAPH)
    flat = nested_world()
    flat.nodes[5]["p"][3] = -0.9  # the glass sphere directly in a list: no box involved, but still not a sphere world
    ctx.upload(flat)
    check_against_oracle(ctx, flat, 0.0, 1.0, abi.RT_KERNELS_SCENE_GRAPH)


@pytest.mark.parametrize("world", ["nested", "cluster", "boxes"])
def test_rays_with_nan_follow_the_reference_through_every_member(monkeypatch, world):
    """A ray with a NaN in it makes every comparison of the reference false, so every object it tests counts as hit and
    the last one tested wins.  The float screens mean nothing for such a ray (and a NaN's sign could name padding
    entries): those lanes test every member in double, in order.  Camera origin = NaN makes every ray one."""
    scene = {"nested": nested_world, "cluster": instanced_cluster_world, "boxes": lambda: box_field_world(1.0)}[world]()
    cam, ocam = cam_for(0.0, 1.0), cam_for(0.0, 1.0, ob.camera)
    cam.origin[0] = ocam.origin[0] = float("nan")
    p = abi.render_params(32, 24, 2, 50, seed=5)
    _, fixed, ost = ob.render(scene, ocam, p, rng=ob.RNG_PHILOX, assoc=ob.ASSOC_FORWARD, has_time=True, threads=8)
    want_ids = ob.first_hit(scene, ocam, 16, 12)
    monkeypatch.setenv("RT_SUBTREE_SET_MIN", "2")
    c = lib.Context(0)
    try:
        c.upload(scene)
        got, st = c.render(cam, p)
        ids = c.first_hit(cam, 16, 12)
    finally:
        c.close()
    assert st["rays"] == ost["rays"] and st["nan_samples"] == ost["nan_samples"]
    assert np.array_equal(got, fixed.astype(np.float64) / FRAC)
    assert np.array_equal(ids, want_ids)
