"""rt_render_noalloc, the entry point for callers that cannot allocate, against rt_render: same sums bit for bit, on
two calls in a row of which the second is smaller (the scratch array keeps the first call's contents), and the size
limit.  Calls go through the C ABI (ctypes); torch only provides the caller's device buffer."""
import numpy as np
import pytest

from raytracing_amd import abi, lib
from tests.conftest import CAM_C2

pytestmark = pytest.mark.gpu


def test_noalloc_render_equals_rt_render_on_two_calls_in_a_row(scene_c2):
    import torch
    cam = lib.camera(*CAM_C2)
    calls = [(96, 64, 7, 3), (40, 30, 3, 4)]   # (W, H, spp, seed): the second frame is smaller than the first
    with lib.Context(0) as ctx:
        ctx.upload(scene_c2)
        out = torch.full((96 * 64 * 3,), float("nan"), dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        got = []
        for W, H, spp, seed in calls:
            ctx.render_noalloc(cam, abi.render_params(W, H, spp, 50, seed=seed), out.data_ptr())
            ctx.synchronize()
            got.append(out[: W * H * 3].cpu().numpy().reshape(H, W, 3).copy())
        for (W, H, spp, seed), g in zip(calls, got):
            want, _ = ctx.render(cam, abi.render_params(W, H, spp, 50, seed=seed))
            assert np.array_equal(g, want)
        # the property the entry point exists for: after a first call, a call leaves the device's free memory as it
        # found it (no hipMalloc, no event, no staging buffer), and can be repeated any number of times
        torch.cuda.synchronize()
        free_before, _ = torch.cuda.mem_get_info(0)
        for W, H, spp, seed in calls * 3:
            ctx.render_noalloc(cam, abi.render_params(W, H, spp, 50, seed=seed), out.data_ptr())
        ctx.synchronize()
        free_after, _ = torch.cuda.mem_get_info(0)
        assert free_after == free_before, (free_before, free_after)
        # nothing to render: the output is overwritten with zeros
        ctx.render_noalloc(cam, abi.render_params(40, 30, 0, 50), out.data_ptr())
        ctx.synchronize()
        assert not out[: 40 * 30 * 3].any().item()
        # a frame beyond the scratch array is refused, and says why
        with pytest.raises(lib.RtError) as e:
            ctx.render_noalloc(cam, abi.render_params(2048, 1024, 1, 50), out.data_ptr())
        assert e.value.code == abi.RT_ERR_LIMIT and "RT_NOALLOC_MAX_PIXELS" in str(e.value)
        with pytest.raises(lib.RtError):
            ctx.render_noalloc(cam, abi.render_params(40, 30, 1, 50), 0)


def test_noalloc_on_a_scene_graph_world_and_from_a_second_context():
    """The scratch array belongs to the GPU, not to a context: a second context of the same device uses it after the
    first one (one after the other), on a world rendered by the scene-graph kernels."""
    import json

    import torch

    from tests.conftest import gold
    with open(gold("kat_b2.json")) as fp:
        k = json.load(fp)["cornell_box"]
    world = abi.load_scene(gold("scene_b2_cornell_box.rtsc"))
    cam = lib.camera(k["lookfrom"], k["lookat"], (0, 1, 0), k["vfov"], k["aspect"], k["aperture"], 10.0, 0.0, 1.0)
    p = abi.render_params(48, 48, 4, 50, seed=9)
    out = torch.zeros(48 * 48 * 3, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    with lib.Context(0) as a, lib.Context(0) as b:
        a.upload(world)
        b.upload(world)
        want, _ = a.render(cam, p)
        for ctx in (a, b):
            out.zero_()
            torch.cuda.synchronize()
            ctx.render_noalloc(cam, p, out.data_ptr())
            ctx.synchronize()
            assert np.array_equal(out.cpu().numpy().reshape(48, 48, 3), want)
