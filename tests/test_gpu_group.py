"""The several-GPU entry points (rt_group_*) on the one GPU a test box has: groups of ONE member, whose frames still
pass through a real RCCL communicator (ncclCommInitAll / ncclCommInitRank with one rank, ncclAllReduce of the int64
sums inside ncclGroupStart/End on the kernel's stream).  The bar is the library's own single-GPU path, bit for bit,
which test_gpu_parity.py holds against the oracle; the split of samples over members is host arithmetic, tested
without a GPU in test_group_abi.py and, through rt_render_device, by test_sample_ranges_add_up_exactly.
"""
import numpy as np
import pytest

from raytracing_amd import abi, lib
from tests.conftest import CAM_C2, gold

pytestmark = pytest.mark.gpu

W, H, SPP = 96, 64, 9


@pytest.fixture(scope="module")
def single(scene_c2):
    """The unsplit frames of the single-GPU entry point, by seed."""
    cam = lib.camera(*CAM_C2)
    out = {}
    with lib.Context(0) as ctx:
        ctx.upload(scene_c2)
        for seed in (1, 2, 3, 4, 5):
            out[seed] = ctx.render(cam, abi.render_params(W, H, SPP, 50, seed=seed))
    return out


def test_group_of_one_device_equals_the_single_gpu_frame(scene_c2, single):
    cam = lib.camera(*CAM_C2)
    with lib.Group(devices=[0]) as g:
        assert (g.size, g.local_count, g.frame_slots) == (1, 1, 1)
        g.upload(scene_c2)
        got, st = g.render(cam, abi.render_params(W, H, SPP, 50, seed=1))
        want, wst = single[1]
        assert np.array_equal(got, want)
        assert (st["samples"], st["rays"], st["nan_samples"]) == (wst["samples"], wst["rays"], 0)
        assert st["kernel_launches"] == 1 and st["kernel_ms"] > 0
        # a second frame on the same slot starts from zero again
        got2, _ = g.render(cam, abi.render_params(W, H, SPP, 50, seed=2))
        assert np.array_equal(got2, single[2][0])
        # a frame with an offset sample range: the group splits relative to the frame's own begin
        with lib.Context(0) as ctx:
            ctx.upload(scene_c2)
            p = abi.render_params(W, H, 4, 50, seed=1, sample_begin=5)
            assert np.array_equal(g.render(cam, p)[0], ctx.render(cam, p)[0])


def test_group_created_by_rank_with_a_comm_id(scene_c2, single):
    """The one-process-per-GPU construction: unique id from rank 0, ncclCommInitRank; here with one rank."""
    cam = lib.camera(*CAM_C2)
    cid = lib.comm_id_create()
    assert len(cid) == abi.RT_COMM_ID_BYTES and any(cid)
    with lib.Group(device=0, rank=0, world=1, comm_id=cid) as g:
        g.upload(scene_c2)
        got, _ = g.render(cam, abi.render_params(W, H, SPP, 50, seed=3))
        assert np.array_equal(got, single[3][0])


def test_frames_in_flight_on_several_slots(scene_c2, single):
    """Three frames enqueued back to back on three slots (own streams, contexts and buffers, one communicator), then
    waited for in order: each equals its single-GPU frame; a slot takes a new frame after its wait."""
    cam = lib.camera(*CAM_C2)
    with lib.Group(devices=[0], frame_slots=3) as g:
        g.upload(scene_c2)
        for k in range(3):
            g.render_async(cam, abi.render_params(W, H, SPP, 50, seed=1 + k), slot=k)
        for k in range(3):
            g.wait(k)
            assert np.array_equal(g.read(W, H, slot=k), single[1 + k][0])
        for k in (3, 4):
            g.render_async(cam, abi.render_params(W, H, SPP, 50, seed=1 + k), slot=k % 3)
        for k in (3, 4):
            g.wait(k % 3)
            assert np.array_equal(g.read(W, H, slot=k % 3), single[1 + k][0])
        assert g.device_sums(0, 0) != 0
        st = g.take_stats()
        assert st["samples"] == 5 * W * H * SPP and st["kernel_launches"] == 5


def test_group_renders_a_scene_graph_world():
    """A Next-Week world (boxes, instances, a medium) through the group: the same scene-graph kernels, same sums."""
    import json
    with open(gold("kat_b2.json")) as fp:
        k = json.load(fp)["cornell_smoke"]
    world = abi.load_scene(gold("scene_b2_cornell_smoke.rtsc"))
    cam = lib.camera(k["lookfrom"], k["lookat"], (0, 1, 0), k["vfov"], k["aspect"], k["aperture"], 10.0, 0.0, 1.0)
    p = abi.render_params(48, 48, 5, 50, seed=7)
    with lib.Context(0) as ctx:
        ctx.upload(world)
        want, wst = ctx.render(cam, p)
    with lib.Group(devices=[0]) as g:
        g.upload(world)
        got, st = g.render(cam, p)
    assert np.array_equal(got, want) and st["rays"] == wst["rays"]


def test_group_argument_errors(scene_c2):
    with pytest.raises(lib.RtError) as e:
        lib.Group(devices=[0, 0])
    assert e.value.code == abi.RT_ERR_INVALID_ARGUMENT
    with pytest.raises(lib.RtError) as e:
        lib.Group(devices=[lib.load().rt_get_device_count()])
    assert e.value.code == abi.RT_ERR_NO_DEVICE
    with pytest.raises(lib.RtError) as e:
        lib.Group(devices=[0], frame_slots=9)
    assert e.value.code == abi.RT_ERR_INVALID_ARGUMENT
    cam = lib.camera(*CAM_C2)
    with lib.Group(devices=[0]) as g:
        with pytest.raises(lib.RtError) as e:   # no scene yet
            g.render(cam, abi.render_params(W, H, 1))
        assert e.value.code == abi.RT_ERR_NO_SCENE
        g.upload(scene_c2)
        with pytest.raises(lib.RtError) as e:
            g.render_async(cam, abi.render_params(W, H, 1), slot=1)
        assert e.value.code == abi.RT_ERR_INVALID_ARGUMENT
        with pytest.raises(lib.RtError):
            g.read(W, H, slot=0)               # the slot holds no frame yet


def test_read_sums_equals_rt_render(scene_c2, single):
    """rt_render_device into a caller's buffer + rt_read_sums = rt_render (the device-side conversion of the fixed-point
    sums gives the doubles a host loop would)."""
    import torch
    cam = lib.camera(*CAM_C2)
    with lib.Context(0) as ctx:
        ctx.upload(scene_c2)
        buf = torch.zeros((H, W, 3), dtype=torch.int64, device="cuda:0")
        torch.cuda.synchronize()
        ctx.render_device(cam, abi.render_params(W, H, SPP, 50, seed=1), buf.data_ptr())
        got = ctx.read_sums(buf.data_ptr(), W, H)
        assert np.array_equal(got, single[1][0])
        fixed = buf.cpu().numpy()
        assert np.array_equal(got, fixed.astype(np.float64) / 2.0 ** abi.RT_ACCUM_FRAC_BITS)
        # values beyond 2^53 round to nearest even on both sides
        big = torch.tensor([2 ** 53 + 1, 2 ** 53 + 3, -(2 ** 53 + 1), 2 ** 62 + 2 ** 9, 0, -1], dtype=torch.int64, device="cuda:0")
        torch.cuda.synchronize()
        out = ctx.read_sums(big.data_ptr(), 2, 1)
        assert np.array_equal(out.ravel(), big.cpu().numpy().astype(np.float64) / 2.0 ** abi.RT_ACCUM_FRAC_BITS)


def test_a_member_that_fails_still_joins_the_collective_and_every_wait_reports_it(scene_c2, single):
    """A launch failure on one member after the checks have passed (requested through the validation switch): the call
    returns the error, but only after the all-reduce has been joined -- no peer would be left waiting in it --, the
    frame's failure word travels with the sums, and the group renders the next frame as if nothing had happened."""
    cam = lib.camera(*CAM_C2)
    with lib.Group(devices=[0], frame_slots=2) as g:
        g.upload(scene_c2)
        g.reserve(W, H)
        g.set_fail_next(0)
        with pytest.raises(lib.RtError) as e:
            g.render_async(cam, abi.render_params(W, H, SPP, 50, seed=1), slot=0)
        assert e.value.code == abi.RT_ERR_HIP and "collective was joined" in str(e.value)
        with pytest.raises(lib.RtError) as e:       # nothing hangs, and the wait reports what every rank's wait reports
            g.wait(0)
        assert e.value.code == abi.RT_ERR_COMM and "1 member(s)" in str(e.value)
        with pytest.raises(lib.RtError):
            g.read(W, H, slot=0)                    # the slot holds no frame
        # the failure word as a peer sees it: the word behind the slot's 3 W H sums, summed over the members
        n = 3 * W * H
        src, ctx = g.device_sums(0, 0), g.context(0, 0)
        assert not ctx.read_sums(src, W, H).any()   # the failed member's share (here: everything) is missing
        tail = ctx.read_sums(src + 8 * (n - 2), 1, 1).ravel() * 2.0 ** abi.RT_ACCUM_FRAC_BITS
        assert tail.tolist() == [0.0, 0.0, 1.0]     # one member failed
        # the communicator is still good: the next frames, on either slot, are whole
        for k, slot in ((1, 0), (2, 1)):
            g.render_async(cam, abi.render_params(W, H, SPP, 50, seed=k), slot=slot)
        for k, slot in ((1, 0), (2, 1)):
            g.wait(slot)
            assert np.array_equal(g.read(W, H, slot=slot), single[k][0])


def test_refusals_come_before_anything_is_enqueued(scene_c2, single):
    """What can be refused is refused for every member before a kernel or a collective is enqueued: the slot keeps the
    frame it held, and rt_render_check gives the same verdict without rendering."""
    cam = lib.camera(*CAM_C2)
    with lib.Group(devices=[0]) as g:
        g.upload(scene_c2)
        got, _ = g.render(cam, abi.render_params(W, H, SPP, 50, seed=4))
        assert np.array_equal(got, single[4][0])
        bad = abi.render_params(W, H, SPP, 50, seed=4, flags=abi.RT_RENDER_IMPORTANCE)  # a sphere list has no light shapes
        ctx = g.context(0, 0)
        assert ctx.render_check(cam, bad) == abi.RT_ERR_UNSUPPORTED_SCENE
        assert ctx.render_check(cam, abi.render_params(W, H, SPP, 50, seed=4)) == abi.RT_OK
        with pytest.raises(lib.RtError) as e:
            g.render_async(cam, bad, slot=0)
        assert e.value.code == abi.RT_ERR_UNSUPPORTED_SCENE
        g.wait(0)
        assert np.array_equal(g.read(W, H, slot=0), single[4][0])  # untouched


needs_two_gpus = pytest.mark.skipif(lib.load().rt_get_device_count() < 2, reason="needs two GPUs")


@needs_two_gpus
def test_two_devices_in_one_process_equal_the_single_gpu_frame(scene_c2, single):
    """The real thing: two members on two GPUs (ncclCommInitAll), samples split 5 + 4, one all-reduce per frame."""
    cam = lib.camera(*CAM_C2)
    with lib.Group(devices=[0, 1], frame_slots=3) as g:
        assert (g.size, g.local_count) == (2, 2)
        g.upload(scene_c2)
        got, st = g.render(cam, abi.render_params(W, H, SPP, 50, seed=1))
        assert np.array_equal(got, single[1][0])
        assert st["samples"] == W * H * SPP and st["rays"] == single[1][1]["rays"] and st["kernel_launches"] == 2
        for k in range(3):
            g.render_async(cam, abi.render_params(W, H, SPP, 50, seed=1 + k), slot=k)
        for k in range(3):
            g.wait(k)
            assert np.array_equal(g.read(W, H, slot=k), single[1 + k][0])
        # a failure on the second member reaches the first member's copy of the frame through the all-reduce
        g.set_fail_next(1)
        with pytest.raises(lib.RtError):
            g.render_async(cam, abi.render_params(W, H, SPP, 50, seed=1), slot=0)
        with pytest.raises(lib.RtError) as e:
            g.wait(0)
        assert e.value.code == abi.RT_ERR_COMM
        got, _ = g.render(cam, abi.render_params(W, H, SPP, 50, seed=5))
        assert np.array_equal(got, single[5][0])


def _rank_main(rank, world, cid, scene_path, out_q):
    import numpy as np
    from raytracing_amd import abi, lib
    from tests.conftest import CAM_C2
    cam = lib.camera(*CAM_C2)
    with lib.Group(device=rank, rank=rank, world=world, comm_id=cid, frame_slots=2) as g:
        g.upload(abi.load_scene(scene_path))
        for k in range(2):
            g.render_async(cam, abi.render_params(W, H, SPP, 50, seed=1 + k), slot=k)
        frames = []
        for k in range(2):
            g.wait(k)
            frames.append(g.read(W, H, slot=k))
        out_q.put((rank, frames))


@needs_two_gpus
def test_two_ranks_in_two_processes_equal_the_single_gpu_frame(single):
    """One process per GPU: the id from rank 0, ncclCommInitRank on both, two frames in flight."""
    import multiprocessing as mp
    mpc = mp.get_context("spawn")
    cid = lib.comm_id_create()
    q = mpc.Queue()
    procs = [mpc.Process(target=_rank_main, args=(r, 2, cid, gold("scene_c2.rtsc"), q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=100) for _ in procs)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank in (0, 1):
        for k in range(2):
            assert np.array_equal(results[rank][k], single[1 + k][0])
