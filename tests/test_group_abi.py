"""Host side of the several-GPU entry points, without a GPU: the sample split through the C ABI, and loud failures."""
import ctypes as C

import pytest

from raytracing_amd import abi, frame, lib


def test_split_samples_partitions_exactly():
    for spp in (0, 1, 7, 62, 500, 10000):
        for n in (1, 2, 3, 8, 16):
            got = [lib.split_samples(spp, r, n) for r in range(n)]
            assert [s for b, c in got for s in range(b, b + c)] == list(range(spp))
            counts = [c for _, c in got]
            assert max(counts) - min(counts) <= 1 and counts == sorted(counts, reverse=True)
            assert got == [frame.sample_range(spp, r, n) for r in range(n)]
    assert [lib.split_samples(500, r, 8)[1] for r in range(8)] == [63, 63, 63, 63, 62, 62, 62, 62]
    # bad arguments: nothing to render (the C function has no error channel)
    assert lib.split_samples(10, 2, 2) == (0, 0) and lib.split_samples(10, -1, 2) == (0, 0)
    assert lib.split_samples(-1, 0, 2) == (0, 0) and lib.split_samples(10, 0, 0) == (0, 0)


def test_group_entry_points_fail_loudly_without_a_gpu():
    L = lib.load()
    if L.rt_get_device_count() > 0:
        pytest.skip("a GPU is visible: covered by test_gpu_group.py")
    for make in (lambda: lib.Group(devices=[0]), lambda: lib.Group(devices=[0, 1], frame_slots=3),
                 lambda: lib.Group(device=0, rank=0, world=1, comm_id=bytes(abi.RT_COMM_ID_BYTES))):
        with pytest.raises(lib.RtError) as e:
            make()
        assert e.value.code == abi.RT_ERR_NO_DEVICE


def test_group_null_handles_are_errors_not_crashes():
    L = lib.load()
    st = abi.rt_stats()
    assert L.rt_group_size(None) == 0 and L.rt_group_local_count(None) == 0 and L.rt_group_frame_slots(None) == 0
    assert L.rt_group_context(None, 0, 0) is None
    assert L.rt_group_device_sums(None, 0, 0) is None
    assert L.rt_group_wait(None, 0) == abi.RT_ERR_INVALID_ARGUMENT
    assert L.rt_group_take_stats(None, C.byref(st)) == abi.RT_ERR_INVALID_ARGUMENT
    assert L.rt_group_scene_upload(None, None) == abi.RT_ERR_INVALID_ARGUMENT
    assert L.rt_group_render(None, None, None, None, None) == abi.RT_ERR_INVALID_ARGUMENT
    assert L.rt_group_create(None, 0, 1, None) == abi.RT_ERR_INVALID_ARGUMENT
    L.rt_group_destroy(None)
    assert L.rt_error_string(abi.RT_ERR_COMM) == b"RCCL error"
