#!/usr/bin/env python3
"""Headline benchmark: Msamples/s on the One-Weekend final scene (BASELINE.json configs[1]).

A step is one full frame of the reference's render loop over the dumped 487-sphere world:
1200x800 pixels, 500 samples per pixel, depth 50, camera of RayTracing.cpp:104-110.  With N
GPUs the 500 samples of every pixel are split 63/62 over the GPUs and the fixed-point
accumulation buffers are summed by one RCCL all-reduce per frame inside the library
(rt_group_*), so the N-GPU frame equals the 1-GPU frame bit for bit ("strong" scaling: total
work is fixed).  Scene and accumulation buffers are resident in HBM before the timed region;
tone mapping and PPM output are outside it.

Three ways to start it:
  python3 bench.py                         one GPU, one rt_context
  python -m torch.distributed.run ... bench.py --gpus N    one process per GPU: rank 0 makes the RCCL id, hands it
                                           round over torch.distributed (gloo, control plane only: id, barriers, the
                                           MAX of the ranks' times), every rank joins with rt_group_create_rank
  python3 bench.py --gpus N                one process driving N GPUs (rt_group_create = ncclCommInitAll)

Prints ONE JSON line on rank 0.  With several GPUs `value` is the throughput of frames kept in flight
(--frames-in-flight, default 3: the tail of one frame and its all-reduce overlap the next frame); the time of a
frame rendered alone, first launch to summed buffer complete, is reported beside it as `single_frame`.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_VALU_TFLOPS = 157.3      # MI355X vector FP32 peak: 256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz
PUBLISHED_MSAMPLES_S = 480.0e6 / 36000.0 / 1e6  # "ran ten hours" for this exact frame (reference README.md:2510)


def algorithmic_flop(n_objects, rays, samples):
    """Flops of the REFERENCE algorithm for this scene: a flat list tested in full for every ray.
    23 flop per sphere test up to the discriminant sign (3 sub, 5 + 5 for the two dot products,
    7 for c and the discriminant, 3 for comparisons), 60 per ray for the accepted hit, the scatter
    and the sky, 40 per sample for jitter and camera ray."""
    return rays * (n_objects * 23.0 + 60.0) + samples * 40.0


def kernel_source_sha256():
    """Hash of the kernel sources the library is built from: a committed counter summary carries the hash of the tree
    it was measured on, so figures taken from it can be marked stale once the kernels have changed."""
    import glob
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "raytracing_amd", "csrc", "*.h*"))):
        with open(path, "rb") as fp:
            h.update(os.path.basename(path).encode() + b"\0" + fp.read())
    return h.hexdigest()


def profiled_execution(n_objects):
    """What k_render_spheres executes per ray, from the newest committed rocprofv3 counter summary of this command
    (profiles/*_bench_c2.json, tools/profile_bench.sh): vector flops as the hardware counts them
    (SQ_INSTS_VALU_FLOPS_FP32 + _FP64 + _FP64_TRANS: flop per lane of every VALU wave-instruction, x 64 lanes whether or
    not a lane is active), VALU issue utilisation and the share of active lanes.  The kernel does not walk the flat
    list sphere by sphere (two-level screen), so its roofline is priced on these executed flops, not on the
    reference algorithm's.  None when no summary holds the counters."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_c2.json"))):
        try:
            with open(path) as fp:
                prof = json.load(fp)
            line = prof["bench_line_under_profiler"]
            c, d = prof["counters_per_launch"], prof["derived"]
            if line["config"]["objects"] != n_objects:
                continue
            rays = line["rays_per_sample"] * line["config"]["width"] * line["config"]["height"] * line["config"]["spp"]
            flop = 64.0 * (c["SQ_INSTS_VALU_FLOPS_FP32"] + c["SQ_INSTS_VALU_FLOPS_FP64"] + c["SQ_INSTS_VALU_FLOPS_FP64_TRANS"])
            best = {"flop_per_ray": flop / rays, "fp64_share": 64.0 * (c["SQ_INSTS_VALU_FLOPS_FP64"] + c["SQ_INSTS_VALU_FLOPS_FP64_TRANS"]) / flop,
                    "valu_issue_utilisation": d["valu_busy_estimate"], "lanes_active": d["valu_lane_utilisation"],
                    "source": os.path.relpath(path, ROOT), "kernel_source_sha256": prof.get("kernel_source_sha256")}
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return best


def profiled_traffic(width, height, spp, depth):
    """HBM-side bytes per launch of k_render_spheres from the newest committed rocprofv3 counter summary
    (profiles/*_bench_c2.json: FETCH_SIZE + WRITE_SIZE passes of this same command, see tools/profile_bench.sh),
    or None when there is none for this frame size.  The counters are L2 memory-side requests: almost all of
    them are the 64-bit accumulation atomics, which the 256 MB Infinity Cache absorbs."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_c2.json"))):
        try:
            with open(path) as fp:
                prof = json.load(fp)
            cfg = prof["bench_line_under_profiler"]["config"]
            c = prof["counters_per_launch"]
            if (cfg["width"], cfg["height"], cfg["spp"], cfg["max_depth"]) != (width, height, spp, depth):
                continue
            best = {"bytes": (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0, "source": os.path.relpath(path, ROOT)}
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return best


def cpu_baseline(scene_path):
    """The reference renderer itself (oracle/_ref/ref_b1, compiled from the unmodified sources in the
    development container), one thread, on a bounded sample of the same workload; falls back to the
    oracle's rand()-faithful port when that binary is absent."""
    W, H, spp = 240, 160, 96  # about 11 s of one core
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_b1")
    sample = f"config-2 scene and camera, {W}x{H}, {spp} spp, depth 50 ({W * H * spp / 1e6:.2f} Msamples)"
    if os.path.exists(ref):
        with tempfile.TemporaryDirectory() as tmp:
            t0 = time.perf_counter()
            subprocess.run([ref, "render", "c2", str(W), str(H), str(spp), "50", "-", os.path.join(tmp, "o.f64")],
                           check=True, capture_output=True)
            dt = time.perf_counter() - t0
        return {"value": W * H * spp / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "reference",
                "sample": sample + "; unmodified reference sources, g++ -O2, 1 thread"}
    from raytracing_amd import abi
    from tests import oracle_binding as ob
    from tests.conftest import CAM_C2
    scene = abi.load_scene(scene_path)
    cam = ob.camera(*CAM_C2)
    t0 = time.perf_counter()
    ob.render(scene, cam, abi.render_params(W, H, spp, 50), rng=ob.RNG_RAND, assoc=ob.ASSOC_REFERENCE,
              want_fixed=False)
    dt = time.perf_counter() - t0
    return {"value": W * H * spp / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": sample + "; oracle/rt_oracle.c, gcc -O2, 1 thread"}


def cpu_baseline_all_cores(processes):
    """The reference author's own recipe for more cores ("Ray Tracing In a Weekend/README.md":2538-2542): P copies of
    the reference renderer, each with its own seed and spp/P samples, images averaged afterwards.  Context only: P
    processes of the unmodified reference on the same bounded sample as cpu_baseline, started together."""
    W, H, spp = 240, 160, 96
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_b1")
    if not os.path.exists(ref) or processes < 2:
        return None
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        procs = [subprocess.Popen([ref, "render", "c2", str(W), str(H), str(spp), "50", str(1000 + k), os.path.join(tmp, f"o{k}.f64")],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for k in range(processes)]
        ok = all(p.wait() == 0 for p in procs)
        dt = time.perf_counter() - t0
    if not ok:
        return None
    return {"value": processes * W * H * spp / dt / 1e6, "unit": "Msamples/s", "cores": processes, "kind": "reference",
            "sample": f"{processes} processes of the unmodified reference, distinct srand() seeds, {W}x{H}x{spp} each, run together"}


class OneGpu:
    """N = 1: rt_context + rt_render_device into a torch buffer, one frame in flight per slot."""

    def __init__(self, torch, lib, abi, scene, cam, W, H, spp, depth, n_slots):
        self.torch, self.abi, self.cam, self.dims = torch, abi, cam, (W, H, spp, depth)
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        self.dev, self.slots = dev, []
        for _ in range(n_slots):
            s = torch.cuda.Stream(device=dev)
            c = lib.Context(0, s.cuda_stream)
            c.upload(scene)
            with torch.cuda.stream(s):
                buf = torch.zeros((H, W, 3), dtype=torch.int64, device=dev)
            self.slots.append((s, c, buf))

    def submit(self, k):
        W, H, spp, depth = self.dims
        s, c, buf = self.slots[k % len(self.slots)]
        with self.torch.cuda.stream(s):
            buf.zero_()
            c.render_device(self.cam, self.abi.render_params(W, H, spp, depth, seed=1 + k), buf.data_ptr())

    def wait_frame(self, k):
        self.slots[k % len(self.slots)][0].synchronize()

    def drain(self):
        self.torch.cuda.synchronize(self.dev)

    def take_stats(self):
        st = {"samples": 0, "rays": 0, "nan_samples": 0, "kernel_ms": 0.0, "kernel_launches": 0}
        for _, c, _ in self.slots:
            one = c.take_stats()
            for key in st:
                st[key] += one[key]
        return st

    def close(self):
        for _, c, _ in self.slots:
            c.close()


class stdout_to_stderr:
    """RCCL prints a version banner on stdout when its first communicator is made; this program's stdout carries one
    JSON line and nothing else, so file descriptor 1 points at stderr while a group is created."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        os.dup2(self.saved, 1)
        os.close(self.saved)


class ManyGpus:
    """N > 1: an rt_group (this process's members of it); frames go round its slots."""

    def __init__(self, group, abi, scene, cam, W, H, spp, depth):
        self.g, self.abi, self.cam, self.dims = group, abi, cam, (W, H, spp, depth)
        self.g.upload(scene)
        self.busy = [False] * group.frame_slots

    def submit(self, k):
        W, H, spp, depth = self.dims
        slot = k % len(self.busy)
        if self.busy[slot]:          # a slot is waited for before it takes another frame
            self.g.wait(slot)
        self.g.render_async(self.cam, self.abi.render_params(W, H, spp, depth, seed=1 + k), slot)
        self.busy[slot] = True

    def wait_frame(self, k):
        slot = k % len(self.busy)
        self.g.wait(slot)
        self.busy[slot] = False

    def drain(self):
        for slot in range(len(self.busy)):
            self.g.wait(slot)
            self.busy[slot] = False

    def take_stats(self):
        return self.g.take_stats()

    def close(self):
        self.g.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--spp", type=int, default=500)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames rendered concurrently, each on its own HIP stream and accumulation buffer, so that "
                         "the long-path tail of one frame and its all-reduce overlap the next frame.  Default: 1 on "
                         "one GPU (kernel timings stay per launch), 3 on several (with 2, frame k+2 would queue "
                         "behind the all-reduce of frame k on the same stream, and that collective only finds free "
                         "CUs once frame k+1 nears its end)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # RT_BENCH_FORCE_GROUP=1 (rehearsal on a one-GPU box): take the several-GPU path even with one GPU -- under
    # torch.distributed.run the rank path (id broadcast, rt_group_create_rank), else rt_group_create of one device
    force_group = os.environ.get("RT_BENCH_FORCE_GROUP") == "1"
    ranks = world > 1 or (force_group and "WORLD_SIZE" in os.environ)  # one process per GPU (torch.distributed.run)
    n_gpus = world if ranks else args.gpus                              # else this process drives them all
    if n_gpus < 1:
        sys.exit("--gpus must be at least 1")

    import torch
    from raytracing_amd import abi, lib

    if lib.load().rt_get_device_count() < (1 if ranks else n_gpus):
        sys.exit(f"bench.py needs {n_gpus} GPU(s), {lib.load().rt_get_device_count()} visible: the render path has no CPU fallback")

    dist = None
    if ranks:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane only (the RCCL id, barriers, the MAX of the ranks' times): gloo over loopback.  The frames'
        # all-reduce is the library's own RCCL communicator.
        dist.init_process_group(os.environ.get("RT_BENCH_CONTROL_BACKEND", "gloo"))

    scene_path = os.path.join(ROOT, "tests", "golden", "scene_c2.rtsc")
    scene = abi.load_scene(scene_path)
    n_objects = int(scene.nodes[scene.root]["child1"])
    cam = lib.camera((12, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 3.0 / 2.0, 0.1, 10.0)
    W, H, spp = args.width, args.height, args.spp
    n_slots = args.frames_in_flight if args.frames_in_flight > 0 else (1 if n_gpus == 1 and not force_group else 3)

    verbose = os.environ.get("RT_BENCH_VERBOSE") == "1"
    if verbose:
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("RT_BENCH_STALL_S", "60")), exit=True)

    def say(msg):
        if verbose:
            print(f"[rank {rank}] {msg}", file=sys.stderr, flush=True)

    if n_gpus == 1 and not force_group:
        run = OneGpu(torch, lib, abi, scene, cam, W, H, spp, args.depth, n_slots)
    elif ranks:
        ident = torch.zeros(abi.RT_COMM_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            with stdout_to_stderr():
                ident = torch.frombuffer(bytearray(lib.comm_id_create()), dtype=torch.uint8).clone()
        dist.broadcast(ident, src=0)
        say("RCCL id received")
        with stdout_to_stderr():
            group = lib.Group(device=local_rank, rank=rank, world=world, comm_id=bytes(ident.numpy().tobytes()), frame_slots=n_slots)
        run = ManyGpus(group, abi, scene, cam, W, H, spp, args.depth)
    else:
        with stdout_to_stderr():
            group = lib.Group(devices=list(range(n_gpus)), frame_slots=n_slots)
        run = ManyGpus(group, abi, scene, cam, W, H, spp, args.depth)
    say("contexts ready")

    def fence():
        run.drain()
        if ranks:
            dist.barrier()

    def over_ranks(value, op):
        if not ranks:
            return value
        t = torch.tensor(value, dtype=torch.float64 if isinstance(value[0], float) else torch.int64)
        dist.all_reduce(t, op=op)
        return t.tolist()

    for k in range(args.warmup):
        run.submit(k)
    fence()
    say("warm-up done")
    run.take_stats()  # drop warm-up statistics
    t0 = time.perf_counter()
    for k in range(args.steps):
        run.submit(args.warmup + k)
    fence()
    dt = time.perf_counter() - t0
    say("timed region done")
    st = run.take_stats()
    dt = over_ranks([dt], dist.ReduceOp.MAX if ranks else None)[0]
    samples, rays, bad = over_ranks([st["samples"], st["rays"], st["nan_samples"]], dist.ReduceOp.SUM if ranks else None)

    # A frame rendered alone: first launch to summed buffer complete on every GPU, K frames one after the other
    single = None
    if n_gpus > 1 or force_group:
        base = args.warmup + args.steps
        fence()
        t1 = time.perf_counter()
        for k in range(args.steps):
            run.submit(base + k)
            run.wait_frame(base + k)
        fence()
        ds = over_ranks([time.perf_counter() - t1], dist.ReduceOp.MAX if ranks else None)[0]
        run.take_stats()
        single = {"ms_per_frame": ds / args.steps * 1e3, "msamples_per_s": W * H * spp * args.steps / ds / 1e6,
                  "frames": args.steps, "note": "one frame in flight: zero, render, all-reduce, host wait, then the next"}

    if rank == 0:
        assert samples == W * H * spp * args.steps, (samples, W * H * spp * args.steps)
        value = samples / dt / 1e6
        # roofline of the dominant kernel (k_render_spheres) on THIS process's GPUs: executed flops of its launches
        # over their event-timed duration on the kernel's own stream (with several frames in flight launches
        # overlap each other, so the wall time per step is used instead: an upper bound on the kernel's time,
        # i.e. a lower bound on the fraction)
        launches = max(1, st["kernel_launches"])
        k_ms = st["kernel_ms"] / launches if n_slots == 1 else dt / args.steps * 1e3
        reference_flop = algorithmic_flop(n_objects, st["rays"], st["samples"]) / launches
        executed = profiled_execution(n_objects)
        flop_per_launch = executed["flop_per_ray"] * st["rays"] / launches if executed else None
        achieved = flop_per_launch / (k_ms * 1e-3) / 1e12 if executed and k_ms > 0 else None
        frac = achieved / PEAK_FP32_VALU_TFLOPS if achieved is not None else None
        traffic = profiled_traffic(W, H, spp, args.depth)
        stale = (executed["kernel_source_sha256"] != kernel_source_sha256()) if executed else None
        if ranks:
            how = f"one process per GPU, {world} ranks; RCCL communicator from ncclCommInitRank"
        elif n_gpus > 1:
            how = f"one process driving {n_gpus} GPUs; RCCL communicators from ncclCommInitAll"
        else:
            how = "one GPU, no collective"
        out = {
            "metric": "Msamples/sec (paths x spp / s), One-Weekend final scene",
            "value": value, "unit": "Msamples/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": value / PUBLISHED_MSAMPLES_S, "dtype": "f64 (f32 screening pass)",
            "data": "synthetic: the reference's random_scene() as rolled under g++/glibc, dumped to tests/golden/scene_c2.rtsc",
            "config": {"workload": "One Weekend final random-spheres scene (BASELINE configs[1])", "objects": n_objects,
                       "width": W, "height": H, "spp": spp, "max_depth": args.depth,
                       "parallelism": f"spp split over {n_gpus} GPU(s), one int64 RCCL all-reduce per frame inside the library ({how})",
                       "frames_in_flight": n_slots},
            "value_is": "throughput over the timed frames" + ("" if n_slots == 1 else f", {n_slots} frames in flight (single_frame = a frame rendered alone)"),
            "single_frame": single,
            "mrays_per_s": rays / dt / 1e6, "rays_per_sample": rays / max(1, samples), "nan_samples": bad,
            "roofline": {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": frac,
                         "frac_active_lanes": frac * executed["lanes_active"] if frac is not None else None,
                         "traffic": traffic["bytes"] if traffic else None,
                         "traffic_source": traffic["source"] if traffic else None,
                         "kernel": "k_render_spheres", "kernel_ms": k_ms,
                         "flop_per_launch": flop_per_launch,
                         "flop_source": executed["source"] if executed else None,
                         "flop_source_stale": stale,
                         "fp64_share_of_flop": executed["fp64_share"] if executed else None,
                         "valu_issue_utilisation": executed["valu_issue_utilisation"] if executed else None,
                         "lanes_active": executed["lanes_active"] if executed else None,
                         "reference_algorithm_tflops": reference_flop / (k_ms * 1e-3) / 1e12 if k_ms > 0 else None,
                         "note": "FP32 vector roofline.  The kernel screens the list through a two-level "
                                 "structure, so flop = EXECUTED vector flops (hardware counters of the committed "
                                 "profile of this command, per ray, x this run's rays; 64 lanes per "
                                 "wave-instruction whether or not a lane is active: frac_active_lanes = frac x "
                                 "the share of active lanes), not the flat-list price of the reference algorithm "
                                 "(reference_algorithm_tflops = what that price would give, context only).  "
                                 "flop_source_stale: the kernel sources have changed since that profile was taken "
                                 "(kernel_ms, value and the ray counts are always this run's own).  "
                                 "The kernel is bound by VALU issue (valu_issue_utilisation: share of cycles a "
                                 "SIMD issues a vector instruction), most of it double-precision, integer and "
                                 "control work that the flop count does not see.  traffic = L2 memory-side "
                                 "bytes per launch from the committed counter passes (64-bit accumulation "
                                 "atomics, absorbed by the Infinity Cache; algorithmic bytes are the 23 MB frame)"},
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene_path)
            many = cpu_baseline_all_cores(int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or min(16, os.cpu_count() or 1))
            if many:
                out["cpu_baseline_all_cores"] = many
        print(json.dumps(out), flush=True)

    run.close()
    if ranks:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
