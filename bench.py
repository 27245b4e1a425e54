#!/usr/bin/env python3
"""Headline benchmark: Msamples/s on the One-Weekend final scene (BASELINE.json configs[1]).

A step is one full frame of the reference's render loop over the dumped 487-sphere world:
1200x800 pixels, 500 samples per pixel, depth 50, camera of RayTracing.cpp:104-110.  With N
ranks (one process per GPU, launched by torch.distributed.run) the 500 samples of every pixel
are split 63/62 over the ranks and the fixed-point accumulation buffers are summed by one RCCL
all-reduce per frame, so the N-GPU frame equals the 1-GPU frame bit for bit ("strong" scaling:
total work is fixed).  Scene and accumulation buffer are resident in HBM before the timed
region; tone mapping and PPM output are outside it.

Prints ONE JSON line on rank 0.
"""
import argparse
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
                    "source": os.path.relpath(path, ROOT)}
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
                    help="frames rendered concurrently on separate HIP streams (each with its own accumulation "
                         "buffer), so that the long-path tail of one frame and its all-reduce overlap the next "
                         "frame.  Default: 1 on one GPU (kernel timings stay per launch), 3 with several ranks (with 2, "
                         "frame k+2 would queue behind the all-reduce of frame k on the same stream, and that collective "
                         "only finds free CUs once frame k+1 nears its end)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracing_amd import abi, frame, lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the render path has no CPU fallback")
    # Rehearsal on a one-GPU box: RT_BENCH_SHARE_GPU=1 puts every rank on device 0 and RT_BENCH_BACKEND=gloo
    # replaces RCCL (which refuses two ranks on one device); the normal path is one GPU per rank over nccl.
    if os.environ.get("RT_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_reduce = os.environ.get("RT_BENCH_FORCE_REDUCE") == "1"   # rehearsal: collectives even with one rank
    if world > 1 or force_reduce:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    scene_path = os.path.join(ROOT, "tests", "golden", "scene_c2.rtsc")
    scene = abi.load_scene(scene_path)
    n_objects = int(scene.nodes[scene.root]["child1"])
    cam = lib.camera((12, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 3.0 / 2.0, 0.1, 10.0)
    W, H, spp = args.width, args.height, args.spp

    # One slot per frame in flight: its own HIP stream, library context (the scene is tiny: one copy per slot)
    # and accumulation buffer.  Frame k runs entirely on slot k % n: zero -> render kernel -> all-reduce
    # (frame.FramePipeline orders the all-reduce of frame k-1 after the launch of frame k).
    n_slots = args.frames_in_flight if args.frames_in_flight > 0 else (1 if world == 1 else 3)
    slots = []
    for _ in range(n_slots):
        s = torch.cuda.Stream(device=dev)
        c = lib.Context(local_rank, s.cuda_stream)
        c.upload(scene)
        with torch.cuda.stream(s):
            buf = torch.zeros((H, W, 3), dtype=torch.int64, device=dev)
        slots.append((s, c, buf))

    verbose = os.environ.get("RT_BENCH_VERBOSE") == "1"
    if verbose:
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("RT_BENCH_STALL_S", "30")), exit=True)

    def say(msg):
        if verbose:
            print(f"[rank {rank}] {msg}", file=sys.stderr, flush=True)

    def reduce_fn(t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    class Slot:
        def __init__(self, s, c, buf):
            self.s, self.c, self.accum = s, c, buf

        def stream(self):
            return torch.cuda.stream(self.s)

        def drain(self):
            self.s.synchronize()

        def render(self, begin, count, accum, k):
            self.c.render_device(cam, abi.render_params(W, H, count, args.depth, seed=1 + k, sample_begin=begin),
                                 accum.data_ptr())

    pipe = frame.FramePipeline([Slot(*t) for t in slots], spp, rank, world, reduce_fn if world > 1 else None)
    if force_reduce:
        pipe.world, pipe.all_reduce = max(world, 2), reduce_fn
    step = pipe.submit

    def fence():
        pipe.flush()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    say("contexts ready")
    for k in range(args.warmup):
        step(k)
        say(f"warm-up step {k} enqueued")
    fence()
    say("warm-up done")
    for _, c, _ in slots:
        c.take_stats()  # drop warm-up statistics
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    say("timed steps enqueued")
    fence()
    dt = time.perf_counter() - t0
    say("timed region done")
    st = {"samples": 0, "rays": 0, "nan_samples": 0, "kernel_ms": 0.0, "kernel_launches": 0}
    for _, c, _ in slots:
        one = c.take_stats()
        for key in st:
            st[key] += one[key]

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    counts = torch.tensor([st["samples"], st["rays"], st["nan_samples"]], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    samples, rays, bad = (int(x) for x in counts.tolist())

    if rank == 0:
        assert samples == W * H * spp * args.steps, (samples, W * H * spp * args.steps)
        value = samples / dt / 1e6
        # roofline of the dominant kernel (k_render_spheres) on THIS rank: algorithmic flops of its launches
        # over their event-timed duration on the kernel's own stream
        # (with several frames in flight launches overlap each other, so the wall time per step is used
        # instead: an upper bound on the kernel's time, i.e. a lower bound on the fraction)
        k_ms = st["kernel_ms"] / max(1, st["kernel_launches"]) if n_slots == 1 else dt / args.steps * 1e3
        launches = max(1, st["kernel_launches"])
        reference_flop = algorithmic_flop(n_objects, st["rays"], st["samples"]) / launches
        executed = profiled_execution(n_objects)
        flop_per_launch = executed["flop_per_ray"] * st["rays"] / launches if executed else None
        achieved = flop_per_launch / (k_ms * 1e-3) / 1e12 if executed and k_ms > 0 else None
        traffic = profiled_traffic(W, H, spp, args.depth)
        out = {
            "metric": "Msamples/sec (paths x spp / s), One-Weekend final scene",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": value / PUBLISHED_MSAMPLES_S, "dtype": "f64 (f32 screening pass)",
            "data": "synthetic: the reference's random_scene() as rolled under g++/glibc, dumped to tests/golden/scene_c2.rtsc",
            "config": {"workload": "One Weekend final random-spheres scene (BASELINE configs[1])", "objects": n_objects,
                       "width": W, "height": H, "spp": spp, "max_depth": args.depth,
                       "parallelism": f"spp split over {world} rank(s), one int64 all-reduce per frame",
                       "frames_in_flight": n_slots},
            "mrays_per_s": rays / dt / 1e6, "rays_per_sample": rays / max(1, samples), "nan_samples": bad,
            "roofline": {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_VALU_TFLOPS if achieved is not None else None,
                         "traffic": traffic["bytes"] if traffic else None,
                         "traffic_source": traffic["source"] if traffic else None,
                         "kernel": "k_render_spheres", "kernel_ms": k_ms,
                         "flop_per_launch": flop_per_launch,
                         "flop_source": executed["source"] if executed else None,
                         "fp64_share_of_flop": executed["fp64_share"] if executed else None,
                         "valu_issue_utilisation": executed["valu_issue_utilisation"] if executed else None,
                         "lanes_active": executed["lanes_active"] if executed else None,
                         "reference_algorithm_tflops": reference_flop / (k_ms * 1e-3) / 1e12 if k_ms > 0 else None,
                         "note": "FP32 vector roofline.  The kernel screens the list through a two-level "
                                 "structure, so flop = EXECUTED vector flops (hardware counters of the committed "
                                 "profile of this command, per ray, x this run's rays; 64 lanes per "
                                 "wave-instruction), not the flat-list price of the reference algorithm "
                                 "(reference_algorithm_tflops = what that price would give, context only).  "
                                 "The kernel is bound by VALU issue (valu_issue_utilisation: share of cycles a "
                                 "SIMD issues a vector instruction), most of it double-precision, integer and "
                                 "control work that the flop count does not see.  traffic = L2 memory-side "
                                 "bytes per launch from the committed counter passes (64-bit accumulation "
                                 "atomics, absorbed by the Infinity Cache; algorithmic bytes are the 23 MB frame)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene_path)
            many = cpu_baseline_all_cores(int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or min(16, os.cpu_count() or 1))
            if many:
                out["cpu_baseline_all_cores"] = many
        print(json.dumps(out), flush=True)

    for _, c, _ in slots:
        c.close()
    if world > 1 or force_reduce:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
