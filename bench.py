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
  python3 bench.py                         one GPU, rt_context
  python -m torch.distributed.run ... bench.py --gpus N    one process per GPU: rank 0 makes the RCCL id, hands it
                                           round over torch.distributed (gloo, control plane only: id, barriers, the
                                           MAX of the ranks' times), every rank joins with rt_group_create_rank
  python3 bench.py --gpus N                one process driving N GPUs (rt_group_create = ncclCommInitAll)

Prints ONE JSON line on rank 0.  What is timed, at every N alike:
  value / single_frame   K frames one after the other, ONE frame in flight: zero, render, all-reduce, and the next
                         frame behind it -- per frame the time from first launch to summed buffer complete on every
                         GPU.  This is `value`, at N = 1 and at N > 1, so scaling compares like with like.
  pipelined              the same K frames with three frames in flight (own streams and buffers): the end of one
                         frame's launch and its all-reduce overlap the next frame's kernel.  Measured after the
                         headline, at every GPU count (--no-pipelined skips it); on several GPUs a watchdog prints the
                         line without it should that phase stall.
Beside them, on one GPU: the correctness gate of the headline run itself (the last timed frame against the image of
the reference program: RMSE, PSNR, image mean), the other BASELINE configurations (`configs`), each with its kernel's
roofline from the committed counter profile of that configuration, and the reference's own CPU renderer timed on the
host cores (before the GPU region).
"""
import argparse
import glob
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
GOLD = os.path.join(ROOT, "tests", "golden")

PEAK_FP32_VALU_TFLOPS = 157.3      # MI355X vector FP32 peak: 256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz
PEAK_FP64_VALU_TFLOPS = 78.6       # vector FP64: half of it
PUBLISHED_MSAMPLES_S = 480.0e6 / 36000.0 / 1e6  # "ran ten hours" for this exact frame (reference README.md:2510)
PIPELINE_SLOTS = 3


def _b2_camera(lib, name):
    with open(os.path.join(GOLD, "kat_b2.json")) as fp:
        k = json.load(fp)[name]
    return lib.camera(k["lookfrom"], k["lookat"], (0, 1, 0), k["vfov"], k["aspect"], k["aperture"], 10.0, 0.0, 1.0)


# The five BASELINE.json configurations: scene fixture, camera, frame, flags, the kernel that renders it, the image of
# the reference program the frame is gated against (None: no such image is committed), and the gate's tolerance.
CONFIGS = {
    "c1": dict(workload="One Weekend three-sphere scene (BASELINE configs[0])", scene="scene_c1.rtsc", width=400, height=225, spp=100,
               camera=lambda lib: lib.camera((3, 3, 2), (0, 0, -1), (0, 1, 0), 20.0, 16.0 / 9.0, 2.0, 27.0 ** 0.5),
               flags=0, kernel="k_render_spheres<false>", reference=None, tol=None),
    "c2": dict(workload="One Weekend final random-spheres scene (BASELINE configs[1])", scene="scene_c2.rtsc", width=1200, height=800, spp=500,
               camera=lambda lib: lib.camera((12, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 3.0 / 2.0, 0.1, 10.0),
               flags=0, kernel="k_render_spheres<false>", reference="c2_full_ref.npz", tol=1.1),
    "c3": dict(workload="The Next Week bouncing spheres (BASELINE configs[2])", scene="scene_b2_bouncing.rtsc", width=1200, height=675, spp=1000,
               camera=lambda lib: _b2_camera(lib, "bouncing"), flags=0, kernel="k_render_spheres<true>", reference="c3_full_ref.npz", tol=1.1),
    "c4": dict(workload="The Next Week final scene (BASELINE configs[3])", scene="scene_b2_final.rtsc", width=800, height=800, spp=10000,
               camera=lambda lib: _b2_camera(lib, "final"), flags=0, kernel="k_render_scene", reference="c4_full_ref.npz", tol=1.25),
    "c5": dict(workload="The Rest Of Your Life Cornell box, light sampled by importance (BASELINE configs[4]; no reference code exists: parity unpinned)",
               scene="scene_b3_cornell.rtsc", width=600, height=600, spp=1000,
               camera=lambda lib: lib.camera((278, 278, -800), (278, 278, 0), (0, 1, 0), 40.0, 1.0, 0.0, 10.0, 0.0, 1.0),
               flags="importance", kernel="k_render_scene", reference=None, tol=None),
}


def algorithmic_flop(n_objects, rays, samples):
    """Flops of the REFERENCE algorithm for this scene: a flat list tested in full for every ray.
    23 flop per sphere test up to the discriminant sign (3 sub, 5 + 5 for the two dot products,
    7 for c and the discriminant, 3 for comparisons), 60 per ray for the accepted hit, the scatter
    and the sky, 40 per sample for jitter and camera ray."""
    return rays * (n_objects * 23.0 + 60.0) + samples * 40.0


def kernel_source_sha256():
    """Hash of the kernel sources the library is built from: a committed counter summary carries the hash of the tree
    it was measured on, so figures taken from it can be marked stale once the kernels have changed."""
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "raytracing_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "raytracing_amd", "csrc", "*.inc"))):
        with open(path, "rb") as fp:
            h.update(os.path.basename(path).encode() + b"\0" + fp.read())
    return h.hexdigest()


def profiled(name):
    """What the kernel of configuration `name` executes, from the newest committed rocprofv3 counter summary of that
    configuration (profiles/*_config_<name>.json, made by tools/profile_config.sh from tools/bench_config.py under the
    profiler): vector flops as the hardware counts them (SQ_INSTS_VALU_FLOPS_FP32 + _FP64 + _FP64_TRANS: flop per lane
    of every VALU wave-instruction, x 64 lanes whether or not a lane is active) per ray of the profiled run, VALU issue
    utilisation, share of active lanes, L2 hit rate and HBM-side bytes per launch.  None when there is none."""
    best = None
    paths = glob.glob(os.path.join(ROOT, "profiles", f"*_config_{name}.json")) + glob.glob(os.path.join(ROOT, "profiles", f"*_bench_{name}.json"))
    for path in sorted(paths, key=os.path.basename):
        try:
            with open(path) as fp:
                prof = json.load(fp)
            line = prof["bench_line_under_profiler"]
            if "config" in line:  # a line of this program (tools/profile_bench.sh): one launch = one frame of its config
                cf = line["config"]
                line = {"width": cf["width"], "height": cf["height"], "spp": cf["spp"], "kernel_ms": line["roofline"]["kernel_ms"],
                        "rays": line["rays_per_sample"] * cf["width"] * cf["height"] * cf["spp"]}
            c, d = prof["counters_per_launch"], prof["derived"]
            flop = 64.0 * (c["SQ_INSTS_VALU_FLOPS_FP32"] + c["SQ_INSTS_VALU_FLOPS_FP64"] + c["SQ_INSTS_VALU_FLOPS_FP64_TRANS"])
            best = {"flop_per_ray": flop / line["rays"],
                    "fp64_share": 64.0 * (c["SQ_INSTS_VALU_FLOPS_FP64"] + c["SQ_INSTS_VALU_FLOPS_FP64_TRANS"]) / flop,
                    "valu_issue_utilisation": d.get("valu_busy_estimate"), "lanes_active": d.get("valu_lane_utilisation"),
                    "l2_hit_rate": d.get("l2_hit_rate"),
                    "wait_share_of_wave_cycles": d.get("SQ_WAIT_ANY_share_of_wave_cycles"),
                    "bytes_per_launch": (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 if "FETCH_SIZE" in c and "WRITE_SIZE" in c else None,
                    "profiled_frame": {k: line[k] for k in ("width", "height", "spp", "kernel_ms")},
                    "source": os.path.relpath(path, ROOT), "kernel_source_sha256": prof.get("kernel_source_sha256")}
        except (OSError, KeyError, ValueError, TypeError, ZeroDivisionError):
            continue
    return best


def roofline_of(name, cfg, rays, kernel_ms, launches=1):
    """The roofline object of one configuration's kernel: executed vector flops (the committed profile's flop per ray x
    this run's rays) over this run's event-timed kernel time, against the FP32 vector peak."""
    ex = profiled(name)
    out = {"bound": "valu", "achieved": None, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
           "kernel": cfg["kernel"], "kernel_ms": kernel_ms}
    if ex is None or kernel_ms <= 0:
        out["note"] = "no committed counter profile of this configuration"
        return out
    flop = ex["flop_per_ray"] * rays / launches
    achieved = flop / (kernel_ms * 1e-3) / 1e12
    fr = ex["profiled_frame"]
    out.update({
        "achieved": achieved, "frac": achieved / PEAK_FP32_VALU_TFLOPS,
        "frac_active_lanes": achieved / PEAK_FP32_VALU_TFLOPS * ex["lanes_active"] if ex["lanes_active"] else None,
        "traffic": ex["bytes_per_launch"], "traffic_is": f"HBM-side bytes of one {fr['width']}x{fr['height']}x{fr['spp']} launch under the profiler",
        "hbm_gbs": ex["bytes_per_launch"] / (fr["kernel_ms"] * 1e-3) / 1e9 if ex["bytes_per_launch"] else None,
        "l2_hit_rate": ex["l2_hit_rate"], "flop_per_launch": flop, "fp64_share_of_flop": ex["fp64_share"],
        "fp64_frac_of_fp64_peak": achieved * ex["fp64_share"] / PEAK_FP64_VALU_TFLOPS,
        "valu_issue_utilisation": ex["valu_issue_utilisation"], "lanes_active": ex["lanes_active"],
        "wait_share_of_wave_cycles": ex["wait_share_of_wave_cycles"],
        "flop_source": ex["source"], "flop_source_stale": ex["kernel_source_sha256"] != kernel_source_sha256()})
    return out


def gate_against_reference(lib, name, cfg, sums, spp):
    """The correctness gate of tests/test_gpu_fullsize.py on a frame this run rendered: per-pixel RMSE of linear radiance
    against the image of the reference program (mean of 8 reference processes x 32 spp, distinct srand seeds; sigma =
    their per-sample deviation) within tol x sigma x sqrt(1/N + 1/M) -- what two correct, independent renders differ
    by --, image mean within three standard errors, PSNR of the 8-bit gamma images (B1/color.h:8-23) stated."""
    if cfg["reference"] is None:
        return None
    ref = np.load(os.path.join(GOLD, cfg["reference"]))
    want = ref["radiance"].astype(np.float64)
    H, W, _ = sums.shape
    if want.shape != sums.shape:
        return None
    M, sigma = int(ref["spp"]), float(ref["sigma"])
    img = sums / spp
    r = float(np.sqrt(np.mean((img - want) ** 2)))
    expect = sigma * np.sqrt(1.0 / spp + 1.0 / M)
    ga, gb = lib.write_color(img, 1).astype(np.float64), lib.write_color(want, 1).astype(np.float64)
    psnr = float(10 * np.log10(255.0 ** 2 / max(1e-12, np.mean((ga - gb) ** 2))))
    mean_diff = float(np.abs(img.reshape(-1, 3).mean(0) - want.reshape(-1, 3).mean(0)).max())
    se = sigma * np.sqrt(3.0) * np.sqrt((1.0 / spp + 1.0 / M) / (W * H))
    ok = bool(r <= cfg["tol"] * expect and mean_diff <= 3.0 * se + 2e-4)
    return {"rmse": r, "expected": float(expect), "tolerance": cfg["tol"], "psnr_db": psnr, "mean_diff": mean_diff,
            "mean_diff_allowed": float(3.0 * se + 2e-4), "reference": f"tests/golden/{cfg['reference']} (reference program, {M} spp)",
            "pass": ok}


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
    """N = 1: rt_context + rt_render_device into a torch buffer per slot; frame k goes to slot k % use."""

    def __init__(self, torch, lib, abi, scene, cam, W, H, spp, depth, n_slots):
        self.torch, self.abi, self.cam, self.dims = torch, abi, cam, (W, H, spp, depth)
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        self.dev, self.slots, self.use = dev, [], 1
        for _ in range(n_slots):
            s = torch.cuda.Stream(device=dev)
            c = lib.Context(0, s.cuda_stream)
            c.upload(scene)
            with torch.cuda.stream(s):
                buf = torch.zeros((H, W, 3), dtype=torch.int64, device=dev)
            self.slots.append((s, c, buf))

    def submit(self, k):
        W, H, spp, depth = self.dims
        s, c, buf = self.slots[k % self.use]
        with self.torch.cuda.stream(s):
            buf.zero_()
            c.render_device(self.cam, self.abi.render_params(W, H, spp, depth, seed=1 + k), buf.data_ptr())

    def drain(self):
        self.torch.cuda.synchronize(self.dev)

    def read_frame(self, k):
        W, H, _, _ = self.dims
        s, c, buf = self.slots[k % self.use]
        return c.read_sums(buf.data_ptr(), W, H)

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
    """N > 1: an rt_group (this process's members of it); frame k goes to slot k % use."""

    def __init__(self, group, abi, scene, cam, W, H, spp, depth):
        self.g, self.abi, self.cam, self.dims = group, abi, cam, (W, H, spp, depth)
        self.g.upload(scene)
        self.g.reserve(W, H)  # every slot's buffers now: the frames only enqueue work
        self.busy = [False] * group.frame_slots
        self.use = 1

    def submit(self, k):
        W, H, spp, depth = self.dims
        slot = k % self.use
        if self.busy[slot]:          # a slot is waited for before it takes another frame
            self.g.wait(slot)
        self.g.render_async(self.cam, self.abi.render_params(W, H, spp, depth, seed=1 + k), slot)
        self.busy[slot] = True

    def drain(self):
        for slot in range(len(self.busy)):
            if self.busy[slot]:
                self.g.wait(slot)
            self.busy[slot] = False

    def read_frame(self, k):
        W, H, _, _ = self.dims
        return self.g.read(W, H, slot=k % self.use)

    def take_stats(self):
        return self.g.take_stats()

    def close(self):
        self.g.close()


def render_config(lib, abi, ctx, name, cfg):
    """One frame of a BASELINE configuration at its own size on one GPU: a small warm-up launch (code objects, staging),
    then the frame -- wall time of rt_render_device to completion, event-timed kernel time, and the gate where an image
    of the reference program is committed."""
    scene = abi.load_scene(os.path.join(GOLD, cfg["scene"]))
    cam = cfg["camera"](lib)
    flags = abi.RT_RENDER_IMPORTANCE if cfg["flags"] == "importance" else 0
    W, H, spp = cfg["width"], cfg["height"], cfg["spp"]
    ctx.upload(scene)
    ctx.render(cam, abi.render_params(W, H, 4, 50, seed=9, flags=flags))
    t0 = time.perf_counter()
    sums, st = ctx.render(cam, abi.render_params(W, H, spp, 50, seed=1, flags=flags))
    wall = time.perf_counter() - t0
    k_ms = st["kernel_ms"] / max(1, st["kernel_launches"])
    out = {"workload": cfg["workload"], "width": W, "height": H, "spp": spp, "max_depth": 50,
           "msamples_per_s": st["samples"] / k_ms / 1e3, "mrays_per_s": st["rays"] / k_ms / 1e3,
           "rays_per_sample": st["rays"] / max(1, st["samples"]), "nan_samples": st["nan_samples"], "kernel_ms": k_ms,
           "frame_ms_with_readback": wall * 1e3, "frames_timed": 1,
           "roofline": roofline_of(name, cfg, st["rays"], k_ms), "rmse_gate": gate_against_reference(lib, name, cfg, sums, spp)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--spp", type=int, default=500)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--frames-in-flight", type=int, default=1,
                    help="frames in flight of the headline measurement (`value`).  Default 1 at every GPU count: a frame "
                         "is timed from its first launch to its summed buffer, the next one starts behind it")
    ap.add_argument("--no-pipelined", action="store_true",
                    help=f"skip the second measurement, the throughput with {PIPELINE_SLOTS} frames in flight (on several GPUs "
                         "successive all-reduces of one communicator then run on different streams, ordered by the "
                         "library's event chain)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configurations (`configs` block)")
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
    one_gpu = n_gpus == 1 and not force_group

    import torch
    from raytracing_amd import abi, lib

    if lib.load().rt_get_device_count() < (1 if ranks else n_gpus):
        sys.exit(f"bench.py needs {n_gpus} GPU(s), {lib.load().rt_get_device_count()} visible: the render path has no CPU fallback")

    scene_path = os.path.join(GOLD, "scene_c2.rtsc")
    # The reference's CPU renderer first (rank 0 of a one-GPU run): the GPU region then closes the run
    cpu_one = cpu_many = None
    if n_gpus == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu_one = cpu_baseline(scene_path)
        cpu_many = cpu_baseline_all_cores(int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or min(16, os.cpu_count() or 1))

    dist = None
    if ranks:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane only (the RCCL id, barriers, the MAX of the ranks' times): gloo over loopback.  The frames'
        # all-reduce is the library's own RCCL communicator.
        dist.init_process_group(os.environ.get("RT_BENCH_CONTROL_BACKEND", "gloo"))

    scene = abi.load_scene(scene_path)
    n_objects = int(scene.nodes[scene.root]["child1"])
    cfg2 = CONFIGS["c2"]
    cam = cfg2["camera"](lib)
    W, H, spp = args.width, args.height, args.spp
    in_flight = max(1, args.frames_in_flight)
    do_pipelined = not args.no_pipelined and in_flight < PIPELINE_SLOTS
    n_slots = max(in_flight, PIPELINE_SLOTS if do_pipelined else 1)

    verbose = os.environ.get("RT_BENCH_VERBOSE") == "1"
    if verbose:
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("RT_BENCH_STALL_S", "60")), exit=True)

    def say(msg):
        if verbose:
            print(f"[rank {rank}] {msg}", file=sys.stderr, flush=True)

    if one_gpu:
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

    def timed(first_frame, use):
        """W warm-up frames, then exactly K frames between two fences, `use` frames in flight; MAX over ranks."""
        run.use = use
        for k in range(args.warmup):
            run.submit(first_frame + k)
        fence()
        run.take_stats()  # drop warm-up statistics
        t0 = time.perf_counter()
        for k in range(args.steps):
            run.submit(first_frame + args.warmup + k)
        fence()
        dt = time.perf_counter() - t0
        st = run.take_stats()
        dt = over_ranks([dt], dist.ReduceOp.MAX if ranks else None)[0]
        tot = over_ranks([st["samples"], st["rays"], st["nan_samples"]], dist.ReduceOp.SUM if ranks else None)
        return dt, st, tot

    # ---- the headline: K frames, `in_flight` at a time (default: one)
    dt, st, (samples, rays, bad) = timed(0, in_flight)
    say("timed region done")
    last_frame = args.warmup + args.steps - 1
    gate = None
    if rank == 0:
        run.use = in_flight
        gate = gate_against_reference(lib, "c2", cfg2, run.read_frame(last_frame), spp) if args.depth == 50 else None

    # What rank 0 prints; called once -- at the end, or by the watchdog of the pipelined phase.
    def report(pipelined, configs):
        if rank != 0:
            return
        assert samples == W * H * spp * args.steps, (samples, W * H * spp * args.steps)
        value = samples / dt / 1e6
        # roofline of the dominant kernel (k_render_spheres) on THIS process's GPUs: executed flops of its launches
        # over their event-timed duration on the kernel's own stream (with several frames in flight launches
        # overlap each other, so the wall time per step is used instead: an upper bound on the kernel's time,
        # i.e. a lower bound on the fraction)
        launches = max(1, st["kernel_launches"])
        k_ms = st["kernel_ms"] / launches if in_flight == 1 else dt / args.steps * 1e3
        roof = roofline_of("c2", cfg2, st["rays"], k_ms, launches)
        roof["reference_algorithm_tflops"] = algorithmic_flop(n_objects, st["rays"], st["samples"]) / launches / (k_ms * 1e-3) / 1e12 if k_ms > 0 else None
        roof["note"] = ("FP32 vector roofline.  The kernel screens the list through a two-level structure, so flop = EXECUTED "
                        "vector flops (hardware counters of the committed profile of this configuration, per ray, x this run's "
                        "rays; 64 lanes per wave-instruction whether or not a lane is active: frac_active_lanes = frac x the "
                        "share of active lanes), not the flat-list price of the reference algorithm "
                        "(reference_algorithm_tflops = what that price would give, context only).  flop_source_stale: the "
                        "kernel sources have changed since that profile was taken (kernel_ms, value and the ray counts are "
                        "always this run's own).  The kernel is bound by VALU issue (valu_issue_utilisation: share of cycles "
                        "a SIMD issues a vector instruction), most of it double-precision, integer and control work that the "
                        "flop count does not see.  traffic = L2 memory-side bytes per launch from the committed counter passes "
                        "(algorithmic bytes are the 23 MB frame)")
        if ranks:
            how = f"one process per GPU, {world} ranks; RCCL communicator from ncclCommInitRank"
        elif n_gpus > 1:
            how = f"one process driving {n_gpus} GPUs; RCCL communicators from ncclCommInitAll"
        else:
            how = "one GPU, no collective"
        single = {"ms_per_frame": dt / args.steps * 1e3, "msamples_per_s": value, "frames": args.steps,
                  "note": "one frame in flight: zero, render, all-reduce, and the next frame behind it"} if in_flight == 1 else None
        out = {
            "metric": "Msamples/sec (paths x spp / s), One-Weekend final scene",
            "value": value, "unit": "Msamples/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": value / PUBLISHED_MSAMPLES_S, "dtype": "f64 (f32 screening pass)",
            "data": "synthetic: the reference's random_scene() as rolled under g++/glibc, dumped to tests/golden/scene_c2.rtsc",
            "config": {"workload": cfg2["workload"], "objects": n_objects,
                       "width": W, "height": H, "spp": spp, "max_depth": args.depth,
                       "parallelism": f"spp split over {n_gpus} GPU(s), one int64 RCCL all-reduce per frame inside the library ({how})",
                       "frames_in_flight": in_flight},
            "value_is": ("single_frame: K frames one after the other, one in flight -- the same at every GPU count"
                         if in_flight == 1 else f"throughput with {in_flight} frames in flight"),
            "single_frame": single,
            "pipelined": pipelined,
            "mrays_per_s": rays / dt / 1e6, "rays_per_sample": rays / max(1, samples), "nan_samples": bad,
            "rmse_gate": gate,
            "roofline": roof,
        }
        if cpu_one is not None:
            out["cpu_baseline"] = cpu_one
        if cpu_many is not None:
            out["cpu_baseline_all_cores"] = cpu_many
        if configs is not None:
            out["configs"] = configs
        print(json.dumps(out), flush=True)

    # ---- the same frames with three in flight.  On several GPUs this is the first time successive all-reduces of one
    # communicator run on different streams beside persistent kernels that fill every CU: should the phase stall, every
    # rank's watchdog ends its process, rank 0's after printing the line with the headline alone.
    pipelined = None
    if rank == 0:
        print(f"bench.py: headline measured: {samples / dt / 1e6:.1f} Msamples/s, {dt / args.steps * 1e3:.3f} ms per frame on {n_gpus} GPU(s), "
              f"one frame in flight" if in_flight == 1 else f"bench.py: headline measured with {in_flight} frames in flight", file=sys.stderr, flush=True)
    if do_pipelined:
        import threading
        limit = float(os.environ.get("RT_BENCH_PIPELINED_TIMEOUT_S", "120"))

        def give_up():
            report({"frames_in_flight": PIPELINE_SLOTS, "ms_per_frame": None, "msamples_per_s": None,
                    "note": f"not measured: the phase did not finish within {limit:.0f} s"}, None)
            os._exit(0)

        watchdog = threading.Timer(limit, give_up) if not one_gpu else None
        if watchdog:
            watchdog.daemon = True
            watchdog.start()
        dp, _, (psamples, _, _) = timed(args.warmup + args.steps, PIPELINE_SLOTS)
        if watchdog:
            watchdog.cancel()
        pipelined = {"frames_in_flight": PIPELINE_SLOTS, "ms_per_frame": dp / args.steps * 1e3, "msamples_per_s": psamples / dp / 1e6,
                     "frames": args.steps, "note": "the end of a frame's launch and its all-reduce overlap the next frame's kernel"}
        say("pipelined region done")
    run.close()

    # ---- the other BASELINE configurations, one frame each at their own size (one GPU only)
    configs = None
    if one_gpu and rank == 0 and not args.no_configs:
        configs = {}
        with lib.Context(0) as ctx:
            for name in ("c1", "c3", "c4", "c5"):
                configs[name] = render_config(lib, abi, ctx, name, CONFIGS[name])
                say(f"{name} done")

    report(pipelined, configs)

    if ranks:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
