#!/usr/bin/env python3
"""Per-wave timeline of one k_render_spheres launch from a build with -DRT_TAIL_PROBE (tools/libprobe.so):
when the work queue ran dry for each wave, when the wave left, and how full its lanes were afterwards."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RT_LIB_PATH", os.path.join(ROOT, "tools", "libprobe.so"))
OUT = os.path.join(ROOT, "gpurun_out", "probe.bin")
os.environ["RT_TAIL_PROBE_FILE"] = OUT
from raytracing_amd import abi, lib  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 63
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 50
which = sys.argv[3] if len(sys.argv) > 3 else "c2"   # c2 | bouncing (config 3, the book-2 variant of the kernel)
if which == "bouncing":
    scene = abi.load_scene(os.path.join(ROOT, "tests", "golden", "scene_b2_bouncing.rtsc"))
    cam = lib.camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 16.0 / 9.0, 0.1, 10.0, 0.0, 1.0)
    W, H = 1200, 675
else:
    scene = abi.load_scene(os.path.join(ROOT, "tests", "golden", "scene_c2.rtsc"))
    cam = lib.camera((12, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.1, 10.0)
    W, H = 1200, 800
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev)
ctx = lib.Context(0, s.cuda_stream)
ctx.upload(scene)
buf = torch.zeros((H, W, 3), dtype=torch.int64, device=dev)
for rep in range(2):
    ctx.render_device(cam, abi.render_params(W, H, spp, depth, seed=1), buf.data_ptr())
    ctx.synchronize()
    st = ctx.take_stats()
print("kernel ms", st["kernel_ms"])
p = np.fromfile(OUT, dtype=np.uint64).reshape(-1, 16)
p = p[p[:, 7] == 1].astype(np.float64)
t0 = p[:, 0].min()
tick = 1e-5  # ms per tick at 100 MHz
start, tq, tx = (p[:, 0] - t0) * tick, (p[:, 1] - t0) * tick, (p[:, 2] - t0) * tick
print(f"waves {len(p)}  start spread {start.max():.3f} ms")
print(f"queue dry seen: first {tq[p[:,1]>0].min():.3f}  median {np.median(tq[p[:,1]>0]):.3f}  last {tq.max():.3f} ms")
print("wave exit percentiles (ms):", " ".join(f"{q}%={np.percentile(tx, q):.3f}" for q in (1, 10, 25, 50, 75, 90, 99, 100)))
tail = tx - np.median(tq[p[:, 1] > 0])
print("iterations per wave: median", np.median(p[:, 3]), " after dry: median", np.median(p[:, 4]), "max", p[:, 4].max())
print(f"lane fill overall {p[:,5].sum() / (64 * p[:,3].sum()):.3f}   after dry {p[:,6].sum() / max(1.0, 64 * p[:,4].sum()):.3f}")
print(f"wave-iterations after dry / total: {p[:,4].sum() / p[:,3].sum():.4f}")
edges = np.linspace(np.median(tq[p[:, 1] > 0]), tx.max(), 13)
alive = [(tx > e).sum() for e in edges]
print("waves alive after dry:", " ".join(f"{e - edges[0]:.2f}ms:{a}" for e, a in zip(edges, alive)))
sec = p[:, 8:15].sum(axis=0)
names = ["refill+camera", "screen dense", "screen sparse", "exact", "shade", "screen prep", "screen members"]
print("wave time by section: " + "  ".join(f"{n} {100 * v / sec.sum():.1f}%" for n, v in zip(names, sec)))
print(f"cycles per wave trip: {sec.sum() / p[:, 3].sum():.0f} (clock64 ticks), sections: " + " ".join(f"{v / p[:, 3].sum():.0f}" for v in sec))
ctx.close()
print(f"member-screen rounds per wave trip: {p[:, 15].sum() / p[:, 3].sum():.2f}")
