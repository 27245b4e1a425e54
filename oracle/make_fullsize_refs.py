#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Full-size statistical references from the UNMODIFIED reference program.

The reference author's own recipe for using several cores ("Ray Tracing In a Weekend/README.md":2538-2542: run the
program N times with different seeds and average the images) applied to oracle/_ref/: P processes, each with its own
srand seed after the scene build, S samples per pixel each, at the FULL image size of a benchmark configuration.  The
mean radiance is stored as float16 (rounding 2^-11 relative, a twentieth of the image's own noise) together with the
measured noise: the RMSE between the two halves of the seeds, from which the per-sample deviation sigma follows
(rmse_halves = sigma * sqrt(2 / (P*S/2))).

    python oracle/make_fullsize_refs.py [c2] [c3] [c4]

Outputs tests/golden/<name>_full_ref.npz {radiance float16 [H,W,3], spp, seeds, sigma, rmse_halves, image_mean}.
Runs only where /root/reference exists (the development container).
"""
import json
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF = {"b1": os.path.join(ROOT, "oracle", "_ref", "ref_b1"), "b2": os.path.join(ROOT, "oracle", "_ref", "ref_b2")}

# name -> (harness, scene, W, H, processes, spp per process)
JOBS = {
    "c2": ("b1", "c2", 1200, 800, 8, 32),
    "c3": ("b2", "bouncing", 1200, 675, 8, 32),
    "c4": ("b2", "final", 800, 800, 8, 32),
}


def one(job, seed, path):
    book, scene, W, H, _, spp = job
    subprocess.run([REF[book], "render", scene, str(W), str(H), str(spp), "50", str(seed), path], check=True,
                   capture_output=True)


def make(name, tmp):
    job = JOBS[name]
    _, _, W, H, P, spp = job
    seeds = [1000 + 17 * k for k in range(P)]
    paths = [os.path.join(tmp, f"{name}_{s}.f64") for s in seeds]
    with ThreadPoolExecutor(P) as ex:
        list(ex.map(lambda a: one(job, a[0], a[1]), zip(seeds, paths)))
    imgs = np.stack([np.fromfile(p).reshape(H, W, 3) / spp for p in paths])
    for p in paths:
        os.remove(p)
    a, b = imgs[: P // 2].mean(axis=0), imgs[P // 2:].mean(axis=0)
    rmse_halves = float(np.sqrt(np.mean((a - b) ** 2)))
    sigma = rmse_halves / np.sqrt(2.0 / (P * spp / 2))
    mean = imgs.mean(axis=0)
    out = os.path.join(GOLD, f"{name}_full_ref.npz")
    np.savez_compressed(out, radiance=mean.astype(np.float16), spp=np.int32(P * spp), seeds=np.array(seeds),
                        sigma=np.float64(sigma), rmse_halves=np.float64(rmse_halves),
                        image_mean=mean.reshape(-1, 3).mean(axis=0))
    print(json.dumps({"name": name, "spp": P * spp, "sigma": sigma, "rmse_halves": rmse_halves,
                      "image_mean": [float(x) for x in mean.reshape(-1, 3).mean(axis=0)],
                      "bytes": os.path.getsize(out)}), flush=True)


def main():
    if not os.path.isdir("/root/reference"):
        sys.exit("the reference tree is not present: references can only be regenerated in the development container")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    names = sys.argv[1:] or list(JOBS)
    with tempfile.TemporaryDirectory() as tmp:
        for n in names:
            make(n, tmp)


if __name__ == "__main__":
    main()
