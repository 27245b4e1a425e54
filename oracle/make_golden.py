#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Regenerates tests/golden/ from the UNMODIFIED reference program.

Runs only where /root/reference exists (the development container): builds oracle/_ref/
(the reference sources compiled where they lie, driven by our harness main()) and stores
what the tests need as data -- scenes, hashes, small images.  Nothing of the reference's
source text is stored.

    python oracle/make_golden.py [--skip-slow]

Book 1 outputs (tests/golden/):
  scene_c1.rtsc, scene_c2.rtsc   the config-1 / config-2 worlds as RTSCENE1 files
  kat_b1.json                    hashes and facts (see keys below)
  c2_120x80x10_sums.npy          double colour sums of a reduced config-2 render
  c2_firsthit_1200x800.npz       pixel-centre closest-hit ids of config 2 (int16, compressed)
  c1_400x225x100_ref.npz         8-bit image + float32 radiance of the full config-1 render
  c2_300x200_ref.npz             float32 mean radiance of 4 x 500 spp config-2 renders, and the
                                 measured reference-vs-reference noise (for the RMSE gate)
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")
GOLD = os.path.join(ROOT, "tests", "golden")
REF_B1 = os.path.join(ORACLE, "_ref", "ref_b1")
REF_B2 = os.path.join(ORACLE, "_ref", "ref_b2")


def run(*cmd):
    return subprocess.run(list(cmd), check=True, capture_output=True, text=True).stdout


def sha(path_or_bytes):
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        return hashlib.sha256(path_or_bytes).hexdigest()
    with open(path_or_bytes, "rb") as fp:
        return hashlib.sha256(fp.read()).hexdigest()


def book1(tmp, skip_slow):
    kat = {}
    run(REF_B1, "scene", "c1", os.path.join(GOLD, "scene_c1.rtsc"))
    run(REF_B1, "scene", "c2", os.path.join(GOLD, "scene_c2.rtsc"))
    kat["scene_c1_sha256"] = sha(os.path.join(GOLD, "scene_c1.rtsc"))
    kat["scene_c2_sha256"] = sha(os.path.join(GOLD, "scene_c2.rtsc"))

    # facts about the rolled config-2 scene; number of rand() calls random_scene() makes
    kat["c2_scene"] = json.loads(run(REF_B1, "kat3"))

    # config 1, full size, fresh rand() stream: the reference's own P3 output
    f64, ppm = os.path.join(tmp, "c1.f64"), os.path.join(tmp, "c1.ppm")
    out = json.loads(run(REF_B1, "render", "c1", "400", "225", "100", "50", "-", f64, ppm))
    sums = np.fromfile(f64).reshape(225, 400, 3)
    kat["c1_400x225x100"] = {"ppm_sha256": sha(ppm), "sums_sha256": sha(sums.tobytes()),
                             "next_rand": out["next_rand"]}
    rgb = np.loadtxt(ppm, skiprows=3, dtype=np.int64).astype(np.uint8).reshape(225, 400, 3)
    np.savez_compressed(os.path.join(GOLD, "c1_400x225x100_ref.npz"), rgb=rgb,
                        radiance=(sums / 100.0).astype(np.float32))

    # config 2, reduced, continuing the stream after the scene build like the reference does
    f64 = os.path.join(tmp, "c2s.f64")
    out = json.loads(run(REF_B1, "render", "c2", "120", "80", "10", "50", "-", f64))
    sums = np.fromfile(f64).reshape(80, 120, 3)
    np.save(os.path.join(GOLD, "c2_120x80x10_sums.npy"), sums)
    kat["c2_120x80x10"] = {"sums_sha256": sha(sums.tobytes()), "next_rand": out["next_rand"]}

    # config 2 first-hit ids at full size
    i32 = os.path.join(tmp, "ids.i32")
    run(REF_B1, "firsthit", "c2", "1200", "800", i32)
    ids = np.fromfile(i32, dtype=np.int32).reshape(800, 1200)
    kat["c2_firsthit_1200x800"] = {"sha256": sha(ids.tobytes()), "miss_pixels": int((ids < 0).sum()),
                                   "ground_pixels": int((ids == 0).sum())}
    np.savez_compressed(os.path.join(GOLD, "c2_firsthit_1200x800.npz"), ids=ids.astype(np.int16))
    run(REF_B1, "firsthit", "c1", "400", "225", i32)
    ids = np.fromfile(i32, dtype=np.int32).reshape(225, 400)
    kat["c1_firsthit_400x225"] = {"sha256": sha(ids.tobytes())}

    if not skip_slow:
        # 4 independent 500-spp reference renders of config 2 at 300x200 (about 5 minutes each)
        seeds = [11, 12, 13, 14]
        paths = [os.path.join(tmp, f"c2_{s}.f64") for s in seeds]
        with ThreadPoolExecutor(4) as ex:
            list(ex.map(lambda a: run(REF_B1, "render", "c2", "300", "200", "500", "50", str(a[0]), a[1]),
                        zip(seeds, paths)))
        imgs = np.stack([np.fromfile(p).reshape(200, 300, 3) / 500.0 for p in paths])
        pair_rmse = [float(np.sqrt(np.mean((imgs[a] - imgs[b]) ** 2))) for a in range(4) for b in range(a + 1, 4)]
        mean = imgs.mean(axis=0)
        np.savez_compressed(os.path.join(GOLD, "c2_300x200_ref.npz"), radiance=mean.astype(np.float32),
                            one_seed=imgs[0].astype(np.float32))
        kat["c2_300x200"] = {"spp_each": 500, "seeds": seeds, "pair_rmse_500spp": pair_rmse,
                             "image_mean_rgb": [float(x) for x in mean.reshape(-1, 3).mean(axis=0)],
                             "per_seed_image_mean_rgb": [[float(x) for x in im.reshape(-1, 3).mean(axis=0)]
                                                         for im in imgs]}
    else:
        old = os.path.join(GOLD, "kat_b1.json")
        if os.path.exists(old):
            with open(old) as fp:
                prev = json.load(fp)
            if "c2_300x200" in prev:
                kat["c2_300x200"] = prev["c2_300x200"]

    with open(os.path.join(GOLD, "kat_b1.json"), "w") as fp:
        json.dump(kat, fp, indent=1, sort_keys=True)
    print(json.dumps(kat, indent=1, sort_keys=True))


B2_SCENES = ["bouncing", "two_spheres", "two_perlin", "earth", "earth_real", "simple_light", "cornell_box",
             "cornell_smoke", "final"]
# scenes with a constant_medium draw random numbers while intersecting: their first-hit ids are not a function
# of the ray alone, so no id fixture is made for them
B2_FIRSTHIT = ["bouncing", "two_spheres", "two_perlin", "earth_real", "simple_light", "cornell_box"]
# statistical references: (scene, W, H, spp per seed)
B2_IMAGES = [("bouncing", 400, 225, 256), ("cornell_box", 200, 200, 256), ("cornell_smoke", 200, 200, 256),
             ("final", 200, 200, 256), ("two_perlin", 200, 112, 128), ("simple_light", 200, 112, 256)]


def book2(tmp, skip_slow):
    """Book 2/3 outputs: scene_b2_<name>.rtsc(.gz), kat_b2.json (camera + stream facts, hashes of a 64x48x6
    render continuing the rand() stream after the scene build, first-hit id hashes), b2_<name>_64x48x6_sums.npy,
    b2_<name>_firsthit.npz and b2_<name>_ref.npz (mean of 4 seeds + noise figures)."""
    import gzip
    kat = {}
    for name in B2_SCENES:
        k = json.loads(run(REF_B2, "info", name))
        path = os.path.join(GOLD, f"scene_b2_{name}.rtsc")
        run(REF_B2, "scene", name, path)
        k["scene_sha256"] = sha(path)
        if os.path.getsize(path) > 1_000_000:  # the decoded earth map: keep it compressed
            with open(path, "rb") as fp, gzip.GzipFile(path + ".gz", "wb", mtime=0) as gz:
                gz.write(fp.read())
            os.remove(path)
        f64 = os.path.join(tmp, "r.f64")
        out = json.loads(run(REF_B2, "render", name, "64", "48", "6", "50", "-", f64))
        sums = np.fromfile(f64).reshape(48, 64, 3)
        np.save(os.path.join(GOLD, f"b2_{name}_64x48x6_sums.npy"), sums)
        k["render_64x48x6"] = {"sums_sha256": sha(sums.tobytes()), "next_rand": out["next_rand"]}
        if name in B2_FIRSTHIT:
            W = 240
            H = int(W / k["aspect"])
            i32 = os.path.join(tmp, "ids.i32")
            run(REF_B2, "firsthit", name, str(W), str(H), i32)
            ids = np.fromfile(i32, dtype=np.int32).reshape(H, W)
            np.savez_compressed(os.path.join(GOLD, f"b2_{name}_firsthit.npz"), ids=ids.astype(np.int16))
            k["firsthit"] = {"width": W, "height": H, "sha256": sha(ids.tobytes())}
        kat[name] = k
    old = os.path.join(GOLD, "kat_b2.json")
    prev = {}
    if os.path.exists(old):
        with open(old) as fp:
            prev = json.load(fp)
    for name, W, H, spp in B2_IMAGES:
        if skip_slow:
            if "image" in prev.get(name, {}):
                kat[name]["image"] = prev[name]["image"]
            continue
        seeds = [21, 22, 23, 24]
        paths = [os.path.join(tmp, f"{name}_{s}.f64") for s in seeds]
        with ThreadPoolExecutor(4) as ex:
            list(ex.map(lambda a: run(REF_B2, "render", name, str(W), str(H), str(spp), "50", str(a[0]), a[1]),
                        zip(seeds, paths)))
        imgs = np.stack([np.fromfile(p).reshape(H, W, 3) / spp for p in paths])
        pair = [float(np.sqrt(np.mean((imgs[a] - imgs[b]) ** 2))) for a in range(4) for b in range(a + 1, 4)]
        mean = imgs.mean(axis=0)
        np.savez_compressed(os.path.join(GOLD, f"b2_{name}_ref.npz"), radiance=mean.astype(np.float32),
                            one_seed=imgs[0].astype(np.float32))
        kat[name]["image"] = {"width": W, "height": H, "spp_each": spp, "seeds": seeds, "pair_rmse": pair,
                              "image_mean_rgb": [float(x) for x in mean.reshape(-1, 3).mean(axis=0)]}
    with open(os.path.join(GOLD, "kat_b2.json"), "w") as fp:
        json.dump(kat, fp, indent=1, sort_keys=True)
    print(json.dumps({k: v.get("image", {}).get("pair_rmse") for k, v in kat.items()}, indent=1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-slow", action="store_true", help="keep the existing statistical reference images")
    ap.add_argument("--book", choices=["1", "2", "all"], default="all")
    args = ap.parse_args()
    if not os.path.isdir("/root/reference"):
        sys.exit("the reference tree is not present: golden data can only be regenerated in the development container")
    subprocess.run(["make", "-C", ORACLE, "ref"], check=True)
    os.makedirs(GOLD, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        if args.book in ("1", "all"):
            book1(tmp, args.skip_slow)
        if args.book in ("2", "all"):
            book2(tmp, args.skip_slow)


if __name__ == "__main__":
    main()
