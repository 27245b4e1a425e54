"""Scene-graph kernels vs the oracle on every book-2 scene (small renders) + rough timing."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import abi, lib
from tests import oracle_binding as ob
G = os.path.join(ROOT, "tests/golden")
kat = json.load(open(os.path.join(G, "kat_b2.json")))
ctx = lib.Context(0)
FR = 2.0 ** 36
for name in ["two_spheres", "cornell_box", "bouncing", "simple_light", "two_perlin", "earth", "earth_real", "cornell_smoke", "final"]:
    k = kat[name]
    path = os.path.join(G, f"scene_b2_{name}.rtsc")
    sc = abi.load_scene(path if os.path.exists(path) else path + ".gz")
    cam = lib.camera(k["lookfrom"], k["lookat"], (0, 1, 0), k["vfov"], k["aspect"], k["aperture"], 10.0, 0.0, 1.0)
    t = time.time(); ctx.upload(sc); tu = time.time() - t
    W, H, spp = 64, 48, 4
    p = abi.render_params(W, H, spp, 50, seed=3)
    g, st = ctx.render(cam, p)
    _, of, ost = ob.render(sc, cam, p, rng=ob.RNG_PHILOX, assoc=ob.ASSOC_FORWARD, has_time=True, threads=16)
    o = of / FR
    d = np.abs(g - o)
    print(f"{name:14s} upload {tu*1e3:6.1f} ms  equal {np.array_equal(g, o)}  max|d| {d.max():.3g}  n_diff {(d>0).sum()}  rays gpu {st['rays']} oracle {ost['rays']}  nan {st['nan_samples']}", flush=True)
    if "firsthit" in k:
        fw, fh = k["firsthit"]["width"], k["firsthit"]["height"]
        ids = ctx.first_hit(cam, fw, fh)
        ref = np.load(os.path.join(G, f"b2_{name}_firsthit.npz"))["ids"].astype(np.int32)
        print(f"{'':14s} first-hit equal {np.array_equal(ids, ref)}  mismatches {(ids != ref).sum()}", flush=True)
# timing at larger size
for name, W, H, spp in (("bouncing", 1200, 675, 16), ("final", 800, 800, 16), ("cornell_box", 600, 600, 16)):
    k = kat[name]
    sc = abi.load_scene(os.path.join(G, f"scene_b2_{name}.rtsc"))
    cam = lib.camera(k["lookfrom"], k["lookat"], (0, 1, 0), k["vfov"], k["aspect"], k["aperture"], 10.0, 0.0, 1.0)
    ctx.upload(sc)
    g, st = ctx.render(cam, abi.render_params(W, H, spp, 50, seed=1))
    print(f"{name:14s} {W}x{H}x{spp}: kernel {st['kernel_ms']:.1f} ms  {W*H*spp/st['kernel_ms']/1e3:.1f} Msamples/s  rays/sample {st['rays']/st['samples']:.2f}", flush=True)
# the sphere-list scene through the graph kernels
c2 = abi.load_scene(os.path.join(G, "scene_c2.rtsc"))
cam2 = lib.camera((12, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.1, 10.0)
p = abi.render_params(96, 64, 4, 50, seed=1)
ctx.upload(c2); a, _ = ctx.render(cam2, p)
ctx.set_force_graph(True); ctx.upload(c2); b, st = ctx.render(cam2, p); ctx.set_force_graph(False)
print("config-2 scene: graph kernels == sphere-list kernels:", np.array_equal(a, b), st)
