"""First contact with the GPU: parity of the sphere kernels vs the oracle + a rough timing."""
import hashlib, math, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import abi, lib
from tests import oracle_binding as ob

c1 = abi.load_scene(os.path.join(ROOT, "tests/golden/scene_c1.rtsc"))
c2 = abi.load_scene(os.path.join(ROOT, "tests/golden/scene_c2.rtsc"))
lf, la = (3, 3, 2), (0, 0, -1)
cam1 = lib.camera(lf, la, (0, 1, 0), 20.0, 16.0 / 9.0, 2.0, math.sqrt(sum((a - b) ** 2 for a, b in zip(lf, la))))
cam2 = lib.camera((12, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.1, 10.0)

ctx = lib.Context(0)
print("devices", lib.load().rt_get_device_count(), flush=True)

# first hit, config 2, full size
ctx.upload(c2)
t = time.time(); ids = ctx.first_hit(cam2, 1200, 800); print("first_hit s", time.time() - t)
print("first-hit sha", hashlib.sha256(ids.tobytes()).hexdigest(), flush=True)

# small render parity vs oracle (philox, forward)
for name, sc, cam, W, H, spp in (("c1", c1, cam1, 80, 45, 8), ("c2", c2, cam2, 60, 40, 4)):
    ctx.upload(sc)
    p = abi.render_params(W, H, spp, 50, seed=1)
    g, st = ctx.render(cam, p)
    o, of, ost = ob.render(sc, cam, p, rng=ob.RNG_PHILOX, assoc=ob.ASSOC_FORWARD, threads=8)
    ofd = of.astype(np.float64) / 2.0 ** 36
    print(name, "gpu stats", st, "oracle", ost)
    print(name, "max|gpu-oracle_fixed|", np.abs(g - ofd).max(), "equal", np.array_equal(g, ofd), flush=True)
    ctx.set_exact_only(True)
    g2, st2 = ctx.render(cam, p)
    ctx.set_exact_only(False)
    print(name, "exact-only equal", np.array_equal(g, g2), st2, flush=True)

# timing
ctx.upload(c2)
for spp in (4, 16, 62):
    p = abi.render_params(1200, 800, spp, 50, seed=1)
    t = time.time(); g, st = ctx.render(cam2, p); dt = time.time() - t
    print("c2 1200x800 spp", spp, "wall", dt, "kernel_ms", st["kernel_ms"], "Msamples/s(kernel)",
          1200 * 800 * spp / st["kernel_ms"] / 1e3, "rays/sample", st["rays"] / st["samples"], st, flush=True)
