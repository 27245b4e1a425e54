"""Kernel time of the Rest-Of-Your-Life Cornell box with light sampling at 600x600 and a few sample counts (used by
tools/ab_all.sh)."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from raytracing_amd import abi, lib
from tests import oracle_binding as ob
s = abi.load_scene('tests/golden/scene_b3_cornell.rtsc')
cam = lib.camera((278,278,-800),(278,278,0),(0,1,0),40.0,1.0,0.0,10.0,0.0,1.0)
ctx = lib.Context(0); ctx.upload(s)
FR = 2.0**36
for flags in (abi.RT_RENDER_IMPORTANCE, 0):
    p = abi.render_params(64,64,6,50,seed=3,flags=flags)
    g, st = ctx.render(cam, p)
    _, of, ost = ob.render(s, cam, p, rng=ob.RNG_PHILOX, assoc=ob.ASSOC_FORWARD, has_time=True, threads=16)
    o = of/FR; d = np.abs(g-o)
    print("flags", flags, "equal", np.array_equal(g,o), "max|d|", d.max(), "n_diff", (d>0).sum(), "rays", st['rays'], ost['rays'], "nan", st['nan_samples'], ost['nan_samples'], flush=True)
ids = ctx.first_hit(cam, 120, 120); ref = ob.first_hit(s, cam, 120, 120)
print("first hit equal", np.array_equal(ids, ref))
for flags in (abi.RT_RENDER_IMPORTANCE, 0):
    g, st = ctx.render(cam, abi.render_params(600,600,32,50,seed=1,flags=flags))
    print("flags", flags, "600x600x32 kernel ms", st['kernel_ms'], "Msamples/s", 600*600*32/st['kernel_ms']/1e3, "rays/sample", st['rays']/st['samples'], "nan", st['nan_samples'], "mean", g.mean(axis=(0,1))/32)
