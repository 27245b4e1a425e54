#!/usr/bin/env python3
"""Back-to-back timing of k_render_spheres under two environment settings on the same GPU.
usage: ab_env.py NAME=VALUE_A NAME=VALUE_B [spp list]"""
import os, subprocess, sys
a, b = sys.argv[1], sys.argv[2]
spps = sys.argv[3] if len(sys.argv) > 3 else "63,500"
for rep in range(3):
    for setting in (a, b):
        k, v = setting.split("=", 1)
        env = dict(os.environ, **{k: v})
        out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "tail_probe.py"), spps, "50"], env=env, capture_output=True, text=True).stdout
        print(setting, " | ".join(l.split("kernel")[1].split("ms")[0].strip() for l in out.splitlines() if "spp" in l and "kernel" in l), flush=True)
