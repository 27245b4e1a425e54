import os, sys, subprocess
for rep in range(3):
    for name, lib in (("old", "tools/libold.so"), ("new", "raytracing_amd/libraytracing_amd.so")):
        env = dict(os.environ, RT_LIB_PATH=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "tools/tail_probe.py", "63,500", "50"], env=env, capture_output=True, text=True).stdout
        print(name, " | ".join(l.split("kernel")[1].split("ms")[0].strip() for l in out.splitlines() if "spp" in l and "kernel" in l), flush=True)
