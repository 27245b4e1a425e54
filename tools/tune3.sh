#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for depth in 50 8 3; do for spp in 63 125 250; do timeout -k 10 120 python3 bench.py --spp $spp --depth $depth --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"depth $depth spp\", d[\"config\"][\"spp\"], \"ms/step\", round(d[\"ms_per_step\"],3), \"rays/sample\", round(d[\"rays_per_sample\"],3))"; done; done
