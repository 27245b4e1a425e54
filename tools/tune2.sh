#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for spp in 1 2 4 8 16 32 63 125 250; do timeout -k 10 120 python3 bench.py --spp $spp --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"spp\", d[\"config\"][\"spp\"], \"ms/step\", round(d[\"ms_per_step\"],3), \"kernel_ms\", round(d[\"roofline\"][\"kernel_ms\"],3))"; done
