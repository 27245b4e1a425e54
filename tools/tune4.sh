#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for fif in 1 2 3; do for spp in 63 500; do timeout -k 10 120 python3 bench.py --spp $spp --frames-in-flight $fif --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"in-flight $fif spp\", d[\"config\"][\"spp\"], \"ms/step\", round(d[\"ms_per_step\"],3), \"Msamples/s\", round(d[\"value\"],1), \"kernel_ms\", round(d[\"roofline\"][\"kernel_ms\"],2), \"frac\", round(d[\"roofline\"][\"frac\"],3))"; done; done
