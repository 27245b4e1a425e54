#!/bin/bash
# Same-box comparison of sphere-kernel builds on config 2 (500 spp) and the bouncing-spheres world (100 spp):
#   tools/ab_spheres.sh <lib>...   (paths relative to the repo root)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for l in "$@"; do
    c2=$(RT_LIB_PATH=$PWD/$l timeout -k 10 120 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-pipelined 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],2))") || exit 1
    bo=$(RT_LIB_PATH=$PWD/$l timeout -k 10 120 python3 tools/bench_scenes.py 100 bouncing 2>/dev/null | grep -i bouncing | tr -s ' ' | cut -c1-110) || exit 1
    echo "$l pass $rep: c2 ${c2} ms | $bo"
  done
done
