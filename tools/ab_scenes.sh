#!/bin/bash
# Same-box comparison of scene-kernel builds: [NAMES=a,b] tools/ab_scenes.sh <spp> <lib>...   (paths relative to the repo root)
spp=$1; shift
for rep in 1 2; do
  for l in "$@"; do
    echo "== $l (pass $rep)"
    RT_LIB_PATH=$PWD/$l timeout -k 10 120 python tools/bench_scenes.py $spp $NAMES || exit 1
  done
done
