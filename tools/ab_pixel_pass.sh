#!/bin/bash
# Kernel time of config 2 with the frame-by-frame and the pixel-by-pixel item plans (RT_PIXEL_PASS=0 / 1) at low sample
# counts: where the pixel-by-pixel plan starts to pay.   tools/ab_pixel_pass.sh "<spp list>"
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for pp in 0 1; do
    out=$(RT_PIXEL_PASS=$pp timeout -k 10 120 python3 tools/tail_probe.py $1 50 2>/dev/null | grep "spp" | grep kernel | sed -e 's/.*kernel *//' -e 's/ ms .*//' | tr '\n' ' ') || exit 1
    echo "pixel pass $pp: $out"
  done
done
