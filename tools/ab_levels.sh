#!/bin/bash
# Kernel time of config 2 at a few sample counts against the number of runs a pixel's samples are cut into
# (RT_ITEM_LEVELS; "default": the library's own choice):   tools/ab_levels.sh "<spp list>" <levels>...
cd ${GRAFT_REPO_ROOT:-.}
SPPS=$1; shift
for rep in 1 2; do
  for lv in "$@"; do
    if [ "$lv" = "default" ]; then unset RT_ITEM_LEVELS; else export RT_ITEM_LEVELS=$lv; fi
    out=$(timeout -k 10 120 python3 tools/tail_probe.py $SPPS 50 2>/dev/null | grep "spp" | grep kernel | sed -e 's/.*kernel *//' -e 's/ ms .*//' | tr '\n' ' ') || exit 1
    echo "runs per pixel $lv: $out"
  done
done
