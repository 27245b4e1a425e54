#!/bin/bash
# Kernel time of config 2 at a few sample counts against the run length of a work item (RT_ITEM_SAMPLES):
#   tools/ab_cap.sh "<spp list>" <cap>...      ("default": the library's own choice)
cd ${GRAFT_REPO_ROOT:-.}
SPPS=$1; shift
for rep in 1 2 3; do
  for cap in "$@"; do
    if [ "$cap" = "default" ]; then unset RT_ITEM_SAMPLES; else export RT_ITEM_SAMPLES=$cap; fi
    out=$(timeout -k 10 120 python3 tools/tail_probe.py $SPPS 50 2>/dev/null | grep "spp" | grep kernel | sed -e 's/.*kernel *//' -e 's/ ms .*//' | tr '\n' ' ') || exit 1
    echo "cap $cap: $out"
  done
done
