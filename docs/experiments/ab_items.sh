#!/bin/bash
# Kernel time of config 2 at a few sample counts against the run length of a work item (RT_ITEM_SAMPLES), with the
# tile order on and off:  tools/ab_items.sh "<spp list>" <cap>...
cd ${GRAFT_REPO_ROOT:-.}
SPPS=$1; shift
for rep in 1 2; do
  for cap in "$@"; do
    for ord in 1 0; do
      if [ "$cap" = "default" ]; then unset RT_ITEM_SAMPLES; else export RT_ITEM_SAMPLES=$cap; fi
      out=$(RT_PIXEL_ORDER=$ord timeout -k 10 120 python3 tools/tail_probe.py $SPPS 50 2>/dev/null | grep "spp" | grep kernel | sed -e 's/.*kernel *//' -e 's/ ms .*//' | tr '\n' ' ') || exit 1
      echo "cap $cap order $ord: $out"
    done
  done
done
