#!/bin/bash
# Same-box comparison of two library builds on config 2 and every book-2 world: tools/ab_all.sh <lib A> <lib B>
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for l in "$@"; do
    c2=$(RT_LIB_PATH=$PWD/$l timeout -k 10 120 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-pipelined 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],2))") || exit 1
    echo "== $l pass $rep: c2 ${c2} ms"
    RT_LIB_PATH=$PWD/$l timeout -k 10 200 python3 tools/bench_scenes.py 64 2>/dev/null | cut -c1-75 || exit 1
    RT_LIB_PATH=$PWD/$l timeout -k 10 200 python3 tools/gpu_probe_b3.py 2>/dev/null | grep "600x600x32" | cut -c1-120
  done
done
