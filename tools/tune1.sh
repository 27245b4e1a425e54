#!/bin/bash
# occupancy / work-item-size sweep of the render kernel (bench.py, 2 steps each)
cd ${GRAFT_REPO_ROOT:-.}
run() { echo "== $*"; env "$@" timeout -k 10 120 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1),'Msamples/s  frac',round(d['roofline']['frac'],4))"; }
timeout -k 10 120 tools/ubench/screen_rate
run RT_BLOCKS_PER_CU=4
run RT_BLOCKS_PER_CU=3
run RT_BLOCKS_PER_CU=2
run RT_BLOCKS_PER_CU=1
run RT_LIB_PATH=$PWD/raytracing_amd/libraytracing_amd_w5.so RT_BLOCKS_PER_CU=5
run RT_LIB_PATH=$PWD/raytracing_amd/libraytracing_amd_w6.so RT_BLOCKS_PER_CU=6
run RT_LIB_PATH=$PWD/raytracing_amd/libraytracing_amd_w8.so RT_BLOCKS_PER_CU=8
run RT_ITEM_SAMPLES=4
run RT_ITEM_SAMPLES=8
run RT_ITEM_SAMPLES=32
run RT_ITEM_SAMPLES=125
