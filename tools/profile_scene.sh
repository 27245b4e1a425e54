#!/bin/bash
# Counter passes for the scene-graph render kernel on one book-2 world.
# Usage: tools/profile_scene.sh <tag> <scene> [spp]    results: gpurun_out/prof_<tag>/...
set -u
TAG=${1:-scene}; SCENE=${2:-final}; SPP=${3:-8}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- \
    python3 "$ROOT/tools/bench_scenes.py" $SPP $SCENE > "$OUT/trace_bench.txt" 2> "$OUT/trace.err" || { echo "trace pass failed"; tail -5 "$OUT/trace.err"; exit 1; }
pass() {
    local name=$1; shift
    timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o bench -- \
        python3 "$ROOT/tools/bench_scenes.py" $SPP $SCENE > "$OUT/${name}_bench.txt" 2> "$OUT/$name.err" \
        || { echo "counter pass $name failed"; tail -5 "$OUT/$name.err"; exit 1; }
}
pass pmcA SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE GRBM_COUNT
pass pmcB SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS
pass pmcD FETCH_SIZE
pass pmcE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass pmcF TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum
echo "profile passes done: $OUT"
