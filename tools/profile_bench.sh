#!/bin/bash
# Profiles bench.py on the GPU box: one kernel-trace pass, then separate counter passes
# (counter passes never share a run with each other's blocks beyond the slot limits, and never
# with trace domains other than the kernel trace).  Usage: tools/profile_bench.sh <tag> [bench args]
# Results: gpurun_out/prof_<tag>/{trace,pmcA..pmcE}/...
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-pipelined $*"

timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- \
    python3 "$ROOT/bench.py" $ARGS > "$OUT/trace_bench.json" 2> "$OUT/trace.err" || { echo "trace pass failed"; exit 1; }

pass() {  # name, counters...
    local name=$1; shift
    timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o bench -- \
        python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-configs --no-pipelined > "$OUT/${name}_bench.json" 2> "$OUT/$name.err" \
        || { echo "counter pass $name failed"; tail -5 "$OUT/$name.err"; exit 1; }
}
pass pmcA SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE GRBM_COUNT
pass pmcB SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64
pass pmcC SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH
pass pmcF SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FLOPS_FP32
pass pmcG SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_FLOPS_FP64_TRANS
pass pmcH TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD
pass pmcD FETCH_SIZE
pass pmcE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
echo "profile passes done: $OUT"
