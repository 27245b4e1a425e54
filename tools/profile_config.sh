#!/bin/bash
# Kernel trace and counter passes of ONE BASELINE configuration's render kernel (tools/bench_config.py under rocprofv3):
#   tools/profile_config.sh <tag> <c1|c2|c3|c4|c5> [spp]      results: gpurun_out/prof_<tag>/...
# then, in the repository:  python3 tools/summarize_profile.py <tag> <kernel> config_<name>
# One rocprofv3 run per pass; a counter pass never shares a run with a trace domain other than the kernel trace, and
# stays inside the per-block counter limits (8 SQ, 4 TCC with FETCH_SIZE = 3 and WRITE_SIZE = 2, 4 TCP, 2 GRBM).
set -u
TAG=${1:?tag}; NAME=${2:?config}; SPP=${3:--}   # "-": the configuration's own sample count
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- \
    python3 "$ROOT/tools/bench_config.py" $NAME $SPP 2 > "$OUT/trace_bench.txt" 2> "$OUT/trace.err" || { echo "trace pass failed"; tail -5 "$OUT/trace.err"; exit 1; }
pass() {
    local name=$1; shift
    timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o bench -- \
        python3 "$ROOT/tools/bench_config.py" $NAME $SPP 1 > "$OUT/${name}_bench.txt" 2> "$OUT/$name.err" \
        || { echo "counter pass $name failed"; tail -5 "$OUT/$name.err"; exit 1; }
}
pass pmcA SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE GRBM_COUNT
pass pmcB SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS
pass pmcC SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64
pass pmcF SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64
pass pmcG SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_FLOPS_FP64_TRANS
pass pmcH TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD
pass pmcD FETCH_SIZE
pass pmcE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
echo "profile passes done: $OUT"
