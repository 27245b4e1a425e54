#!/bin/bash
# One rocprofv3 counter pass over a single bench.py frame: VALU instruction count, lane use, waiting.
# Usage (on the GPU box): tools/quick_counters.sh <tag>   -> gpurun_out/qc_<tag>.txt
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/qc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT" -o bench -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/err.txt" || { echo "counter pass failed"; tail -5 "$OUT/err.txt"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, sys, collections, glob
per = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/bench_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render_spheres" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in per.items()}
for k, v in sorted(c.items()):
    print(f"{k:28s} {v:.5g}")
cyc = c["GRBM_GUI_ACTIVE"] / 8
print("cycles/VALU instr/SIMD", 1024 * cyc / c["SQ_INSTS_VALU"], "lane use", c["SQ_THREAD_CYCLES_VALU"] / 64 / c["SQ_ACTIVE_INST_VALU"],
      "wait share", c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"])
PY
