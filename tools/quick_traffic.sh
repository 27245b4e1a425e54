#!/bin/bash
# One rocprofv3 counter pass over a single bench.py frame: memory-side write bytes of the render kernel.
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/qt_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT" -o bench -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline ${@:2} > "$OUT/bench.json" 2> "$OUT/err.txt" || { echo "counter pass failed"; tail -5 "$OUT/err.txt"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, sys, collections, glob
per = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/bench_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render_spheres" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(per.items()):
    print(f"{k:20s} {sum(v) / len(v):.5g}")
PY
