#!/bin/bash
# Per-kernel durations of the configs[3] leg (capture excerpt, 41 markers, QP). usage (GPU box, repo root): bash tools/mocap_kernel_times.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/mocaptrace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --ik-iters 2 > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print("%-56s calls %5s avg %9.2f us total %9.2f ms %6s%%" % (r["Name"][:56], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
