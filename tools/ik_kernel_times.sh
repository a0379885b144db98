#!/bin/bash
# Per-kernel average durations of the IK loop of bench.py (rocprofv3 --kernel-trace --stats). usage (GPU box, repo root): bash tools/ik_kernel_times.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/iktrace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("%-56s calls %5s avg %9.2f us %6s%%" % (r["Name"][:56], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
