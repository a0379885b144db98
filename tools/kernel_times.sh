#!/bin/bash
# Per-kernel average durations of the FK step (rocprofv3 kernel trace). usage (GPU box, repo root): bash tools/kernel_times.sh [frames]
N=${1:-1024}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/ktimes; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/p -- python3 $ROOT/tools/quick_fk_bench.py $N 200 > $OUT/p.log 2>&1
python3 - <<PY
import csv, glob, collections
d = collections.defaultdict(list)
rows = list(csv.DictReader(open(glob.glob("$OUT/p/*/*kernel_trace.csv")[0])))
for r in rows: d[r["Kernel_Name"][:44]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = v[len(v) // 5:]
    print("%-46s n=%3d avg %8.2f us  min %8.2f" % (k, len(v), sum(v) / len(v), min(v)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows[-60:], rows[-59:])]
print("gap between consecutive kernels (last 60): avg %.2f us" % (sum(gaps) / len(gaps)))
PY
