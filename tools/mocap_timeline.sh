#!/bin/bash
# Start/end timeline of the kernels of a few frames of the capture-fitting loop (rocprofv3 --kernel-trace), R restarts.
# usage (GPU box, repo root): bash tools/mocap_timeline.sh [R]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/mocaptl; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/mocap_only.py ${1:-64} > $OUT/out.txt 2> $OUT/err.txt
cat $OUT/out.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-40:-12]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-44s start %9.2f us  end %9.2f us  dur %7.2f" % (r["Kernel_Name"][:44], s / 1e3, e / 1e3, (e - s) / 1e3))
PY
