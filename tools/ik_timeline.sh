#!/bin/bash
# Start/end timeline of the kernels of a few IK iterations (rocprofv3 --kernel-trace): shows which kernels overlap.
# usage (GPU box, repo root): bash tools/ik_timeline.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/iktl; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/quick_ik.py > $OUT/out.txt 2> $OUT/err.txt
cat $OUT/out.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 18 kernels of the IK leg (three iterations)
ik = [r for r in rows if "ik_" in r["Kernel_Name"] or "proj_" in r["Kernel_Name"] or "pose_kernel" in r["Kernel_Name"] or "skin_kernel" in r["Kernel_Name"]]
tail = ik[-20:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-40s start %9.2f us  end %9.2f us  dur %7.2f" % (r["Kernel_Name"][:40], s / 1e3, e / 1e3, (e - s) / 1e3))
PY
