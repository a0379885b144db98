#!/bin/bash
# Start/end timeline of the kernels of the last iterations of the configs[4] workload (VPoser-latent IK, 512 frames).
# usage (GPU box, repo root): bash tools/vposer_timeline.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/vptl; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/quick_vposer_ik.py 512 12 > $OUT/out.txt 2> $OUT/err.txt
cat $OUT/out.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-40:-8]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-44s start %9.2f us  end %9.2f us  dur %7.2f" % (r["Kernel_Name"][:44], s / 1e3, e / 1e3, (e - s) / 1e3))
PY
