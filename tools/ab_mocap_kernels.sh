#!/bin/bash
# dev: median durations of the full-frame ik_solve_kernel / ik_eval_kernel launches of the capture-excerpt leg for a list of
# library variants (ab/<name>.so; "default" = the in-tree library), two rounds.   usage: bash tools/ab_mocap_kernels.sh R name...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; R=$1; shift
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do for v in "$@"; do
  lib=""; [ "$v" != default ] && lib=$ROOT/ab/$v.so
  OUT=$ROOT/gpurun_out/abmk; rm -rf $OUT; mkdir -p $OUT
  SMPLPP_HIP_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/mocap_only.py $R > $OUT/out.txt 2> $OUT/err.txt
  python3 - <<PY
import csv, glob, statistics
rows = [r for r in csv.DictReader(open(glob.glob("$OUT/*/*kernel_trace.csv")[0]))]
out = []
for name in ("ik_solve_kernel", "ik_eval_kernel"):
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"])
    full = [x for x in d if x > 12.0]
    out.append("%s %.1f us" % (name[3:8], statistics.median(full)))
print("%-10s %s" % ("$v", "   ".join(out)))
PY
done; done
