#!/bin/bash
# dev: median duration of the full-frame ik_solve_kernel launches of the capture-excerpt leg, default library vs a variant
#   usage: bash tools/ab_mocap_solve.sh <variant.so> [R]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; V=$1; R=${2:-64}
cd /tmp && export TMPDIR=/tmp
for lib in "" "$ROOT/ab/$V" "" "$ROOT/ab/$V"; do
  OUT=$ROOT/gpurun_out/abms; rm -rf $OUT; mkdir -p $OUT
  SMPLPP_HIP_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/mocap_only.py $R > $OUT/out.txt 2> $OUT/err.txt
  python3 - <<PY
import csv, glob, statistics
rows = [r for r in csv.DictReader(open(glob.glob("$OUT/*/*kernel_trace.csv")[0]))]
for name in ("ik_solve_kernel", "ik_eval_kernel"):
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"])
    full = [x for x in d if x > 12.0]
    print("%-10s %-16s median of the full-frame launches %.1f us (%d)" % ("${lib##*/}" or "default", name, statistics.median(full), len(full)))
PY
done
