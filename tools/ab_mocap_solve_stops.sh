#!/bin/bash
# dev: duration of the full-frame launches of ik_solve_kernel in the capture-excerpt leg truncated at debug stops (median of the
# launches longer than 12 us: frames skipped for too few markers return at once). OVERLAP=0: nothing runs beside the solve.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for st in ${STOPS:-1 4 2 3 0}; do
  OUT=$ROOT/gpurun_out/mstop$st; rm -rf $OUT; mkdir -p $OUT
  SMPLPP_IK_OVERLAP=${OVERLAP:-1} SMPLPP_IK_DBG_STOP=$st rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/mocap_only.py ${R:-8} > $OUT/out.txt 2> $OUT/err.txt
  python3 - <<PY
import csv, glob, statistics
rows = [r for r in csv.DictReader(open(glob.glob("$OUT/*/*kernel_trace.csv")[0])) if "ik_solve_kernel" in r["Kernel_Name"]]
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows)
full = [x for x in d if x > (12.0 if $st != 1 else 0.0)]
print("== stop $st: %d launches, all: median %.1f us; the longer ones: median %.1f us (%d)" % (len(d), statistics.median(d), statistics.median(full) if full else 0, len(full)))
PY
done
