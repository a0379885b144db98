#!/bin/bash
# Development aid: duration of ik_eval_kernel truncated at its numbered stops (SMPLPP_IK_DBG_STOP: 19 at once, 20 behind the set-up, 21 behind
# the chain derivatives, 23 / 24 / 25 / 28 behind A0 / A1 / A2 / A3, 0 whole) for the capture excerpt, under rocprofv3 — phase
# lengths without stamps in the kernel.  usage (GPU box, repo root): bash tools/eval_stops.sh [chains] [latent]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/evalstops; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for s in 19 20 21 23 24 25 28 0; do
  SMPLPP_IK_DBG_STOP=$s timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$s -- python3 $ROOT/tools/mocap_only.py ${1:-8} $2 > $O/out$s.txt 2> $O/err$s.txt
  python3 - <<PY
import csv, glob
f = glob.glob("$O/s$s/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "ik_eval_kernel" in r["Name"]: print("stop %2d: ik_eval_kernel avg %7.2f us (%s calls)" % ($s, float(r["AverageNs"]) / 1e3, r["Calls"]))
PY
done
rm -rf $O
