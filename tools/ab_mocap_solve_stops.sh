#!/bin/bash
# dev: average duration of ik_solve_kernel in the capture-excerpt leg truncated at debug stops
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for st in ${STOPS:-1 4 3 0}; do
  OUT=$ROOT/gpurun_out/mstop$st; rm -rf $OUT; mkdir -p $OUT
  SMPLPP_IK_DBG_STOP=$st rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/mocap_only.py ${R:-8} > $OUT/out.txt 2> $OUT/err.txt
  echo "== stop $st: $(grep -h ik_solve_kernel $OUT/*/*kernel_stats.csv | awk -F, '{print $(NF-4)}')"
done
