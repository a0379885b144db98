#!/bin/bash
# dev: average duration of ik_solve_kernel truncated at each debug stop (serial schedule)
for st in ${STOPS:-1 31 32 33 34 0}; do
  echo "== stop $st"
  SMPLPP_IK_OVERLAP=0 SMPLPP_IK_DBG_STOP=$st bash tools/ik_kernel_times.sh 2>&1 | grep -i "solve"
done
