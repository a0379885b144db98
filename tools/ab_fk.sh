#!/bin/bash
# dev aid: A/B of library variants on one box, alternating, past the clock ramp. usage: bash tools/ab_fk.sh <variant.so> [reps]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; V=$1; R=${2:-3}
timeout -k 10 300 python -m pytest tests/test_fk_gpu.py -x -q 2>&1 | tail -2
for i in $(seq $R); do
  echo -n "default: "; timeout -k 10 120 python tools/quick_fk_bench.py 1024 2000 2>/dev/null | tail -1
  echo -n "$V: "; SMPLPP_HIP_LIB=$PWD/ab/$V timeout -k 10 120 python tools/quick_fk_bench.py 1024 2000 2>/dev/null | tail -1
done
