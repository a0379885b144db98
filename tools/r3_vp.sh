#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
timeout -k 10 600 python -m pytest tests/test_vposer_gpu.py -x -q 2>&1 | tail -8
for i in 1 2; do
  echo -n "jac2: "; timeout -k 10 200 python tools/quick_vposer_ik.py 512 50 2>/dev/null | tail -1
  echo -n "jac1: "; SMPLPP_VPOSER_JAC=1 timeout -k 10 200 python tools/quick_vposer_ik.py 512 50 2>/dev/null | tail -1
done
bash tools/vposer_timeline.sh 2>/dev/null | grep -E "vposer_jac|VPoser IK" | head -6
