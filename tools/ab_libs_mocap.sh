#!/bin/bash
# dev: whole-sequence capture fit (tools/mocap_full.py R) with the in-tree library and with ab/<name>.so, alternating
#   usage: bash tools/ab_libs_mocap.sh <name> [R] [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; V=$1; R=${2:-64}; N=${3:-2}
for i in $(seq $N); do
  echo -n "default: "; timeout -k 10 200 python tools/mocap_full.py $R 2>/dev/null | tail -n 1
  echo -n "$V: "; SMPLPP_HIP_LIB=$ROOT/ab/$V.so timeout -k 10 200 python tools/mocap_full.py $R 2>/dev/null | tail -n 1
done
