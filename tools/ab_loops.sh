#!/bin/bash
# Development aid: the three IK loops, default library against ab/<variant>.so, alternating on ONE box (boxes differ by 5-10 %).
#   usage (GPU box, repo root): bash tools/ab_loops.sh <variant.so> [pairs]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; V=$1; P=${2:-2}
for i in $(seq $P); do
  for lib in "$ROOT/ab/$V" ""; do
    tag=${lib##*/}; tag=${tag:-default}
    echo "== $tag"
    SMPLPP_HIP_LIB=$lib timeout -k 10 120 python3 $ROOT/tools/quick_ik.py || exit 1
    SMPLPP_HIP_LIB=$lib timeout -k 10 120 python3 $ROOT/tools/mocap_only.py 8 || exit 1
    SMPLPP_HIP_LIB=$lib timeout -k 10 120 python3 $ROOT/tools/mocap_only.py 64 || exit 1
    SMPLPP_HIP_LIB=$lib timeout -k 10 120 python3 $ROOT/tools/quick_vposer_ik.py || exit 1
  done
done
