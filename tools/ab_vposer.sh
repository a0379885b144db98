#!/bin/bash
# A/B on one box: configs[4] leg with the library in the tree against ab/<name>.so (default vpold), alternating, three pairs
N=${1:-vpold}
for i in 1 2 3; do
  echo -n "new: "; timeout -k 10 120 python tools/quick_vposer_ik.py 512 50 2>&1 | tail -1
  echo -n "$N: "; SMPLPP_HIP_LIB=$PWD/ab/$N.so timeout -k 10 120 python tools/quick_vposer_ik.py 512 50 2>&1 | tail -1
done
