#!/bin/bash
# Development aid: step times of variant builds of skin_kernel_x (ab/<name>.so built from skin_x.hip with -DSKINX_ABL / -DSKINX_RD ...;
# results are wrong under every ablation mask, timing only), one box.   usage (GPU box, repo root): bash tools/ab_x.sh name...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  lib=""; [ "$v" != built ] && lib=$ROOT/ab/$v.so
  echo -n "$v: "; SMPLPP_SKIN=x SMPLPP_HIP_LIB=$lib timeout -k 10 120 python3 $ROOT/tools/quick_fk_bench.py 1024 2000 2>/dev/null | tail -1 || exit 1
done
