#!/bin/bash
# Development aid: ab/<name>.so = the whole library as it was at git revision <rev> (A/B of two states of the tree on ONE box:
# SMPLPP_HIP_LIB=$PWD/ab/<name>.so python ...).   usage: [IK_FLAGS="-DSMPLPP_EVAL_STAMPS"] tools/build_rev.sh <name> <rev>
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; name=$1; rev=$2
T=/tmp/rev_$name; rm -rf $T; mkdir -p $T/smplpp_amd/csrc $T/include "$ROOT/ab"
(cd "$ROOT" && git archive $rev smplpp_amd/csrc include | tar -x -C $T)
objs=""
for src in $T/smplpp_amd/csrc/*.hip; do
  b=$(basename $src); extra="-fno-slp-vectorize"
  [ "$b" = skin_h.hip ] && extra="$extra -mllvm -amdgpu-mfma-vgpr-form"
  case $b in skin_p.hip|skin_b.hip|skin_h.hip|fk.hip|ik.hip) ;; *) extra="" ;; esac
  [ "$b" = ik.hip ] && extra="$extra $IK_FLAGS"
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function $extra -c $src -o $T/$b.o &
  objs="$objs $T/$b.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/ab/$name.so" $objs
echo "built ab/$name.so from $rev"
