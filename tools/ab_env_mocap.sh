#!/bin/bash
# dev: whole-sequence capture fit with and without an environment switch, alternating.  usage: bash tools/ab_env_mocap.sh VAR=VALUE [R] [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; E=$1; R=${2:-64}; N=${3:-2}
for i in $(seq $N); do
  echo -n "default: "; timeout -k 10 200 python tools/mocap_full.py $R 2>/dev/null | tail -n 1
  echo -n "$E: "; env $E timeout -k 10 200 python tools/mocap_full.py $R 2>/dev/null | tail -n 1
done
