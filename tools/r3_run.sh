#!/bin/bash
# dev aid: FK tests + quick bench + kernel times on the GPU box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r3run; mkdir -p $O; cd $ROOT
timeout -k 10 900 python -m pytest tests/test_fk_gpu.py -x -q > $O/pytest_fk.txt 2>&1; rc=$?; tail -15 $O/pytest_fk.txt
[ $rc -ne 0 ] && exit $rc
for e in "SMPLPP_POSE_FUSED=1" "SMPLPP_POSE_FUSED=0" "SMPLPP_POSE_FUSED=0 SMPLPP_POSE_WAVE=0"; do
  echo -n "$e: " | tee -a $O/quick.txt; env $e timeout -k 10 120 python tools/quick_fk_bench.py 1024 400 2>/dev/null | tail -1 | tee -a $O/quick.txt
done
export SMPLPP_POSE_FUSED=0
for n in 1 64 256 4096; do timeout -k 10 120 python tools/quick_fk_bench.py $n 300 2>/dev/null | tail -1 | tee -a $O/quick.txt; done
bash tools/kernel_times.sh 1024 > $O/ktimes.txt 2>&1; cat $O/ktimes.txt
