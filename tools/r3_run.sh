#!/bin/bash
# dev aid: FK tests + quick bench + kernel times + pose stamps on the GPU box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r3run; mkdir -p $O; cd $ROOT; rm -f $O/quick.txt
timeout -k 10 900 python -m pytest tests/test_fk_gpu.py -x -q > $O/pytest_fk.txt 2>&1; rc=$?; tail -5 $O/pytest_fk.txt
[ $rc -ne 0 ] && exit $rc
SMPLPP_HIP_LIB=$PWD/ab/pst.so timeout -k 10 120 python tools/pose_stamps.py 2>/dev/null | tee $O/pose_stamps.txt
for n in 1024 1 64 256 4096; do timeout -k 10 120 python tools/quick_fk_bench.py $n 400 2>/dev/null | tail -1 | tee -a $O/quick.txt; done
bash tools/kernel_times.sh 1024 > $O/ktimes.txt 2>&1; grep -E "pose|skin|gap" $O/ktimes.txt
timeout -k 10 300 python tools/quick_ik.py 2>/dev/null | tail -1 | tee $O/ik.txt
