#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r3stamps; mkdir -p $O; cd $ROOT
SMPLPP_POSE_FUSED=0 SMPLPP_HIP_LIB=$PWD/ab/h512.so timeout -k 10 120 python tools/hphase_times.py > $O/hphase2.txt 2>&1; cat $O/hphase2.txt
