#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r3stamps; mkdir -p $O; cd $ROOT
SMPLPP_HIP_LIB=$PWD/ab/h1536.so timeout -k 10 120 python tools/pose_stamps_h.py > $O/pose.txt 2>&1; cat $O/pose.txt
