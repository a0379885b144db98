#!/bin/bash
# dev aid: whole GPU suite, then a marker trace of a short IK run
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r3full; mkdir -p $O; cd $ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; tail -15 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
SMPLPP_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $O/mk -- python3 $ROOT/tools/quick_ik.py > $O/mk.log 2>&1
ls $O/mk/*/ | head; head -12 $O/mk/*/*marker_api_trace.csv
