#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
timeout -k 10 900 python -m pytest tests/test_ik_gpu.py tests/test_mocap_gpu.py tests/test_vposer_gpu.py -x -q 2>&1 | tail -4
bash tools/r3_mocap8.sh 8 2>/dev/null | head -10
bash tools/r3_mocap8.sh 64 2>/dev/null | head -10
timeout -k 10 300 python tools/quick_ik.py 2>/dev/null | tail -1
