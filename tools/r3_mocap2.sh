#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
timeout -k 10 900 python -m pytest tests/test_ik_gpu.py tests/test_mocap_gpu.py -x -q 2>&1 | tail -4
bash tools/r3_mocap.sh
bash tools/r3_mocap_dbg.sh
timeout -k 10 300 python tools/quick_ik.py 2>/dev/null | tail -1
