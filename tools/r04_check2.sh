#!/bin/bash
O=gpurun_out/r04e; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ik_gpu.py tests/test_mocap_gpu.py tests/test_vposer_gpu.py -m gpu -q -x > $O/pytest.txt 2>&1; echo rc=$? >> $O/pytest.txt
tail -8 $O/pytest.txt
for R in 8 64; do timeout -k 10 200 python tools/mocap_full.py $R 800 2>&1 | tail -1; done | tee $O/mocap_plain.txt
timeout -k 10 200 python tools/quick_ik.py 2>&1 | tail -1 | tee $O/ik_plain.txt
bash tools/mocap_chains_profile.sh 8 > $O/mocap_8chains.txt 2>&1; head -10 $O/mocap_8chains.txt
