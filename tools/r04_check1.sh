#!/bin/bash
# round 4, first GPU check: new invariance tests, the IK sweep's spread + outlier dump, decoder instantiations timed
O=gpurun_out/r04b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_vposer_gpu.py tests/test_mocap_gpu.py tests/test_bench_gpu.py -m gpu -q > $O/pytest.txt 2>&1; echo rc=$? >> $O/pytest.txt
tail -15 $O/pytest.txt
timeout -k 10 300 python tests/ik_stress_cases.py $O/ik_outliers_dump.npz > $O/stress.txt 2>&1; echo rc=$? >> $O/stress.txt
tail -40 $O/stress.txt
for n in 128 256 512; do
  for jf in 1 2; do echo "n=$n SMPLPP_VPOSER_JAC=$jf"; SMPLPP_VPOSER_JAC=$jf timeout -k 10 120 python tools/quick_vposer_ik.py $n 50 2>&1 | tail -1; done
done > $O/vposer_forms.txt 2>&1
cat $O/vposer_forms.txt
