set -x
mkdir -p gpurun_out/r5b
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > gpurun_out/r5b/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5b/pytest.log
grep -n "worst\|passed\|failed\|rc " gpurun_out/r5b/pytest.log | tail -20
