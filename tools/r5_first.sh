set -x
mkdir -p gpurun_out/r5a
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5a/pytest.log
tail -5 gpurun_out/r5a/pytest.log
timeout -k 10 150 python tools/rccl_same_device_probe.py > gpurun_out/r5a/rccl_probe.log 2>&1; echo "probe rc $?" >> gpurun_out/r5a/rccl_probe.log
tail -15 gpurun_out/r5a/rccl_probe.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r5a/bench.json 2> gpurun_out/r5a/bench.err; echo "bench rc $?"
tail -c 3000 gpurun_out/r5a/bench.json
