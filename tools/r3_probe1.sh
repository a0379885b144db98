#!/bin/bash
# round-3 probe 1 (dev aid): launch-shape costs, store cache policies, phase stamps, kernel times, then the GPU test suite
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r3p1; mkdir -p $O
cd $ROOT
./ab/launch_shape > $O/launch_shape.txt 2>&1
for v in "" aux1 aux2 aux3 aux17 aux19; do
  if [ -z "$v" ]; then lib=""; else lib="$PWD/ab/$v.so"; fi
  echo -n "${v:-default}: " >> $O/aux.txt; SMPLPP_HIP_LIB=$lib timeout -k 10 120 python tools/quick_fk_bench.py 1024 400 2>/dev/null | tail -1 >> $O/aux.txt
done
SMPLPP_HIP_LIB=$PWD/ab/h512.so timeout -k 10 120 python tools/hphase_times.py > $O/hphase.txt 2>&1
bash tools/kernel_times.sh 1024 > $O/ktimes.txt 2>&1
cd $ROOT && timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
