#!/bin/bash
# The ablation ladder of the exact fused kernel (skin_kernel_e, SKINE_ABL switches in skin_e.hip; results are WRONG under every mask
# but 0, timing only): the step as built, without the epilogue (1: skinning + stores of the previous item), without ring DMAs (2),
# without barriers (8), without fragment reads (16), without stores (32), MFMAs + epilogue only (26), MFMAs alone (59).
#   step 1, anywhere (cross-compiles):   bash tools/ab_e.sh build [extra hipcc flags]
#   step 2, on the GPU box (repo root):  bash tools/ab_e.sh run > gpurun_out/fk_e_ablations.txt
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"
MASKS="1 2 8 16 32 26 59"
if [ "$1" = build ]; then
  shift; mkdir -p ab
  for m in $MASKS; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-slp-vectorize "$@" -DSKINE_ABL=$m -c smplpp_amd/csrc/skin_e.hip -o /tmp/variant_e$m.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/e$m.so /tmp/variant_e$m.o $(ls smplpp_amd/build/*.o | grep -v "/skin_e.hip.o") || exit 1
  done
  ls ab/e*.so; exit 0
fi
echo -n "ABL=0: "; timeout -k 10 120 python3 tools/quick_fk_bench.py 1024 2000 2>/dev/null | tail -1
for m in $MASKS; do
  echo -n "ABL=$m: "; SMPLPP_HIP_LIB=$ROOT/ab/e$m.so timeout -k 10 120 python3 tools/quick_fk_bench.py 1024 2000 2>/dev/null | tail -1 || exit 1
done
echo -n "ABL=0: "; timeout -k 10 120 python3 tools/quick_fk_bench.py 1024 2000 2>/dev/null | tail -1
