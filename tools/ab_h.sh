#!/bin/bash
# The ablation ladder of the fused kernel (skin_kernel_h, SKINH_ABL switches in skin_h.hip; results are WRONG under every mask but 0,
# timing only): what a step costs as built, without its barriers (1), without its DMAs (2), without the blend phase (8), without
# fragment reads (16), without stores (32), the GEMM + blend MFMAs alone (1|2|16|32 = 51), the GEMM MFMAs alone (27 = 1|2|8|16), and
# the per-slot / per-phase stamps (256, 512).  This is what the "a loop without any overhead ends at 32-34 us" claim of DESIGN.md §3.2
# rests on (profiles/r03_c_fk_ablations.txt was made by the first version of this script; restored in round 5).
#   step 1, anywhere (cross-compiles):   bash tools/ab_h.sh build
#   step 2, on the GPU box (repo root):  bash tools/ab_h.sh run > gpurun_out/fk_ablations.txt
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"
MASKS="1 2 8 16 32 51 27"
if [ "$1" = build ]; then
  for m in $MASKS 256 512; do bash tools/build_variant.sh h$m skin_h.hip -DSKINH_ABL=$m > /dev/null || exit 1; done
  ls ab/h*.so; exit 0
fi
echo -n "ABL=0: "; timeout -k 10 120 python3 tools/quick_fk_bench.py 1024 2000 2>/dev/null | tail -1
for m in $MASKS; do
  echo -n "ABL=$m: "; SMPLPP_HIP_LIB=$ROOT/ab/h$m.so timeout -k 10 120 python3 tools/quick_fk_bench.py 1024 2000 2>/dev/null | tail -1 || exit 1
done
echo -n "ABL=0: "; timeout -k 10 120 python3 tools/quick_fk_bench.py 1024 2000 2>/dev/null | tail -1
SMPLPP_HIP_LIB=$ROOT/ab/h256.so timeout -k 10 120 python3 tools/hslot_times.py 2>/dev/null | grep -v amdgpu.ids
SMPLPP_HIP_LIB=$ROOT/ab/h512.so timeout -k 10 120 python3 tools/hphase_times.py 2>/dev/null | grep -v amdgpu.ids
