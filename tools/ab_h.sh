#!/bin/bash
# Development aid: time the FK step with each ablation variant of skin_kernel_h (ab/h*.so, built by tools/build_variant.sh)
for v in "" 1 2 8 16 32 27; do
  if [ -z "$v" ]; then lib=""; else lib="$PWD/ab/h$v.so"; fi
  echo -n "ABL=${v:-0}: "; SMPLPP_HIP_LIB=$lib timeout -k 10 120 python tools/quick_fk_bench.py 1024 300 2>/dev/null | tail -1
done
SMPLPP_HIP_LIB=$PWD/ab/h256.so timeout -k 10 120 python tools/hslot_times.py 2>/dev/null
