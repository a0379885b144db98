#!/bin/bash
# Development aid: FK step time with each library variant given (names under ab/, "" = the in-tree build), interleaved rounds
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = "base" ]; then lib=""; else lib="$PWD/ab/$v.so"; fi
  echo -n "$v: "; SMPLPP_HIP_LIB=$lib timeout -k 10 120 python tools/quick_fk_bench.py 1024 400 2>/dev/null | tail -1
done; done
