#!/bin/bash
# Development aid: build ab/<name>.so = the library with one translation unit recompiled with extra flags
#   usage: tools/build_variant.sh <name> <file.hip> [flags...]     (run SMPLPP_HIP_LIB=$PWD/ab/<name>.so ... to A/B on one box)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; src=$2; shift 2
mkdir -p "$ROOT/ab"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form "$@" -c "$ROOT/smplpp_amd/csrc/$src" -o "/tmp/variant_$name.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/ab/$name.so" "/tmp/variant_$name.o" $(ls "$ROOT"/smplpp_amd/build/*.o | grep -v "/$src.o")
echo "built ab/$name.so"
