#!/bin/bash
# Development aid: registers, scratch and spills of every kernel of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
#   usage: tools/kernel_resources.sh ik.hip [extra flags]
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; src=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 "$@" -Rpass-analysis=kernel-resource-usage -c "$ROOT/smplpp_amd/csrc/$src" -o /tmp/kres.o 2>&1 | python3 -c '
import re, sys
cur = None; rows = []
for ln in sys.stdin:
    m = re.search(r"Function Name: (\S+)", ln)
    if m: cur = {"name": m.group(1)[:70]}; rows.append(cur); continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/(?:lane|block)\])?(?: \[waves/SIMD\])?: (\S+)", ln)
    if m and cur is not None: cur[m.group(1).strip()] = m.group(2)
for r in rows:
    print("%-72s VGPR %4s AGPR %3s scratch %4s sgpr-spill %4s vgpr-spill %3s LDS %6s occ %s" % (r["name"], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("LDS Size"), r.get("Occupancy")))
'
