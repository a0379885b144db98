#!/bin/bash
# Collect PMC counters for the FK kernels at batch N (separate passes; --pmc only with --kernel-trace).
# usage (on the GPU box, from the repo root): bash tools/pmc_fk.sh <tag> [frames]
set -e
TAG=${1:-pmc}
N=${2:-1024}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES"
P2="FETCH_SIZE GRBM_GUI_ACTIVE"
P3="WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P4="TCC_HIT_sum TCC_MISS_sum"
i=1
for P in "$P1" "$P2" "$P3" "$P4"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -- python3 $ROOT/tools/quick_fk_bench.py $N 10 > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.log; }
  i=$((i+1))
done
python3 $ROOT/tools/pmc_summary.py $OUT | tee $OUT/summary.txt
