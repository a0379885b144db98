#!/bin/bash
# Issue-side PMC breakdown of the FK kernels (which instruction class keeps the wavefront busy / waiting).
# usage (GPU box, repo root): bash tools/pmc_issue.sh <tag> [frames]
set -e
TAG=${1:-pmci}; N=${2:-1024}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS"
P2="SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
P3="SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_IFETCH SQ_THREAD_CYCLES_VALU SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_LDS"
i=1
for P in "$P1" "$P2" "$P3"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -- python3 $ROOT/tools/quick_fk_bench.py $N 10 > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.log; }
  i=$((i+1))
done
python3 $ROOT/tools/pmc_summary.py $OUT | tee $OUT/summary.txt
