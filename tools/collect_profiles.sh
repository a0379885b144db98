#!/bin/bash
# Round profile collection (run on the GPU box from the repo root): kernel-trace stats of the bench command (FK leg only,
# so the fused kernel's average is over batch-1024 launches), then separate --pmc passes for HBM traffic and MFMA busy.
# usage: bash tools/collect_profiles.sh <tag>
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --preroll-steps 600 --steps 100 --warmup 10 --no-ik --no-cpu-baseline --no-extra --no-side-form --sustained-steps 0 > $OUT/bench_under_trace.json 2> $OUT/trace.err
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
i=1
for P in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -- python3 $ROOT/bench.py --preroll-steps 600 --steps 20 --warmup 5 --no-ik --no-cpu-baseline --no-extra --no-side-form --sustained-steps 0 > $OUT/pass$i.json 2> $OUT/pass$i.err || echo "pass $i failed"
  i=$((i+1))
done
# the fp16x2 form, reported by bench.py as within_tolerance_form: its kernel trace and traffic counters (the form is read at model creation)
export SMPLPP_SKIN=h
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_h -- python3 $ROOT/bench.py --preroll-steps 600 --steps 100 --warmup 10 --no-ik --no-cpu-baseline --no-extra --no-side-form --sustained-steps 0 > $OUT/bench_under_trace_h.json 2> $OUT/trace_h.err
cp $(ls $OUT/trace_h/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_h.csv
for P in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -- python3 $ROOT/bench.py --preroll-steps 600 --steps 20 --warmup 5 --no-ik --no-cpu-baseline --no-extra --no-side-form --sustained-steps 0 > $OUT/pass$i.json 2> $OUT/pass$i.err || echo "pass $i failed"
  i=$((i+1))
done
unset SMPLPP_SKIN
python3 $ROOT/tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
cat $OUT/pmc_summary.txt
python3 $ROOT/tools/make_traffic_json.py $OUT/pmc_summary.txt $OUT/traffic.json 1024 $TAG $OUT/kernel_stats.csv $OUT/kernel_stats_h.csv
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$OUT/kernel_stats.csv")))[:6]:
    print("%-60s calls %5s avg %9.2f us %6s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
cd $ROOT && python3 bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json
