for b in ${BLOCKS:-1536 2304 3072}; do
  SMPLPP_SCAN_BLOCKS=$b timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/b_sb$b.json 2> gpurun_out/b_sb$b.err || exit 1
  echo "== blocks $b"; SMPLPP_SCAN_BLOCKS=$b bash tools/ik_timeline.sh 2>&1 | tail -7
done
