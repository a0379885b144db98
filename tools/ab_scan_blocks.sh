for b in 768 1536 3072 6144; do
  SMPLPP_SCAN_BLOCKS=$b timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/b_sb$b.json 2> gpurun_out/b_sb$b.err || exit 1
done
