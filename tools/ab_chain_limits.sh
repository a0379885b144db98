#!/bin/bash
# dev: IK iteration time with one chain artificially shortened (results are garbage; timing only)
for st in 0 10 1; do
  SMPLPP_IK_DBG_STOP=$st timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/b_cl$st.json 2> gpurun_out/b_cl$st.err || exit 1
done
