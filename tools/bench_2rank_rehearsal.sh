#!/bin/bash
# rehearsal of the N = 2 bench on a one-GPU box: two ranks on GPU 0, gloo for the control plane and the final gather
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/bench_2rank; mkdir -p $O; cd $ROOT
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --all-ranks-on-device0 --no-cpu-baseline --sustained-steps 200 --mocap-frames 200 > $O/bench_2rank.json 2> $O/err.txt; echo rc=$?
tail -c 300 $O/err.txt; python - <<PY
import json
d = json.loads(open("$O/bench_2rank.json").read().strip().splitlines()[-1])
print("n_gpus", d["n_gpus"], "value %.3g" % d["value"], "gather_ms", d.get("final_gather_ms"), "ranks", d.get("ranks_reported_by_rccl"), "mocap restarts", d["mocap"]["restarts_this_rank"], "vposer frames", d["vposer_ik"]["frames_this_rank"])
PY
