#!/bin/bash
# rehearsal of the N > 1 bench on a one-GPU box: N ranks on GPU 0 (RCCL refuses two ranks on one device — "Duplicate GPU detected",
# profiles/r05_rccl_same_device_probe.txt — so gloo carries the control plane and the final gather), started by bench.py itself
# (no launcher in front: the way the driver starts `--gpus 1`).   usage (GPU box, repo root): bash tools/bench_2rank_rehearsal.sh [N, default 2]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; N=${1:-2}; O=$ROOT/gpurun_out/bench_${N}rank; mkdir -p $O; cd $ROOT
env -u RANK -u WORLD_SIZE -u LOCAL_RANK timeout -k 10 900 python bench.py --gpus $N --steps 20 --warmup 5 --backend gloo --all-ranks-on-device0 --no-cpu-baseline --sustained-steps 200 --mocap-frames 300 > $O/bench_${N}rank.json 2> $O/err.txt; echo rc=$?
tail -c 400 $O/err.txt; python - <<PY
import json
d = json.loads([l for l in open("$O/bench_${N}rank.json").read().strip().splitlines() if l.startswith("{")][-1])
print("n_gpus", d["n_gpus"], "value %.3g" % d["value"], "gather_ms %.1f" % d["final_gather_ms"], "backend", d["collective_backend"], "ranks counted by all-reduce", d["ranks_counted_by_allreduce"],
      "ms_per_step per rank", d["ms_per_step_ranks"], "chains per rank", d["mocap"]["chains_per_rank"], "per_frame_us per rank", ["%.1f" % x for x in d["mocap"]["per_frame_us_per_rank"]],
      "vposer frames", d["vposer_ik"]["frames_this_rank"])
PY
