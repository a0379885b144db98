#!/bin/bash
# Everything profiles/ holds for a round, in one GPU call. usage (GPU box, repo root): bash tools/collect_round.sh <tag>
TAG=${1:-r03_a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O; cd $ROOT
bash tools/collect_profiles.sh $TAG > $O/collect.log 2>&1; tail -12 $O/collect.log
bash tools/ik_timeline.sh > $O/ik_timeline.txt 2>&1
bash tools/vposer_timeline.sh > $O/vposer_timeline.txt 2>&1
bash tools/mocap_full_profile.sh > $O/mocap_full_sequence.txt 2>&1
bash tools/mocap_timeline.sh 64 > $O/mocap_timeline.txt 2>&1
bash tools/mocap_chains_profile.sh 8 > $O/mocap_8chains_timeline.txt 2>&1
bash tools/mocap_chains_profile.sh 64 > $O/mocap_64chains_timeline.txt 2>&1
bash tools/mocap_chains_profile.sh 8 latent > $O/mocap_8chains_latent_timeline.txt 2>&1
bash tools/mocap_chains_profile.sh 64 latent > $O/mocap_64chains_latent_timeline.txt 2>&1
timeout -k 10 300 python tests/ik_stress_cases.py > $O/ik_sweep.txt 2>&1
cd /tmp && export TMPDIR=/tmp
SMPLPP_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $O/mk -- python3 $ROOT/tools/quick_ik.py > $O/mk.log 2>&1
head -40 $O/mk/*/*marker_api_trace.csv > $O/marker_trace_head.csv
rm -rf $O/mk $O/trace $O/pass*
cd $ROOT && timeout -k 10 300 python tools/fk_ramp.py > $O/fk_ramp.txt 2>/dev/null
ls $O
