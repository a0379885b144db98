#!/bin/bash
# round 4 baselines before the kernel work: IK / VPoser / capture timelines at 8 and 64 chains
O=gpurun_out/r04d; mkdir -p $O
bash tools/ik_timeline.sh > $O/ik_timeline.txt 2>&1
bash tools/vposer_timeline.sh > $O/vposer_timeline.txt 2>&1
bash tools/mocap_chains_profile.sh 8 > $O/mocap_8chains.txt 2>&1
bash tools/mocap_chains_profile.sh 64 > $O/mocap_64chains.txt 2>&1
tail -28 $O/ik_timeline.txt; tail -34 $O/vposer_timeline.txt; cat $O/mocap_8chains.txt; head -12 $O/mocap_64chains.txt
