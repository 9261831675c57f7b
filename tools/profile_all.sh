#!/bin/bash
# The round's counter profiles, one after the other (each: kernel stats + five PMC passes, tools/profile_round.sh):
#   <tag>           the headline workload                      <tag>_dense    the walk without pruning, 2 048 samples (UGP_NO_PRUNE=1)
#   <tag>_sars2     config 3's size: SARS-CoV-2-shaped 15 M nodes x 10 000    <tag>_config5  high-ambiguity queries on the headline tree
# bash tools/profile_all.sh r06   (on the GPU box; then python tools/summarize_profile.py <tag...> here)
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh ${TAG} "" > /dev/null 2>&1
export UGP_NO_PRUNE=1
bash tools/profile_round.sh ${TAG}_dense "--queries 2048" > /dev/null 2>&1
unset UGP_NO_PRUNE
bash tools/profile_round.sh ${TAG}_sars2 "--shape sars2 --nodes 15000000 --queries 10000" > /dev/null 2>&1
bash tools/profile_round.sh ${TAG}_config5 "--ambiguous" > /dev/null 2>&1
python3 tools/summarize_profile.py ${TAG} > /dev/null 2>&1
python3 tools/summarize_profile.py ${TAG}_dense "--queries 2048   (with UGP_NO_PRUNE=1 in the environment)" > /dev/null 2>&1
python3 tools/summarize_profile.py ${TAG}_sars2 "--shape sars2 --nodes 15000000 --queries 10000" > /dev/null 2>&1
python3 tools/summarize_profile.py ${TAG}_config5 "--ambiguous" > /dev/null 2>&1
for t in ${TAG} ${TAG}_dense ${TAG}_sars2 ${TAG}_config5; do tail -c 300 gpurun_out/${t}_stats.log; echo; done
mkdir -p gpurun_out/profiles_${TAG}; cp profiles/${TAG}*_kernel_stats.csv profiles/${TAG}*_pmc_summary.json gpurun_out/profiles_${TAG}/ 2>/dev/null; ls gpurun_out/profiles_${TAG}
