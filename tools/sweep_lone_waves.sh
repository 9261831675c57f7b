cd $GRAFT_REPO_ROOT
for f in "--ambiguous" "--shape sars2" "--nodes 1000000" "--queries 65536" "--queries 4096" "--shape sars2 --nodes 15000000 --queries 10000"; do
  echo "== $f"; bash tools/sweep_env.sh "--no-overlap $f" "UGP_WAVES_PER_CU=16" "UGP_WAVES_PER_CU=12" "UGP_WAVES_PER_CU=10"
done
