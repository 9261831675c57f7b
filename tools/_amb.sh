cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "not million and not 10m and not full_size" > gpurun_out/b3_pytest.log 2>&1; tail -3 gpurun_out/b3_pytest.log
UGP_NO_OVERLAP=1 python3 tools/sweep_knobs.py "UGP_LDS_SLOTS=8" "UGP_LDS_SLOTS=10" "UGP_LDS_SLOTS=7" "UGP_UNIT_GROW=8" 2>&1 | tail -5
for S in "A=1" "UGP_LDS_SLOTS=8" "UGP_LDS_SLOTS=10" "UGP_UNIT_GROW=8" "UGP_SHARED_WAVES=7" "UGP_SHARED_WAVES=9"; do
echo "== $S"
env $S python3 bench.py --steps 20 --warmup 5 --cpu-queries 0 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_alone'])"
done
