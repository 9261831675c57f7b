#!/bin/bash
# round-3 first visit: parity tests, bench, stats and a few unit-size probes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3a_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3a_pytest.log
tail -3 gpurun_out/r3a_pytest.log
timeout 600 python bench.py --cpu-queries 0 > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err; tail -c 1500 gpurun_out/r3a_bench.json
UGP_STATS=1 timeout 600 python bench.py --cpu-queries 0 --steps 2 > gpurun_out/r3a_stats.json 2> gpurun_out/r3a_stats.err; grep "ugp stats" gpurun_out/r3a_stats.err | tail -12
for hc in 2 4 8 32; do
  echo "HEAVY_CHUNKS=$hc: $(UGP_HEAVY_CHUNKS=$hc timeout 600 python bench.py --cpu-queries 0 --steps 5 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["merge_ms"])')"
done
for uc in 4 8 32; do
  echo "UNIT_CHUNKS=$uc: $(UGP_UNIT_CHUNKS=$uc timeout 600 python bench.py --cpu-queries 0 --steps 5 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["merge_ms"])')"
done
for w in 8 12; do
  echo "WAVES_PER_CU=$w: $(UGP_WAVES_PER_CU=$w timeout 600 python bench.py --cpu-queries 0 --steps 5 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["merge_ms"])')"
done
