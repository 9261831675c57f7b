#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_multi_gpu.py -x -q -k "random or knobs or locality or edge or polytom or caterpillar or config2 or global or syn or sixteen or sub_batch or full_size or multiplacer or one_gpu" > gpurun_out/r3i_pytest.log 2>&1; tail -3 gpurun_out/r3i_pytest.log
run() { echo "$1 $2: $(env $1 timeout 200 python bench.py --cpu-queries 0 --steps 20 --warmup 3 $2 2>gpurun_out/err.txt | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["merge_ms"], d["roofline"]["coarse_ms"], d["host_buffer_path"]["placements_per_s"])')"; tail -2 gpurun_out/err.txt | grep -v amdgpu.ids; }
run A=1 ""
run UGP_NO_OVERLAP=1 ""
run A=1 "--shape sars2"
run UGP_NO_OVERLAP=1 "--shape sars2"
run A=1 "--queries 65536"
run UGP_NO_OVERLAP=1 "--queries 65536"
run A=1 "--ambiguous"
run A=1 "--nodes 100000 --queries 1024"
run UGP_NO_OVERLAP=1 "--nodes 100000 --queries 1024"
