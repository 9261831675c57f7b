cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python3 bench.py --queries 16384 --steps 2 --warmup 1 --cpu-queries 0"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d gpurun_out/pmc1 -o p --output-format csv -- $B > gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS -d gpurun_out/pmc2 -o p --output-format csv -- $B > gpurun_out/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d gpurun_out/pmc3 -o p --output-format csv -- $B > gpurun_out/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d gpurun_out/pmc4 -o p --output-format csv -- $B > gpurun_out/pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum -d gpurun_out/pmc5 -o p --output-format csv -- $B > gpurun_out/pmc5.log 2>&1
ls gpurun_out/pmc*/ ; tail -2 gpurun_out/pmc1.log
