cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python3 bench.py --queries 16384 --steps 2 --warmup 1 --cpu-queries 0 --sort-by-source"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d gpurun_out/ps1 -o p --output-format csv -- $B > gpurun_out/ps1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR TCC_HIT_sum TCC_MISS_sum -d gpurun_out/ps2 -o p --output-format csv -- $B > gpurun_out/ps2.log 2>&1
tail -1 gpurun_out/ps1.log | cut -c1-100
