cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/trace -o r01 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof/bench_trace.log 2>&1
ls -R gpurun_out/prof/trace | head -20
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d gpurun_out/prof/pmc1 -o r01 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d gpurun_out/prof/pmc2 -o r01 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof/bench_pmc2.log 2>&1
ls -R gpurun_out/prof | head -40
