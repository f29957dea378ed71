cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
rm -rf gpurun_out/prof/pmcA gpurun_out/prof/pmcB
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE -d gpurun_out/prof/pmcA -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof/bench_pmcA.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA -d gpurun_out/prof/pmcB -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof/bench_pmcB.log 2>&1
python - <<'PY'
import sqlite3
for db in ['pmcA','pmcB']:
    c=sqlite3.connect(f'gpurun_out/prof/{db}/r_results.db')
    q="select grid_size, counter_name, avg(value), avg(duration), count(*) from counters_collection where kernel_name like '%conv_gemm%' group by grid_size, counter_name order by grid_size desc"
    for r in c.execute(q):
        if r[0] in (2621440, 1310720, 40960): print(db, r)
PY
