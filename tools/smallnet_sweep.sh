# persistent small-layer launch: chunk size / grid sweep at the bench batch (gpurun -- bash tools/smallnet_sweep.sh)
run() { env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-train --no-e2e --fullscale-rows 0 --no-pipelined 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); pc=d['stage_ms_per_step']['per_conv']; print('$*', d['value'], d['ms_per_step'], d['roofline']['frac'], 'small layers ms:', round(sum(pc[10:16]),4))"; }
run NAFP_SMALLNET=0
for pr in 0 1 2; do for st in 12 16 20; do for w in 256 320 384 512; do run NAFP_SMALLNET_PRIO=$pr NAFP_SMALLNET_STEPS=$st NAFP_SMALLNET_WGS=$w; done; done; done
