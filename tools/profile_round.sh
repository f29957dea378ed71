#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel-trace stats of the bench command, separate
# PMC passes for HBM traffic (FETCH_SIZE / WRITE_SIZE cannot share a pass), and the plain bench.
# Results land in gpurun_out/prof_<tag>/; tools/profile_summarize.py turns them into profiles/.
TAG=${1:-r01}
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
python bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python bench.py --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-pipelined --no-train --no-e2e > $O/bench_trace.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-pipelined --no-train --no-e2e > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-pipelined --no-train --no-e2e > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -o p -- python bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-pipelined --no-train --no-e2e > /dev/null 2> $O/pmc_sq.err
# second SQ pass: the front end's issue mix (bound the STFT/mel kernel with counters)
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq2 -o p -- python bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-pipelined --no-train --no-e2e > /dev/null 2> $O/pmc_sq2.err
find $O -name "*.csv" | head -20
tail -c 600 $O/bench.json
# [r6] the exact-split forward (NAFP_OPT_BF16X3 = 2) alone: kernel trace and one SQ pass (matrix-pipe busy share, clock)
X6_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/x6_trace -o t -- python tools/x6_per_conv.py > $O/x6_per_conv.txt 2> $O/x6_trace.err
X6_ONLY=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/x6_pmc_sq -o p -- python tools/x6_per_conv.py > /dev/null 2> $O/x6_pmc_sq.err
python tools/x6_per_conv.py > $O/x6_per_conv_plain.txt 2>&1
# train step (BSZ 1280, Adam: SURVEY 8d config 3) kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_trace -o t -- python tools/train_probe.py 1280 adam 5 > $O/train_probe.txt 2> $O/train_trace.err
# train step at the headline batch (BASELINE configs[3] on one GPU: BSZ 5120, LAMB)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train5120_trace -o t -- python tools/train_probe.py 5120 lamb 3 > $O/train5120_probe.txt 2> $O/train5120_trace.err
# the 8-GPU operating point of the train metric on one GPU: per-rank batch 640, LAMB
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train640_trace -o t -- python tools/train_probe.py 640 lamb 5 > $O/train640_probe.txt 2> $O/train640_trace.err
# [r6] the same two steps with forward_train and the transposed convs on the exact 3-way bf16 split (NAFP_BF16X3=2 in the environment of the probe)
export NAFP_BF16X3=2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train5120x6_trace -o t -- python tools/train_probe.py 5120 lamb 3 > $O/train5120x6_probe.txt 2> $O/train5120x6_trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train640x6_trace -o t -- python tools/train_probe.py 640 lamb 5 > $O/train640x6_probe.txt 2> $O/train640x6_trace.err
unset NAFP_BF16X3
for b in 640 1280 5120; do d=train_trace; [ $b = 640 ] && d=train640_trace; [ $b = 5120 ] && d=train5120_trace; python tools/train_layer_table.py $(find $O/$d -name "*kernel_trace.csv" | head -1) $b > $O/layers_$b.txt 2>&1; done
# eval side: exact search of 38,000 query segments over 10 M resident fingerprints; training loader
rocprofv3 --kernel-trace --stats --output-format csv -d $O/search_trace -o t -- python tools/search_bench.py 10000000 38000 2 > $O/search_bench.txt 2> $O/search_trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/loader_trace -o t -- python tools/loader_bench.py 300 > $O/loader_bench.txt 2> $O/loader_trace.err
