"""Per-layer table of one train step from a rocprofv3 kernel trace of tools/train_probe.py: forward conv, LayerNorm
backward, weight gradient and transposed conv per layer, in us and useful TFLOP/s (effective MACs).
usage: python tools/train_layer_table.py <t_kernel_trace.csv> <batch>"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

path, B = sys.argv[1], int(sys.argv[2])
macs = bench.conv_effective_macs()
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'melspec_kernel' in r['Kernel_Name'] or 'melspec_r16_kernel' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
us = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
step = rows[a:b]
print(f'step: {(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e6:.3f} ms, {b - a} launches')
c0 = next(i for i, r in enumerate(step) if 'conv0_kernel' in r['Kernel_Name'])
tail = next(i for i, r in enumerate(step) if 'tail_kernel' in r['Kernel_Name'])
fwd, j = [], 0
main_fwd = step[c0].get('Stream_Id')          # (the re-pack of the weights runs next to conv0 / conv1 on other streams: not the forward's launches)
for r in step[c0 + 1:tail]:
    if r.get('Stream_Id') != main_fwd:
        continue
    n = r['Kernel_Name']
    if 'conv_gemm' in n:
        j += 1; fwd.append([j, us(r)])
    elif 'splitk_finish' in n and fwd:
        fwd[-1][1] += us(r)
print('forward:  ' + '  '.join(f'c{j}:{t:.0f}us/{2 * macs[j] * B / t / 1e6:.0f}TF' for j, t in fwd))
# weight gradients of the small layers run on a side stream (NAFP_OPT_BWD_OVERLAP=2), one layer behind the main
# stream's LayerNorm backward: they are attributed by their own launch order (15, 14, ...) and marked '*' -- their
# durations overlap the main stream's kernels and are not additive
main = next(r.get('Stream_Id') for r in step[tail + 1:] if 'ln_bwd_fused' in r['Kernel_Name'])
side = [us(r) for r in step[tail + 1:] if 'wgrad' in r['Kernel_Name'] and r.get('Stream_Id') != main]
layer, cur, out = 15, {'ln': 0.0, 'wg': 0.0, 'dg': 0.0}, []
for r in step[tail + 1:]:
    n = r['Kernel_Name']
    if 'wgrad' in n and r.get('Stream_Id') != main:
        continue
    if 'ln_bwd_fused' in n:
        if cur['ln']:
            out.append((layer, cur)); layer -= 1; cur = {'ln': 0.0, 'wg': 0.0, 'dg': 0.0}
        cur['ln'] += us(r)
    elif 'wgrad' in n:
        cur['wg'] += us(r)
    elif ('conv_gemm' in n and 'plain' in n) or 'plain_finish' in n or 'dgrad_ln' in n:
        cur['dg'] += us(r)
out.append((layer, cur))
print('backward: layer   ln_bwd us   wgrad us (TF)   dgrad us (TF)')
for l, c in out:
    if l < 1:
        print(f'  {l:2d}  {c["ln"]:9.1f}')
        continue
    m = 2 * macs[l] * B / 1e6
    star = ' '
    if 15 - l < len(side) and not c['wg']:
        c['wg'], star = side[15 - l], '*'
    print(f'  {l:2d}  {c["ln"]:9.1f}  {c["wg"]:9.1f}{star}({m / max(c["wg"], 1e-9):5.0f})  {c["dg"]:9.1f} ({m / max(c["dg"], 1e-9):5.0f})')
