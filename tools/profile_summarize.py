#!/usr/bin/env python
"""Turn gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) into the
tracked artefacts under profiles/:  <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats
summary, verbatim), <tag>_summary.md, <tag>_bench.json, and profiles/traffic.json (per-launch
HBM bytes of the dominant kernel, read by bench.py for roofline.traffic).

HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE come from separate
--pmc passes, are in KiB, and on gfx950 FETCH_SIZE counts a wide (16 B/lane) coalesced read
stream at half its bytes -> doubled here.  WRITE_SIZE is taken as reported (uncalibrated).
"""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src = os.path.join(ROOT, 'gpurun_out', f'prof_{tag}')
dst = os.path.join(ROOT, 'profiles')
os.makedirs(dst, exist_ok=True)
sys.path.insert(0, ROOT)
import bench  # noqa: E402

GEMM = 'conv_gemm'
BSZ = bench.BSZ


def read_csv(path):
    with open(path, newline='') as f:
        return list(csv.DictReader(f))


shutil.copy(os.path.join(src, 'trace', 't_kernel_stats.csv'), os.path.join(dst, f'{tag}_kernel_stats.csv'))
bench_line = [l for l in open(os.path.join(src, 'bench.json')).read().splitlines() if l.startswith('{')][-1]
bj = json.loads(bench_line)
json.dump(bj, open(os.path.join(dst, f'{tag}_bench.json'), 'w'), indent=1)

stats = read_csv(os.path.join(src, 'trace', 't_kernel_stats.csv'))
trace = read_csv(os.path.join(src, 'trace', 't_kernel_trace.csv'))
# per-shape durations of the GEMM conv: group by grid size (15 distinct layer shapes share some grids)
by_grid = defaultdict(list)
for r in trace:
    if GEMM in r['Kernel_Name']:
        g = (int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r['Grid_Size']), int(r.get('Grid_Size_Y', 1) or 1))
        by_grid[g].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
# full-mode launches: drop the 30 PLAIN launches of set_weights (one sample: tiny grids, issued once)
gemm_all = sorted(((g, v) for g, v in by_grid.items()), key=lambda t: -sum(t[1]))


def pmc(dirname, counter):
    rows = read_csv(os.path.join(src, dirname, 'p_counter_collection.csv'))
    out = defaultdict(list)
    for r in rows:
        if r['Counter_Name'] == counter:
            out[r['Kernel_Name']].append((int(r['Grid_Size']), float(r['Counter_Value']),
                                          int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    return out


fetch, write = pmc('pmc_fetch', 'FETCH_SIZE'), pmc('pmc_write', 'WRITE_SIZE')


def per_step_sum(d, name_part, min_grid=0):
    """sum of a counter over the launches of one bench step (launch count / steps)."""
    vals = [v for k, lst in d.items() if name_part in k for (g, v, _) in lst if g >= min_grid]
    return sum(vals), len(vals)


macs = bench.conv_effective_macs()


def per_bench_step(dirname, counter):
    """Mean over the bench steps of the pass (a step = the dispatches from one front-end kernel to the next) of the counter
    summed over the step's GEMM-conv kernels AND their split-K finish kernels; the one-off launches of set_weights (G / Hb
    images) and anything before the first step are left out.  Returns (mean KiB per step, steps)."""
    rows = [r for r in read_csv(os.path.join(src, dirname, 'p_counter_collection.csv')) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'melspec_r16_kernel' in r['Kernel_Name'] or 'melspec_kernel' in r['Kernel_Name']]
    per = []
    for a_, b_ in zip(marks, marks[1:] + [len(rows)]):
        ks = [float(r['Counter_Value']) for r in rows[a_:b_] if GEMM in r['Kernel_Name'] or 'splitk_finish' in r['Kernel_Name']]
        n_gemm = sum(1 for r in rows[a_:b_] if GEMM in r['Kernel_Name'])
        if n_gemm == 15:                          # a whole forward step (the per-conv pass and partial tails are dropped)
            per.append(sum(ks))
    return (sum(per) / len(per) if per else None), len(per)


f_step, f_n = per_bench_step('pmc_fetch', 'FETCH_SIZE')
w_step, w_n = per_bench_step('pmc_write', 'WRITE_SIZE')
# per launch = per step / 15 (the 15 GEMM-conv launches of a step, their split-K finish kernels included)
fetch_bytes_per_launch = 2.0 * f_step * 1024 / 15 if f_step else None       # gfx950 x2 correction
write_bytes_per_launch = w_step * 1024 / 15 if w_step else None
alg_bytes_per_launch = None
# algorithmic HBM bytes of the 15 convs per step: each activation read once + written once (z tensors)
geo_out = []
F, T, C = 256, 32, 1
hidden = [128, 128, 256, 256, 512, 512, 1024, 1024]
st = [2, 2, 2, 2, 1, 2, 1, 2]
sizes = []
for i in range(8):
    T = -(-T // st[i]); C = hidden[i]; sizes.append(F * T * C)
    F = -(-F // 2); sizes.append(F * T * C)
alg = sum((sizes[j - 1] + sizes[j]) * 4 * BSZ for j in range(1, 16))
alg_bytes_per_launch = alg / 15

traffic = None
if fetch_bytes_per_launch and write_bytes_per_launch:
    traffic = fetch_bytes_per_launch + write_bytes_per_launch


def kernel_traffic(name_part):
    """(fetch bytes x2-corrected, write bytes, mean duration us, launches) per launch of the kernels matching name_part."""
    fs = [(v, d) for k, lst in fetch.items() if name_part in k for (g, v, d) in lst]
    ws = [(v, d) for k, lst in write.items() if name_part in k for (g, v, d) in lst]
    if not fs or not ws:
        return None
    return {'fetch_bytes_x2_corrected': 2.0 * 1024 * sum(v for v, _ in fs) / len(fs),
            'write_bytes': 1024 * sum(v for v, _ in ws) / len(ws),
            'mean_us_under_pmc': sum(d for _, d in fs + ws) / len(fs + ws) / 1e3, 'launches': len(fs)}


# front end (north star: "rocprof HBM GB/s on the STFT/mel path") and the Cin = 1 conv, per launch of 640 segments
front = {}
for key, part in (('melspec_kernel', 'melspec_kernel'), ('melspec_r16_kernel', 'melspec_r16_kernel'), ('melspec_finalize_kernel', 'melspec_finalize'), ('conv0_kernel', 'conv0_kernel')):
    kt = kernel_traffic(part)
    if kt:
        kt['bytes_per_segment'] = (kt['fetch_bytes_x2_corrected'] + kt['write_bytes']) / BSZ
        front[key] = kt
fe_total = sum(front[k]['fetch_bytes_x2_corrected'] + front[k]['write_bytes'] for k in front if k.startswith('melspec'))
json.dump({'tag': tag, 'kernel': 'conv_gemm_* (the 15 GEMM-conv launches of a step)', 'per_launch_bytes': traffic,
           'frontend': {'kernels': front, 'bytes_per_launch': fe_total, 'bytes_per_segment': fe_total / BSZ,
                        'algorithmic_bytes_per_segment': 32000 + 32768,
                        'ratio_to_algorithmic': fe_total / BSZ / (32000 + 32768) if fe_total else None,
                        'note': 'f32 audio in (32,000 B) + log-mel out (32,768 B) per segment; the kernels of the front end '
                                'that ran in the profiled bench (melspec_r16_kernel -- or melspec_kernel with NAFP_MELSPEC_R4=1 --, plus melspec_finalize_kernel when the '
                                'log-mel tail is not deferred into conv0)'},
           'fetch_bytes_x2_corrected': fetch_bytes_per_launch, 'write_bytes': write_bytes_per_launch,
           'algorithmic_activation_bytes_per_launch': alg_bytes_per_launch,
           'launches_averaged': {'fetch': f_n, 'write': w_n},
           'method': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over '
                     '`python bench.py --steps 5 --warmup 2`; KiB -> bytes; FETCH_SIZE doubled '
                     '(gfx950 counts a 16 B/lane stream at half, MI355X_MICROARCH.md HBM)'},
          open(os.path.join(dst, 'traffic.json'), 'w'), indent=1)

sq = read_csv(os.path.join(src, 'pmc_sq', 'p_counter_collection.csv'))
agg = defaultdict(lambda: defaultdict(list))
for r in sq:
    if GEMM in r['Kernel_Name'] and int(r['Grid_Size']) >= 256 * 8:
        agg[int(r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
        agg[int(r['Grid_Size'])]['_dur'].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))

with open(os.path.join(dst, f'{tag}_summary.md'), 'w') as f:
    f.write(f'# Profile summary {tag}\n\n')
    f.write('Command profiled: `python bench.py --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-pipelined --no-train --no-e2e` '
            '(1x MI355X, BSZ 640 per step, single stream) under `rocprofv3 --kernel-trace --stats` '
            '(full CSV: `%s_kernel_stats.csv`).\n\n' % tag)
    f.write(f'Bench line of the same run (un-profiled): **{bj["value"]} {bj["unit"]}**, '
            f'roofline {bj["roofline"]["achieved"]} / {bj["roofline"]["peak"]} TFLOP/s '
            f'= {bj["roofline"]["frac"]} (HIP-event mean launch {bj["roofline"]["ms_per_launch_avg"]} ms).\n\n')
    f.write('## Kernel stats (rocprofv3 --stats)\n\n| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|\n')
    for r in stats[:12]:
        f.write(f'| `{r["Name"][:70]}` | {r["Calls"]} | {float(r["AverageNs"]) / 1e3:.1f} | '
                f'{float(r["TotalDurationNs"]) / 1e6:.2f} | {r["Percentage"]} |\n')
    # per bench step (from one melspec_kernel to the next): the durations of the GEMM-conv kernels and their split-K
    # finish kernels, summed; the timed region's HIP events give the SPAN from the first to the last of them
    tr = sorted(trace, key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(tr) if ('melspec_kernel' in r['Kernel_Name'] or 'melspec_r16_kernel' in r['Kernel_Name'])]
    per_step = []
    for a_, b_ in zip(marks[:-1], marks[1:]):
        ks = [r for r in tr[a_:b_] if GEMM in r['Kernel_Name'] or 'splitk_finish' in r['Kernel_Name']]
        if ks:
            per_step.append((sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in ks),
                             int(ks[-1]['End_Timestamp']) - int(ks[0]['Start_Timestamp']), len(ks)))
    per_step = per_step[5:35]                      # the 30 timed steps (after 5 warm-up steps; the per-conv pass follows)
    busy = sum(p[0] for p in per_step) / len(per_step) / 1e3
    span = sum(p[1] for p in per_step) / len(per_step) / 1e3
    f.write(f'\nGEMM convs of one timed bench step under the profiler ({per_step[0][2]} kernels: 15 `conv_gemm_*` launches + the '
            f'split-K finish kernels of the launches that still use one): kernel durations sum to {busy:.0f} us '
            f'= {busy / 15:.1f} us per conv launch; first start to last end {span:.0f} us = {span / 15:.1f} us per launch.  '
            f'The un-profiled HIP events of the timed region measure that span: {bj["roofline"]["ms_per_launch_avg"] * 1e3:.1f} us '
            f'per launch (`roofline.ms_per_launch_avg`).\n\n')
    # per-conv kernel durations of the timed steps (4 stamps per forward, none between the convs): mean over the steps
    seqs = []
    for a_, b_ in list(zip(marks[:-1], marks[1:]))[5:35]:
        ks = [r for r in tr[a_:b_] if GEMM in r['Kernel_Name'] or 'splitk_finish' in r['Kernel_Name']]
        seqs.append([(r['Kernel_Name'].split('(')[0].replace('nafp::', '').replace('void ', ''),
                      (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in ks])
    if seqs and all(len(q) == len(seqs[0]) for q in seqs):
        macs_l = bench.conv_effective_macs()
        f.write('Per launch, in launch order (mean over the 30 timed steps, kernel durations from the trace; a split-K finish '
                'kernel is added to its conv):\n\n| conv | kernel | us | TFLOP/s |\n|---|---|---|---|\n')
        j = 0; rows_out = []
        for k in range(len(seqs[0])):
            name = seqs[0][k][0]; us = sum(q[k][1] for q in seqs) / len(seqs)
            if 'splitk_finish' in name:
                rows_out[-1][2] += us; rows_out[-1][1] += ' + finish'
            else:
                j += 1; rows_out.append([j, name, us])
        for j, name, us in rows_out:
            f.write(f'| {j} | `{name}` | {us:.1f} | {2 * macs_l[j] * BSZ / (us * 1e-6) / 1e12:.0f} |\n')
        f.write('\n')
    f.write('## HBM traffic of the dominant kernel (separate PMC passes)\n\n')
    if traffic:
        f.write(f'* FETCH_SIZE (x2 gfx950 correction): {fetch_bytes_per_launch / 1e6:.1f} MB per launch (mean of {f_n} bench steps, 15 launches + split-K finish kernels each)\n')
        f.write(f'* WRITE_SIZE: {write_bytes_per_launch / 1e6:.1f} MB per launch (mean of {w_n} bench steps)\n')
        f.write(f'* total {traffic / 1e6:.1f} MB per launch vs {alg_bytes_per_launch / 1e6:.1f} MB algorithmic '
                f'activation bytes (each z tensor read once + written once; weights/G/Hb/gamma not counted)\n\n')
    if front:
        f.write('## HBM traffic of the front end and of the Cin = 1 conv (same PMC passes, per launch of 640 segments)\n\n')
        f.write('| kernel | FETCH x2 MB | WRITE MB | bytes / segment | us (under PMC) | GB/s |\n|---|---|---|---|---|---|\n')
        for k, kt in front.items():
            tot = kt['fetch_bytes_x2_corrected'] + kt['write_bytes']
            f.write(f'| `{k}` | {kt["fetch_bytes_x2_corrected"] / 1e6:.2f} | {kt["write_bytes"] / 1e6:.2f} | '
                    f'{kt["bytes_per_segment"]:.0f} | {kt["mean_us_under_pmc"]:.1f} | {tot / kt["mean_us_under_pmc"] / 1e3:.0f} |\n')

        f.write(f'\nFront end (STFT/mel path) total: {fe_total / BSZ:.0f} B per segment vs 64,768 B algorithmic '
                f'(32,000 B f32 audio + 32,768 B log-mel) = {fe_total / BSZ / 64768:.2f}x.\n\n')

    # ---- the front end in SQ counters (two passes: pmc_sq and pmc_sq2) ----
    fe = defaultdict(list)
    for d_ in ('pmc_sq', 'pmc_sq2'):
        pth = os.path.join(src, d_, 'p_counter_collection.csv')
        if not os.path.exists(pth):
            continue
        for r in read_csv(pth):
            if 'melspec_r16_kernel' in r['Kernel_Name']:
                fe[r['Counter_Name']].append(float(r['Counter_Value']))
                fe['_dur_' + d_].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    if fe.get('SQ_WAVE_CYCLES'):
        m = {k: sum(v) / len(v) for k, v in fe.items()}
        dur_us = m.get('_dur_pmc_sq2', m.get('_dur_pmc_sq', 0.0)) / 1e3
        wc = m['SQ_WAVE_CYCLES']                       # quad-cycles summed over waves
        f.write('\n## The front end (`melspec_r16_kernel`, 640 segments per launch) in SQ counters\n\n')
        f.write('Separate `--pmc` passes of the bench command (`pmc_sq`, `pmc_sq2`); SQ_* cycle counters are in quad-cycles summed over '
                'all waves (MI355X_MICROARCH.md), shares are of SQ_WAVE_CYCLES.\n\n| quantity | value |\n|---|---|\n')
        f.write(f'| kernel duration under PMC | {dur_us:.1f} us |\n')
        if 'SQ_WAVES' in m:
            f.write(f'| waves launched | {m["SQ_WAVES"]:.0f} (= {m["SQ_WAVES"] / 4:.0f} workgroups of 4 waves) |\n')
        if 'GRBM_GUI_ACTIVE' in m and m.get('_dur_pmc_sq'):
            cyc = m['GRBM_GUI_ACTIVE'] / 8.0               # the counter is summed over the 8 XCDs
            clk = cyc / m['_dur_pmc_sq']
            f.write(f'| shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) | {clk:.2f} GHz |\n')
            occ = wc * 4 / (cyc * 256)
            f.write(f'| mean resident waves per CU (SQ_WAVE_CYCLES x 4 / (cycles x 256 CUs)) | {occ:.1f} of 8 possible at 200 VGPRs / 80.9 KB LDS (2 workgroups) |\n')
        for name, label in (('SQ_ACTIVE_INST_ANY', 'issuing an instruction (ACTIVE_INST_ANY)'),
                            ('SQ_ACTIVE_INST_VALU', '... a vector ALU instruction (ACTIVE_INST_VALU)'),
                            ('SQ_ACTIVE_INST_LDS', '... an LDS instruction (ACTIVE_INST_LDS)'),
                            ('SQ_WAIT_INST_ANY', 'waiting to issue (WAIT_INST_ANY: dependency / pipe busy)'),
                            ('SQ_WAIT_INST_LDS', '... of which on the LDS pipe (WAIT_INST_LDS)'),
                            ('SQ_WAIT_ANY', 'parked on s_waitcnt / barrier (WAIT_ANY)')):
            if name in m:
                f.write(f'| wave time {label} | {m[name] / wc * 100:.1f} % |\n')
        if 'SQ_INSTS_VALU' in m and 'SQ_WAVES' in m:
            f.write(f'| vector ALU instructions per wave | {m["SQ_INSTS_VALU"] / m["SQ_WAVES"]:.0f} |\n')
        if 'SQ_INSTS_LDS' in m and 'SQ_WAVES' in m:
            f.write(f'| LDS instructions per wave | {m["SQ_INSTS_LDS"] / m["SQ_WAVES"]:.0f} |\n')
        if 'SQ_LDS_BANK_CONFLICT' in m and 'SQ_LDS_IDX_ACTIVE' in m and m['SQ_LDS_IDX_ACTIVE']:
            f.write(f'| LDS bank-conflict cycles / LDS active cycles | {m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"] * 100:.1f} % |\n')
        if 'SQ_INSTS_VALU' in m and dur_us:
            # a wave64 VALU instruction occupies its SIMD for 4 cycles (2 for the packed-rate ones): issue-time floor of the kernel
            ghz = (m['GRBM_GUI_ACTIVE'] / 8.0 / m['_dur_pmc_sq']) if ('GRBM_GUI_ACTIVE' in m and m.get('_dur_pmc_sq')) else 2.1
            floor_us = m['SQ_INSTS_VALU'] * 4 / (256 * 4) / (ghz * 1e3)
            f.write(f'| VALU issue-time floor (instructions x 4 cycles / 1024 SIMDs at {ghz:.2f} GHz) | {floor_us:.1f} us = {floor_us / dur_us * 100:.0f} % of the kernel |\n')
        f.write('\n')
    f.write('## SQ counters per GEMM-conv shape (grid threads -> mean over launches)\n\n')
    f.write('| grid threads | dur us | clock GHz | MFMA busy % | wave occupancy/CU | WAIT_ANY % | WAIT_INST % | LDS bank conflicts |\n|---|---|---|---|---|---|---|---|\n')
    for g in sorted(agg, reverse=True):
        a = {k: sum(v) / len(v) for k, v in agg[g].items()}
        if 'GRBM_GUI_ACTIVE' not in a:
            continue
        cyc = a['GRBM_GUI_ACTIVE'] / 8.0
        clk = cyc / a['_dur']
        mf = a['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc) * 100
        occ = a['SQ_WAVE_CYCLES'] * 4 / (cyc * 256)
        f.write(f'| {g} | {a["_dur"] / 1e3:.1f} | {clk:.2f} | {mf:.1f} | {occ:.1f} | '
                f'{a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"] * 100:.1f} | {a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"] * 100:.1f} | '
                f'{a.get("SQ_LDS_BANK_CONFLICT", 0):.0f} |\n')
# ---- train steps (tools/train_probe.py under --kernel-trace --stats) ----
def train_section(dirname, probe_name, csv_name, title, cmd, layers_b=None):
    tt = os.path.join(src, dirname, 't_kernel_stats.csv')
    if not os.path.exists(tt):
        return
    shutil.copy(tt, os.path.join(dst, csv_name))
    ttrace = read_csv(os.path.join(src, dirname, 't_kernel_trace.csv'))
    ttrace.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(ttrace) if ('melspec_kernel' in r['Kernel_Name'] or 'melspec_r16_kernel' in r['Kernel_Name'])]
    a, b = idx[-2], idx[-1]
    agg2 = defaultdict(lambda: [0, 0.0])
    for r in ttrace[a:b]:
        n = r['Kernel_Name'].replace('nafp::', '').replace('void ', '').split('(')[0]
        agg2[n][0] += 1; agg2[n][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    span = (int(ttrace[b]['Start_Timestamp']) - int(ttrace[a]['Start_Timestamp'])) / 1e3
    pp = os.path.join(src, probe_name)
    probe = open(pp).read() if os.path.exists(pp) else ''
    with open(os.path.join(dst, f'{tag}_summary.md'), 'a') as f:
        f.write(f'\n## {title}\n\n')
        f.write(f'Command: `{cmd}` under `rocprofv3 --kernel-trace --stats` '
                f'(full CSV: `{csv_name}`).  One step (second to last of the run): '
                f'{b - a} launches, {span / 1e3:.2f} ms from first to next-first kernel.\n\n')
        f.write('| kernel | launches/step | us/step |\n|---|---|---|\n')
        for n, (c, d) in sorted(agg2.items(), key=lambda kv: -kv[1][1])[:14]:
            f.write(f'| `{n[:60]}` | {c} | {d:.0f} |\n')
        if probe:
            f.write('\nUn-profiled stage timing of the same script (torch events):\n\n```\n' + probe.strip() + '\n```\n')
        lt = os.path.join(src, f'layers_{layers_b}.txt') if layers_b else None
        if lt and os.path.exists(lt):
            f.write('\nPer layer of that step (tools/train_layer_table.py: kernel durations from the trace, useful TFLOP/s on the '
                    'effective MACs of SURVEY appendix A; wgrad includes its reduce kernel, dgrad its split-K finish):\n\n```\n' +
                    open(lt).read().strip() + '\n```\n')


train_section('train_trace', 'train_probe.txt', f'{tag}_train_kernel_stats.csv',
              'Train step (SURVEY 8d config 3: BSZ 1280 = 640 anchors + 640 replicas, Adam, 1 GPU)',
              'python tools/train_probe.py 1280 adam 5', 1280)
train_section('train5120_trace', 'train5120_probe.txt', f'{tag}_train5120_kernel_stats.csv',
              'Train step at the headline batch (BASELINE configs[3] on ONE GPU: global BSZ 5120, LAMB)',
              'python tools/train_probe.py 5120 lamb 3', 5120)
train_section('train640_trace', 'train640_probe.txt', f'{tag}_train640_kernel_stats.csv',
              'Train step at the 8-GPU operating point on ONE GPU (per-rank batch 640 of the global 5120, LAMB; no process group here: '
              'the bench object `train_rank640` runs the same step through a 1-rank RCCL group)',
              'python tools/train_probe.py 640 lamb 5', 640)
train_section('train5120x6_trace', 'train5120x6_probe.txt', f'{tag}_train5120_x6_kernel_stats.csv',
              'Train step at the headline batch with forward_train, the transposed convs and the weight gradients of layers 1 - 9 on the exact 3-way bf16 split '
              '(NAFP_BF16X3=2; global BSZ 5120, LAMB): the bench object `train_x6_experimental`',
              'NAFP_BF16X3=2 python tools/train_probe.py 5120 lamb 3')
train_section('train640x6_trace', 'train640x6_probe.txt', f'{tag}_train640_x6_kernel_stats.csv',
              'Train step at the 8-GPU operating point (per-rank batch 640) on the exact split',
              'NAFP_BF16X3=2 python tools/train_probe.py 640 lamb 5')
for name, title in (('search', 'Exact search (eval side): `python tools/search_bench.py 10000000 38000 2`'),
                    ('loader', 'Training loader + augmentation: `python tools/loader_bench.py 300`')):
    st = os.path.join(src, f'{name}_trace', 't_kernel_stats.csv')
    if not os.path.exists(st):
        continue
    shutil.copy(st, os.path.join(dst, f'{tag}_{name}_kernel_stats.csv'))
    txt = open(os.path.join(src, f'{name}_bench.txt')).read().strip().splitlines()[-4:]
    with open(os.path.join(dst, f'{tag}_summary.md'), 'a') as f:
        f.write(f'\n## {title}\n\nunder `rocprofv3 --kernel-trace --stats` (full CSV: `{tag}_{name}_kernel_stats.csv`).\n\n')
        f.write('| kernel | calls | avg us | total ms |\n|---|---|---|---|\n')
        for r in read_csv(st)[:5]:
            f.write(f'| `{r["Name"][:70]}` | {r["Calls"]} | {float(r["AverageNs"]) / 1e3:.1f} | {float(r["TotalDurationNs"]) / 1e6:.2f} |\n')
        f.write('\n```\n' + '\n'.join(txt) + '\n```\n')
# ---- [r6] the exact-split forward: per-kernel durations and matrix-pipe busy share ----
x6s = os.path.join(src, 'x6_trace', 't_kernel_stats.csv')
if os.path.exists(x6s):
    shutil.copy(x6s, os.path.join(dst, f'{tag}_x6_kernel_stats.csv'))
    with open(os.path.join(dst, f'{tag}_summary.md'), 'a') as f:
        f.write('\n## The exact 3-way bf16 split (NAFP_OPT_BF16X3 = 2; `tools/x6_per_conv.py`, B = 640, one stream)\n\n')
        for nm in ('x6_per_conv_plain.txt', 'x6_per_conv.txt'):
            pth = os.path.join(src, nm)
            if os.path.exists(pth):
                lines = [l for l in open(pth).read().splitlines() if l.startswith('opt')]
                if lines:
                    f.write(('un-profiled' if 'plain' in nm else 'under rocprofv3') + ' (per-conv ms from the dispatch-attached stamps):\n\n```\n' + '\n'.join(lines) + '\n```\n\n')
        f.write(f'Kernels under `rocprofv3 --kernel-trace --stats` (full CSV: `{tag}_x6_kernel_stats.csv`):\n\n| kernel | calls | avg us | total ms |\n|---|---|---|---|\n')
        for r in read_csv(x6s)[:12]:
            f.write(f'| `{r["Name"][:80]}` | {r["Calls"]} | {float(r["AverageNs"]) / 1e3:.1f} | {float(r["TotalDurationNs"]) / 1e6:.2f} |\n')
        pq = os.path.join(src, 'x6_pmc_sq', 'p_counter_collection.csv')
        if os.path.exists(pq):
            acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
            for r in read_csv(pq):
                k = r['Kernel_Name'].split('(')[0].replace('nafp::', '').replace('void ', '')[:60]
                acc[k][r['Counter_Name']] += float(r['Counter_Value'])
                if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
                    acc[k]['_dur'] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); cnt[k] += 1
            f.write('\nSQ counters (one `--pmc` pass; sums over the launches of the run; busy % = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x shader cycles), '
                    'GHz = GRBM_GUI_ACTIVE / 8 XCDs / duration):\n\n| kernel | launches | mean us | clock GHz | matrix pipe busy % | WAIT_ANY % | WAIT_INST % |\n|---|---|---|---|---|---|---|\n')
            for k, a in sorted(acc.items(), key=lambda kv: -kv[1].get('_dur', 0)):
                if 'conv_gemm' not in k or not a.get('GRBM_GUI_ACTIVE') or not cnt[k]:
                    continue
                cyc = a['GRBM_GUI_ACTIVE'] / 8.0
                f.write(f'| `{k}` | {cnt[k]} | {a["_dur"] / cnt[k] / 1e3:.1f} | {cyc / a["_dur"]:.2f} | {a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc) * 100:.1f} | '
                        f'{a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"] * 100:.1f} | {a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"] * 100:.1f} |\n')
for b_, title_ in ((5120, 'Global batch 5120, LAMB'), (640, 'Per-rank batch 640, LAMB')):
    sqp = os.path.join(dst, f'{tag}_train{b_}_sq.txt')
    if os.path.exists(sqp):
        with open(os.path.join(dst, f'{tag}_summary.md'), 'a') as f:
            if b_ == 5120:
                f.write('\n## SQ counters of the train step\'s kernels (one `--pmc` pass over `tools/train_probe.py`, `tools/pmc_train_sq.sh`)\n\n'
                        'MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x shader cycles); GHz = GRBM_GUI_ACTIVE / 8 XCDs / duration; sums over the '
                        'kernel\'s launches in the run.  Useful TFLOP/s of an MFMA kernel ~ 157.3 x (GHz / 2.4) x busy.\n')
            f.write(f'\n{title_}:\n\n```\n' + open(sqp).read().strip() + '\n```\n')
fs = os.path.join(ROOT, 'gpurun_out', 'fullscale_r04.json')
if os.path.exists(fs):
    rec = json.load(open(fs))
    with open(os.path.join(dst, f'{tag}_summary.md'), 'a') as f:
        f.write('\n## BASELINE configs[4] at one rank\'s full share (tests/test_gpu_configs.py::test_config4_one_rank_full_share_12_5_million_rows)\n\n')
        f.write(f'{rec["rows"]:,} rows ({rec["bytes"] / 1e9:.1f} GB of fingerprints) through `write_fingerprints_from_device_rows` '
                f'(launches of {rec["launch_rows"]} rows on {rec["streams"]} HIP streams, pinned D2H, np.memmap stores): '
                f'**{rec["seconds_incl_flush"]} s including the final flush = {rec["rows_per_s_incl_flush"]:,.0f} rows/s sustained** '
                f'({rec["seconds_before_flush"]} s before the flush); whole 125-groups (first, last, either side of a launch boundary '
                'in mid-file) equal the float64 oracle (1 - cos < 1e-5), the exact index over all rows returns every probed row at its own id.\n')
print(open(os.path.join(dst, f'{tag}_summary.md')).read())
