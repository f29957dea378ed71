"""Throughput of the device-side training loader: batches of 640 anchors + 640 augmented replicas
(bg mix at random SNR + 600-tap IR + normalisation) from a synthetic corpus resident in HBM.
usage: python tools/loader_bench.py [n_clips=300]"""
import os
import sys
import time
import wave

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_audio_fp_amd.model.utils.dataloader_keras import genUnbalSequence  # noqa: E402

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 300
d = '/tmp/nafp_loader'
rng = np.random.default_rng(0)


def mk(sub, n, length, scale):
    os.makedirs(f'{d}/{sub}', exist_ok=True)
    out = []
    for k in range(n):
        p = f'{d}/{sub}/{k:05d}.wav'
        if not os.path.exists(p):
            with wave.open(p, 'w') as w:
                w.setnchannels(1); w.setsampwidth(2); w.setframerate(8000)
                w.writeframes((scale * rng.normal(size=length)).astype('<i2').tobytes())
        out.append(p)
    return out


ev, bg, ir = mk('ev', n_clips, 240000, 3000), mk('bg', 40, 80000, 2000), mk('ir', 60, 4000, 500)
t0 = time.perf_counter()
ds = genUnbalSequence(ev, bsz=1280, n_anchor=640, shuffle=True, random_offset_anchor=True,
                      bg_mix_parameter=[True, bg, (0, 10)], ir_mix_parameter=[True, ir], seed=1)
ds._resident(); torch.cuda.synchronize()
print(f'{len(ev) + len(bg) + len(ir)} files, {ds.arena.total * 2 / 1e6:.0f} MB of PCM resident after {time.perf_counter() - t0:.2f} s; '
      f'{len(ds)} batches per epoch')
for rep in range(3):
    t0 = time.perf_counter(); tp = 0.0
    n = min(len(ds), 20)
    for i in range(n):
        a = time.perf_counter(); rows = ds.plan(i); tp += time.perf_counter() - a
        out = ds.run(rows)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ds.run(rows); e1.record(); torch.cuda.synchronize()
    print(f'batch of 1280 rows: {dt * 1e3:.2f} ms wall ({tp / n * 1e3:.2f} ms host plan, {e0.elapsed_time(e1):.3f} ms kernel+upload) '
          f'= {1280 / dt:.0f} segments/s')
