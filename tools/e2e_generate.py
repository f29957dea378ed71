"""End-to-end generate throughput (disk -> .mm), SURVEY.md 8d config 2 second figure.
Writes N synthetic 30-s 8 kHz 16-bit mono WAV clips (seeded noise + 3 tones, like config 1), then times
write_fingerprints over them with (a) host-assembled int16 rows, (b) whole-file upload + device windows.
usage: python tools/e2e_generate.py [n_clips=600] [dir=/tmp/nafp_e2e]"""
import os
import sys
import time
import wave

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_audio_fp_amd.model import generate as g  # noqa: E402
from neural_audio_fp_amd.model.utils.audio_utils import SegmentSource  # noqa: E402

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 600
d = sys.argv[2] if len(sys.argv) > 2 else '/tmp/nafp_e2e'
os.makedirs(d, exist_ok=True)
t = np.arange(240000) / 8000.0
t0 = time.perf_counter()
for k in range(n_clips):
    p = os.path.join(d, f'{k:05d}.wav')
    if os.path.exists(p):
        continue
    rng = np.random.default_rng(1000 + k)
    x = rng.integers(-8192, 8192, size=240000).astype(np.float64)
    for f in rng.uniform(300, 3900, size=3):
        x += 4000 * np.sin(2 * np.pi * f * t)
    with wave.open(p, 'w') as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(8000)
        w.writeframes(np.clip(x, -32768, 32767).astype('<i2').tobytes())
print(f'{n_clips} clips ready in {time.perf_counter() - t0:.1f} s')
cfg = yaml.safe_load(open(os.path.join(ROOT, 'config', 'default.yaml')))
paths = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith('.wav'))[:n_clips]
t0 = time.perf_counter()
src = SegmentSource(paths, bsz=cfg['BSZ']['TS_BATCH_SZ'])
print(f'header scan of {len(paths)} files: {time.perf_counter() - t0:.2f} s, {src.n_samples} segments')
m_pre, m_fp = g.build_fp(cfg)
group = cfg['BSZ']['TS_BATCH_SZ']
configs = [(False, 640, 0), (True, 640, 0), (True, 640, 2), (True, 1920, 0), (True, 1920, 2), (True, 3840, 0), (True, 7680, 0)]
for windows, launch, prefetch in configs:
    g.LAUNCH_SEGMENTS = launch
    arr = np.memmap(os.path.join(d, 'out.mm'), dtype='float32', mode='w+', shape=(src.n_samples, 128))
    emb = g.StreamedEmbedder(m_pre, m_fp, windows=windows)
    emb.prefetch = prefetch
    emb.h_pcm = [None] * (len(emb.streams) + prefetch)
    best = 0.0
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g.write_fingerprints(src, emb, arr, group)
        arr.flush()
        best = max(best, src.n_samples / (time.perf_counter() - t0))
    print(f'{"windows" if windows else "rows   "} launch~{launch:5d} prefetch {prefetch}: {best:10.0f} segments/s end to end')
    del arr
