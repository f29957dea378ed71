"""Throughput of the exact search (eval side): N resident fingerprints, nq query segments, k = 20.
usage: python tools/search_bench.py [N=10000000] [nq=38000] [reps=3]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_audio_fp_amd.eval.eval_faiss import FlatL2Index  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 38_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
d = 128
g = torch.Generator(device='cuda').manual_seed(0)
idx = FlatL2Index(d, capacity=N)
step = 1 << 20
for a in range(0, N, step):
    x = torch.randn((min(step, N - a), d), generator=g, device='cuda')
    idx.add(torch.nn.functional.normalize(x, dim=1))
q = torch.nn.functional.normalize(torch.randn((nq, d), generator=g, device='cuda'), dim=1)
idx.search_device(q[:256], 20)
torch.cuda.synchronize()
for r in range(reps):
    t0 = time.perf_counter()
    D, I = idx.search_device(q, 20)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fl = 2.0 * nq * N * d
    print(f'N={N} nq={nq} k=20: {dt * 1e3:.1f} ms  {nq / dt:.0f} queries/s  {fl / dt / 1e12:.1f} TFLOP/s (fp32 MFMA, peak 157.3)'
          f'  index read {N * d * 4 * ((nq + 127) // 128) / dt / 1e12:.2f} TB/s (L2+HBM)')
# self check: the nearest neighbour of an index row is itself
Ds, Is = idx.search_device(idx._x[12345:12345 + 64].clone(), 1)
assert (Is[:, 0].cpu() == torch.arange(12345, 12345 + 64)).all() and float(Ds.max()) < 1e-5
