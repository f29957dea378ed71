"""NT-Xent loss + both gradients: the single-device shape (n x n) and one rank's shape of the 8-way sharded loss
(n/8 local rows against n gathered columns): usage  python tools/ntxent_bench.py [n_anchors=2560] [world=8] [d=128]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neural_audio_fp_amd as nafp  # noqa: E402
from neural_audio_fp_amd import _lib  # noqa: E402
from neural_audio_fp_amd.model.fp.NTxent_loss_single_gpu import _ntxent_call  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
lib = _lib.load()
g = torch.Generator(device='cuda').manual_seed(0)
a = torch.nn.functional.normalize(torch.randn((n, d), device='cuda', generator=g), dim=1)
b = torch.nn.functional.normalize(a + 0.3 * torch.randn((n, d), device='cuda', generator=g), dim=1)


def t(f, reps=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms = t(lambda: _ntxent_call(lib, a, b, a, b, 0, 0.05, False, True))
print(f'd = {d}: single device  {2 * n} x {2 * n}: {ms:.3f} ms (loss + both gradients)')
nl = n // world
al, bl = a[:nl].contiguous(), b[:nl].contiguous()
ms = t(lambda: _ntxent_call(lib, al, bl, a, b, 0, 0.05, False, True))
print(f'one of {world} ranks  {2 * nl} x {2 * n}: {ms:.3f} ms (loss + gradient w.r.t. all gathered rows)')
