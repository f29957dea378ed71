"""ms per forward (B = 640) and per forward_train + backward (B = 640) of the three MODEL.BN variants, one box.
    python tools/norm_alternates_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neural_audio_fp_amd as nafp                      # noqa: E402


def timed(fn, n=20, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


if __name__ == '__main__':
    B = 640
    g = torch.Generator(device='cuda').manual_seed(1)
    feat = -1.2 * torch.rand((B, 256, 32, 1), generator=g, device='cuda')
    d_emb = torch.randn((B, 128), generator=g, device='cuda')
    for norm in ('layer_norm2d', 'batch_norm', 'layer_norm1d'):
        m = nafp.FingerPrinter(seed=0, norm=norm)
        fwd = timed(lambda: m(feat))

        def step():
            m.forward_train(feat)
            m.backward(d_emb)
        print(f'{norm:13s} forward {fwd:7.3f} ms ({B / fwd:7.1f} k segments/s)   forward_train + backward {timed(step, n=10):7.3f} ms', flush=True)
