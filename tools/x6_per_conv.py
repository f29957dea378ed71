"""Per-conv ms (HIP-event stamps on the kernels' own dispatch packets) of the forward at B = 640 for NAFP_OPT_BF16X3 = 0 / 1 / 2.
    python tools/x6_per_conv.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neural_audio_fp_amd as nafp                      # noqa: E402

if __name__ == '__main__':
    B = 640
    g = torch.Generator(device='cuda').manual_seed(1)
    feat = -1.2 * torch.rand((B, 256, 32, 1), generator=g, device='cuda')
    m = nafp.FingerPrinter(seed=0)
    for opt in ((2,) if os.environ.get('X6_ONLY') else (0, 1, 2)):
        m.set_option(3, opt)
        for _ in range(4):
            m(feat)
        torch.cuda.synchronize()
        m.profile_enable(10)
        for _ in range(10):
            m(feat)
        torch.cuda.synchronize()
        rows = m.profile_read()
        m.profile_enable(0)
        avg = [sum(r[k] for r in rows) / len(rows) for k in range(17)]
        print(f'opt {opt}: conv0 {avg[0]:.3f} | ' + ' '.join(f'{v:.3f}' for v in avg[1:16]) + f' | tail {avg[16]:.3f} | convs 1-15 {sum(avg[1:16]):.3f} ms', flush=True)
