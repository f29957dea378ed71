"""The fused conv0 + conv1 of the exact-split forward (conv.hip, conv_gemm_k16s2_fuse0_bf16x6) against the unfused launches: bit equality
on finished features and on the deferred front-end path, then per-conv ms of both.   NAFP_X6_FUSE0=0 python tools/x6_fuse_check.py"""
import os
import sys

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neural_audio_fp_amd as nafp                      # noqa: E402

if __name__ == '__main__':
    assert os.environ.get('NAFP_X6_FUSE0') == '0', 'run with NAFP_X6_FUSE0=0: option 1 then selects the fused form'
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'config', 'default.yaml')))
    g = torch.Generator(device='cuda').manual_seed(1)
    m = nafp.FingerPrinter(seed=0)
    m_pre = nafp.get_melspec_layer(cfg)
    m.set_option(3, 2)
    for B in (640, 9, 130):
        feat = -1.2 * torch.rand((B, 256, 32, 1), generator=g, device='cuda')
        x = 0.1 * torch.randn((B, 1, 8000), generator=g, device='cuda')
        m.set_option(1, 0)
        a, a_raw = m(feat).clone(), m(m_pre(x, group_size=125, defer=True)).clone()
        m.set_option(1, 1)
        b, b_raw = m(feat).clone(), m(m_pre(x, group_size=125, defer=True)).clone()
        print(f'B {B}: fused == unfused {bool(torch.equal(a, b))} (max diff {float((a - b).abs().max()):.3g}); deferred path {bool(torch.equal(a_raw, b_raw))} '
              f'(max diff {float((a_raw - b_raw).abs().max()):.3g}); finite {bool(torch.isfinite(b).all())}', flush=True)
    B = 640
    feat = -1.2 * torch.rand((B, 256, 32, 1), generator=g, device='cuda')
    for fused in (0, 1, 0, 1):
        m.set_option(1, fused)
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
        print(f'fused {fused}: conv0 {avg[0]:.3f} | ' + ' '.join(f'{v:.3f}' for v in avg[1:16]) + f' | tail {avg[16]:.3f} | conv0 + convs 1-15 {sum(avg[0:16]):.3f} ms', flush=True)
