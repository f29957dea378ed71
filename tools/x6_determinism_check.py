"""Bit equality of the exact-split forward under concurrency: several streams, each with its OWN input, against the single-stream result
of the same input.   OPT=2 python tools/x6_determinism_check.py"""
import os
import sys

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neural_audio_fp_amd as nafp                      # noqa: E402

if __name__ == '__main__':
    g = torch.Generator(device='cuda').manual_seed(1)
    opt = int(os.environ.get('OPT', '2'))
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'config', 'default.yaml')))
    m = nafp.FingerPrinter(seed=0)
    m_pre = nafp.get_melspec_layer(cfg)
    m.set_option(3, opt)
    for sizes, with_pre, rounds in (((750, 750, 750, 750), False, 1), ((750, 750, 750, 110), False, 1), ((125,) * 4, False, 5), ((750, 750, 750, 110), True, 1),
                                    ((125,) * 4, True, 5)):
        n_l = len(sizes) * rounds
        xs = [0.1 * torch.randn((sizes[i % len(sizes)], 1, 8000), generator=g, device='cuda') for i in range(n_l)]
        feats = [m_pre(x, group_size=125) for x in xs]
        refs = [m(f).clone() for f in feats]
        torch.cuda.synchronize()
        bad = 0
        for rep in range(6):
            streams = [torch.cuda.Stream() for _ in range(4)]
            outs = []
            for i in range(n_l):
                with torch.cuda.stream(streams[i % 4]):
                    f = m_pre(xs[i], group_size=125, defer=True) if with_pre else feats[i]
                    outs.append(m(f))
            torch.cuda.synchronize()
            for i, o in enumerate(outs):
                if not torch.equal(o, refs[i]):
                    bad += 1
                    rows = (o != refs[i]).any(1).nonzero().flatten().tolist()
                    if bad <= 3:
                        print(f'  sizes {sizes} rep {rep} launch {i}: {len(rows)} rows differ (first {rows[:6]}), max {float((o - refs[i]).abs().max()):.3g}')
        print(f'opt {opt} sizes {sizes} x {rounds} front end in the loop {with_pre}: {bad} of {6 * n_l} results differ', flush=True)
