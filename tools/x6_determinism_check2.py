"""Which side moves when front end and exact-split encoder overlap on several streams: the deferred features or the encoder's output?"""
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
    sizes, rounds = (125,) * 4, 5
    n_l = len(sizes) * rounds
    xs = [0.1 * torch.randn((sizes[i % len(sizes)], 1, 8000), generator=g, device='cuda') for i in range(n_l)]
    ref_f = [m_pre(x, group_size=125, defer=True) for x in xs]
    ref_raw = [f.raw.clone() for f in ref_f]
    ref_gs = [f.gstat.clone() for f in ref_f]
    refs = [m(f).clone() for f in ref_f]
    torch.cuda.synchronize()
    for rep in range(6):
        streams = [torch.cuda.Stream() for _ in range(4)]
        outs = []
        for i in range(n_l):
            with torch.cuda.stream(streams[i % 4]):
                f = m_pre(xs[i], group_size=125, defer=True)
                outs.append((f.raw, f.gstat, m(f), f))
        torch.cuda.synchronize()
        n_raw = sum(0 if torch.equal(o[0], ref_raw[i]) else 1 for i, o in enumerate(outs))
        if rep == 0:
            for i, o in enumerate(outs):
                d = (o[0] - ref_raw[i]).abs().reshape(o[0].shape[0], 256, 32)
                if float(d.max()) > 0:
                    nz = (d > 0).nonzero()
                    segs = sorted(set(nz[:, 0].tolist())); mels = sorted(set(nz[:, 1].tolist())); frames = sorted(set(nz[:, 2].tolist()))
                    print(f'   launch {i}: {len(nz)} values differ, max {float(d.max()):.3g}; segments {segs[:10]} ({len(segs)}), mel rows {mels[:12]} ({len(mels)}), frames {frames} ')
        n_gs = sum(0 if torch.equal(o[1], ref_gs[i]) else 1 for i, o in enumerate(outs))
        n_emb = sum(0 if torch.equal(o[2], refs[i]) else 1 for i, o in enumerate(outs))
        # the encoder again, alone, on the features the concurrent pass produced
        n_again = sum(0 if torch.equal(m(o[3]), refs[i]) else 1 for i, o in enumerate(outs))
        # ... and on the REFERENCE features once more: has the model's own state moved since the references were taken?
        n_state = sum(0 if torch.equal(m(ref_f[i]), refs[i]) else 1 for i in range(n_l))
        n_pair = sum(0 if torch.equal(m(o[3]), m(ref_f[i])) else 1 for i, o in enumerate(outs))
        print(f'opt {opt} rep {rep}: of {n_l} launches -- raw features differ {n_raw}, group statistics differ {n_gs}, fingerprints differ {n_emb}; '
              f'encoder re-run alone on the same features differs {n_again}; alone on the REFERENCE features vs the first references {n_state}; concurrent-pass features vs reference features, both alone now {n_pair}', flush=True)
