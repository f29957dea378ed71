"""Generate tests/golden/hotpath_v1.npz from the float64 oracle.

The reference (TensorFlow/kapre) cannot run in this image and ships no fixtures for
this path ("parity unpinned"), so these vectors are produced by the
oracle restatement, NOT by the reference.  They pin the oracle against drift and
give the GPU tests fixed expected outputs that travel with the repo.

    python tests/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from oracle import melspec as o_mel, nnfp as o_nnfp, ntxent as o_nt  # noqa: E402
import _inputs  # noqa: E402


def main():
    out = {}
    x = _inputs.audio(4, seed=11)
    feat = o_mel.melspec_layer(x, dtype=np.float64)                     # (4,256,32,1)
    out['mel_seed11'] = feat.astype(np.float32)
    out['mel_seed11_group2'] = o_mel.melspec_layer(x, group_size=2).astype(np.float32)
    w = _inputs.weights(seed=3)
    taps = []
    flat = o_nnfp.front_conv(feat, w, dtype=np.float64, taps=taps)
    out['ln_out_mean'] = np.array([t.mean() for t in taps])
    out['ln_out_absmean'] = np.array([np.abs(t).mean() for t in taps])
    out['flat_seed11_w3'] = flat.astype(np.float32)
    out['emb_seed11_w3'] = o_nnfp.l2_normalize(o_nnfp.div_enc(flat, w)).astype(np.float32)
    fb = o_mel.mel_filterbank()
    out['melbank_nnz'] = np.array([(fb != 0).sum()])
    out['melbank_rowsum'] = fb.sum(1)
    for n in (5, 60):
        a, b = _inputs.unit_pairs(n, seed=100 + n)
        loss, sim, _ = o_nt.compute_loss(a, b, tau=0.05)
        out[f'ntxent_loss_n{n}'] = np.array([loss])
        if n == 5:
            out['ntxent_sim_n5'] = sim.astype(np.float32)
    # hard pairs (replica far from its anchor) so that the gradient is not saturated to ~1e-7
    a, b = _inputs.unit_pairs(5, seed=105, noise=1.5)
    ga, gb = o_nt.grad_embeddings(a, b, tau=0.05)
    out['ntxent_grad_a_n5'] = ga
    out['ntxent_grad_b_n5'] = gb
    path = os.path.join(ROOT, 'tests', 'golden', 'hotpath_v1.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
