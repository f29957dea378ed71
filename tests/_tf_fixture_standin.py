"""Dry run of the consumers in tests/test_golden_tf.py: writes a STAND-IN for tests/golden/hotpath_tf_v1.npz into a scratch
directory FROM THE ORACLE (not from TensorFlow -- there is none in the image), with the keys and shapes that
tools/gen_golden_tf.py emits, so that the consumer tests can be executed end to end before a real fixture exists:

    python tests/_tf_fixture_standin.py /tmp/tfgold && NAFP_TF_GOLDEN_DIR=/tmp/tfgold python -m pytest tests/test_golden_tf.py -q

It proves nothing about parity (oracle vs oracle) and its output must never be committed under tests/golden/."""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import melspec as o_mel, nnfp as o_nnfp, ntxent as o_nt, optim as o_opt, torch_ref  # noqa: E402
import _inputs  # noqa: E402
from neural_audio_fp_amd.model.fp.nnfp import tensor_names  # noqa: E402


def main(out_dir):
    os.makedirs(out_dir, exist_ok=True)
    out = {}
    x = _inputs.audio(4, seed=11)
    mel = o_mel.melspec_layer(x, dtype=np.float64).astype(np.float32)
    out['audio_seed11'], out['mel_seed11'] = x, mel
    out['mel_seed11_first2'] = o_mel.melspec_layer(x[:2], dtype=np.float64).astype(np.float32)
    w = _inputs.weights(seed=3)
    h = hashlib.sha256()
    for k in sorted(w):
        h.update(k.encode()); h.update(np.ascontiguousarray(w[k], dtype='<f4').tobytes())
    out['weights_sha256'] = np.array(h.hexdigest())
    flat = o_nnfp.front_conv(mel.astype(np.float64), w, dtype=np.float64)
    out['flat_seed11_w3'] = flat.astype(np.float32)
    out['emb_seed11_w3'] = o_nnfp.l2_normalize(o_nnfp.div_enc(flat, w)).astype(np.float32)
    for n, seed, noise in ((5, 105, 0.3), (60, 160, 0.3), (5, 105, 1.5)):
        a, b = _inputs.unit_pairs(n, seed=seed, noise=noise)
        loss, sim, _ = o_nt.compute_loss(a, b, 0.05)
        tag = f'n{n}' + ('_hard' if noise > 1 else '')
        out[f'ntxent_a_{tag}'], out[f'ntxent_b_{tag}'] = a, b
        out[f'ntxent_loss_{tag}'] = np.array([loss])
        if n == 5:
            out[f'ntxent_sim_{tag}'] = sim.astype(np.float32)
            out[f'ntxent_grad_a_{tag}'], out[f'ntxent_grad_b_{tag}'] = o_nt.grad_embeddings(a, b, 0.05)
    tf_ = torch_ref.TorchFingerprinter(w, dtype=torch.float64, requires_grad=True)
    emb = tf_(torch.from_numpy(mel).double())
    e = emb.detach().numpy()
    out['train_loss'] = np.array([o_nt.compute_loss(e[:2], e[2:], 0.05)[0]])
    d_a, d_b = o_nt.grad_embeddings(e[:2], e[2:], 0.05)
    (emb * torch.from_numpy(np.concatenate([d_a, d_b]))).sum().backward()
    okey = {'div_enc.fc1.kernel': 'div.w1', 'div_enc.fc1.bias': 'div.b1', 'div_enc.fc2.kernel': 'div.w2', 'div_enc.fc2.bias': 'div.b2'}
    for name, p in zip(tensor_names(), tf_.params):
        g = p.grad.numpy()
        out['gradnorm.' + name] = np.array([np.sqrt((g ** 2).sum())])
        if g.size <= 70000:
            out['grad.' + name] = g.astype(np.float32)
            if name.endswith('bias'):
                if name in okey:
                    k = okey[name]
                else:
                    _, blk, layer, kind = name.split('.')
                    k = f'conv{2 * int(blk) + (1 if layer.endswith("3x1") else 0)}.{kind}'
                w0 = w[k].astype(np.float64).reshape(g.shape)
                g64 = g.astype(np.float32).astype(np.float64)
                out['adam_step1.' + name] = o_opt.adam_step(w0, g64, np.zeros_like(w0), np.zeros_like(w0), 1e-4, 1)[0].astype(np.float32)
                if w0.ndim == 1:
                    out['lamb_step1.' + name] = o_opt.lamb_step(w0, g64, np.zeros_like(w0), np.zeros_like(w0), 1e-3, 1)[0].astype(np.float32)
    out['versions'] = np.array('STAND-IN written by the oracle, not by TensorFlow')
    np.savez_compressed(os.path.join(out_dir, 'hotpath_tf_v1.npz'), **out)
    print('stand-in written to', out_dir)


if __name__ == '__main__':
    main(sys.argv[1])
