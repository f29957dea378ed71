"""Emit REFERENCE-PRODUCED golden vectors for the TensorFlow / kapre arithmetic of the hot path.

The build image has no TensorFlow, so the oracle's parity for rows a2-a7 is "unpinned" (DESIGN.md section 2).  This is the
committed recipe that lifts the cap: run it INSIDE the reference's environment (tensorflow, kapre, librosa installed:
requirements.txt / environment.yml of the reference), from the root of the reference checkout:

    cd neural-audio-fp                                   # the reference
    python /path/to/this/repo/tools/gen_golden_tf.py     # [-c default] [-o /path/to/this/repo/tests/golden]

It imports the reference's OWN `get_melspec_layer`, `get_fingerprinter`, `NTxentLoss` and `LAMB`
(model/fp/melspec/melspectrogram.py:102-141, model/fp/nnfp.py:229-258, model/fp/NTxent_loss_single_gpu.py:22-82,
model/fp/lamb_optimizer.py), feeds them the seeded inputs the repository's tests use, and writes DATA ONLY:

    tests/golden/hotpath_tf_v1.npz        audio -> log-mel -> flat -> fingerprint -> NT-Xent loss / sim_mtx, the
                                          gradients of one train step (norm of every variable's gradient + the small
                                          tensors in full), the variables after one Adam and one LAMB step (biases)
    tests/golden/tf_ckpt_tiny/ckpt-1.*    a checkpoint WRITTEN BY TENSORFLOW (tf.train.Checkpoint(optimizer=, model=) +
                                          CheckpointManager, as model/utils/experiment_helper.py:100-111 does) of a
                                          FingerPrinter with small channel counts (a full-size one is 200 MB), and
    tests/golden/tf_ckpt_tiny_expected.npz  its variables read BY ATTRIBUTE -- pins model/utils/tf_checkpoint.py to a real bundle

The encoder weights are NOT shipped: they are `oracle.nnfp.init_weights(seed=3, randomize_affine=True)` of this
repository (numpy only), assigned into the keras model by attribute; their SHA-256 travels in the fixture so that the
tests can tell a numpy whose Generator stream differs from a real mismatch.  Nothing of the reference's source is
copied anywhere: the outputs are arrays.

tests/test_golden_tf.py consumes the files when they exist (oracle vs fixture on CPU, HIP vs fixture under -m gpu,
tf_checkpoint.py vs the bundle) and skips when they do not.  NOT exercised in this repository's image.
"""
import argparse
import hashlib
import os
import sys

import numpy as np
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _assign_by_attribute(m_fp, w):
    """oracle weight dict -> the keras variables, by the attribute names model/fp/nnfp.py defines."""
    blocks = [l for l in m_fp.front_conv.layers if hasattr(l, 'conv2d_1x3')]
    assert len(blocks) == 8, len(blocks)
    for b, blk in enumerate(blocks):
        for k, (conv, bn) in enumerate((('conv2d_1x3', 'BN_1x3'), ('conv2d_3x1', 'BN_3x1'))):
            j = 2 * b + k
            c, n = getattr(blk, conv), getattr(blk, bn)
            c.kernel.assign(w[f'conv{j}.kernel']); c.bias.assign(w[f'conv{j}.bias'])
            n.gamma.assign(w[f'ln{j}.gamma']); n.beta.assign(w[f'ln{j}.beta'])
    for q, seq in enumerate(m_fp.div_enc.split_fc_layers):
        d1, d2 = seq.layers
        d1.kernel.assign(w['div.w1'][q]); d1.bias.assign(w['div.b1'][q])
        d2.kernel.assign(w['div.w2'][q]); d2.bias.assign(w['div.b2'][q])


def _variables_by_name(m_fp):
    """[(library tensor name, [keras variables stacked in that tensor])] in the order of nnfp.tensor_names() of this build."""
    out = []
    blocks = [l for l in m_fp.front_conv.layers if hasattr(l, 'conv2d_1x3')]
    for b, blk in enumerate(blocks):
        for conv, bn in (('conv2d_1x3', 'BN_1x3'), ('conv2d_3x1', 'BN_3x1')):
            c, n = getattr(blk, conv), getattr(blk, bn)
            out += [(f'front_conv.{b}.{conv}.kernel', [c.kernel]), (f'front_conv.{b}.{conv}.bias', [c.bias]),
                    (f'front_conv.{b}.{bn}.gamma', [n.gamma]), (f'front_conv.{b}.{bn}.beta', [n.beta])]
    fc = [seq.layers for seq in m_fp.div_enc.split_fc_layers]
    out += [('div_enc.fc1.kernel', [d1.kernel for d1, _ in fc]), ('div_enc.fc1.bias', [d1.bias for d1, _ in fc]),
            ('div_enc.fc2.kernel', [d2.kernel for _, d2 in fc]), ('div_enc.fc2.bias', [d2.bias for _, d2 in fc])]
    return out


def weights_sha256(w):
    h = hashlib.sha256()
    for k in sorted(w):
        h.update(k.encode()); h.update(np.ascontiguousarray(w[k], dtype='<f4').tobytes())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('-c', '--config', default='default')
    ap.add_argument('-o', '--out', default=os.path.join(REPO, 'tests', 'golden'))
    args = ap.parse_args()
    sys.path.insert(0, os.getcwd())                       # the reference checkout
    sys.path.insert(1, REPO)                              # this repository: oracle/ (numpy only) and tests/_inputs.py
    sys.path.insert(2, os.path.join(REPO, 'tests'))
    import tensorflow as tf
    from model.fp.melspec.melspectrogram import get_melspec_layer
    from model.fp.nnfp import get_fingerprinter, FingerPrinter
    from model.fp.NTxent_loss_single_gpu import NTxentLoss
    from model.fp.lamb_optimizer import LAMB
    from oracle import nnfp as o_nnfp
    import _inputs

    cfg = yaml.safe_load(open(os.path.join('config', args.config + '.yaml')))
    tau = float(cfg['LOSS']['TAU'])
    out = {}
    # ---- a2: audio -> log-mel (melspectrogram.py:102-112) ----
    x = _inputs.audio(4, seed=11)                                            # (4,1,8000) float32
    m_pre = get_melspec_layer(cfg, trainable=False)
    mel = m_pre(tf.constant(x)).numpy()                                      # (4,256,32,1)
    out['audio_seed11'], out['mel_seed11'] = x, mel.astype(np.float32)
    out['mel_seed11_first2'] = m_pre(tf.constant(x[:2])).numpy().astype(np.float32)     # a batch of its own: other reduce_max
    # ---- a3, a4: encoder (nnfp.py:229-258) with the repository's seeded weights ----
    w = o_nnfp.init_weights(seed=3, randomize_affine=True)
    m_fp = get_fingerprinter(cfg, trainable=False)
    m_fp(tf.zeros((1, 256, 32, 1)))                                          # build the variables
    _assign_by_attribute(m_fp, w)
    out['weights_sha256'] = np.array(weights_sha256(w))
    feat = tf.constant(mel)
    out['flat_seed11_w3'] = m_fp.front_conv(feat).numpy().astype(np.float32)
    emb = m_fp(feat)
    out['emb_seed11_w3'] = emb.numpy().astype(np.float32)
    # ---- a5: NT-Xent (NTxent_loss_single_gpu.py:52-82) ----
    for n, seed, noise in ((5, 105, 0.3), (60, 160, 0.3), (5, 105, 1.5)):
        a, b = _inputs.unit_pairs(n, seed=seed, noise=noise)
        loss, sim, _ = NTxentLoss(n_org=n, n_rep=n, tau=tau).compute_loss(tf.constant(a), tf.constant(b))
        tag = f'n{n}' + ('_hard' if noise > 1 else '')
        out[f'ntxent_a_{tag}'], out[f'ntxent_b_{tag}'] = a, b
        out[f'ntxent_loss_{tag}'] = np.array([float(loss)])
        if n == 5:
            out[f'ntxent_sim_{tag}'] = sim.numpy().astype(np.float32)
            ta, tb = tf.constant(a), tf.constant(b)
            with tf.GradientTape() as t:
                t.watch([ta, tb])
                l2 = NTxentLoss(n_org=n, n_rep=n, tau=tau).compute_loss(ta, tb)[0]
            ga, gb = t.gradient(l2, [ta, tb])
            out[f'ntxent_grad_a_{tag}'], out[f'ntxent_grad_b_{tag}'] = ga.numpy(), gb.numpy()
    # ---- a7: one train step's gradients (trainer.py:42-48 without spec-augment) and optimizer updates ----
    m_fp.trainable = True
    loss_obj = NTxentLoss(n_org=2, n_rep=2, tau=tau)
    with tf.GradientTape() as t:
        e = m_fp(feat)
        loss = loss_obj.compute_loss(e[:2, :], e[2:, :])[0]
    named = _variables_by_name(m_fp)
    flat_vars = [v for _, vs in named for v in vs]
    grads = t.gradient(loss, flat_vars)
    out['train_loss'] = np.array([float(loss)])
    k = 0
    small = {}
    for name, vs in named:
        g = np.stack([grads[k + i].numpy() for i in range(len(vs))]) if len(vs) > 1 else grads[k].numpy()
        k += len(vs)
        out['gradnorm.' + name] = np.array([np.sqrt((g.astype(np.float64) ** 2).sum())])
        if g.size <= 70000:                                                     # biases, conv0, the late LN affines, fc biases
            out['grad.' + name] = g.astype(np.float32); small[name] = vs
    for which, opt in (('adam', tf.keras.optimizers.Adam(learning_rate=1e-4)), ('lamb', LAMB(learning_rate=1e-3))):
        _assign_by_attribute(m_fp, w)
        opt.apply_gradients(zip(grads, flat_vars))
        for name, vs in small.items():
            if name.endswith('bias'):
                out[f'{which}_step1.' + name] = (np.stack([v.numpy() for v in vs]) if len(vs) > 1 else vs[0].numpy()).astype(np.float32)
    out['versions'] = np.array(f'tensorflow {tf.__version__}; numpy {np.__version__}; ' +
                               '; '.join(f'{m} {__import__(m).__version__}' for m in ('kapre', 'librosa')))
    os.makedirs(args.out, exist_ok=True)
    path = os.path.join(args.out, 'hotpath_tf_v1.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes,', len(out), 'arrays')

    # ---- f4: a TF-written checkpoint of a small-channel FingerPrinter (experiment_helper.py:100-111) ----
    tiny = FingerPrinter(front_hidden_ch=[4, 4, 8, 8, 16, 16, 32, 32], emb_sz=8, fc_unit_dim=[32, 1], norm='layer_norm2d')
    tiny(tf.random.stateless_normal((2, 256, 32, 1), seed=(1, 2)))
    for i, v in enumerate(tiny.trainable_variables):                          # not the zeros / ones keras starts biases and LN with
        v.assign(tf.random.stateless_normal(v.shape, seed=(7, i)))
    opt = tf.keras.optimizers.Adam(learning_rate=1e-4)
    with tf.GradientTape() as t:
        l = tf.reduce_sum(tiny(tf.random.stateless_normal((2, 256, 32, 1), seed=(3, 4))) ** 2)
    opt.apply_gradients(zip(t.gradient(l, tiny.trainable_variables), tiny.trainable_variables))   # creates the slot variables
    ck_dir = os.path.join(args.out, 'tf_ckpt_tiny')
    os.makedirs(ck_dir, exist_ok=True)
    ckpt = tf.train.Checkpoint(optimizer=opt, model=tiny)
    tf.train.CheckpointManager(checkpoint=ckpt, directory=ck_dir, max_to_keep=3).save(checkpoint_number=1)
    exp = {}
    for name, vs in _variables_by_name(tiny):
        exp[name] = (np.stack([v.numpy() for v in vs]) if len(vs) > 1 else vs[0].numpy()).astype(np.float32)
    np.savez_compressed(os.path.join(args.out, 'tf_ckpt_tiny_expected.npz'), emb_sz=np.array([8]), **exp)
    print('wrote', ck_dir, sorted(os.listdir(ck_dir)), 'and tf_ckpt_tiny_expected.npz:', sum(v.size for v in exp.values()), 'parameters')


if __name__ == '__main__':
    main()
