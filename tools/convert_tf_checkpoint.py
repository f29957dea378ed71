"""Convert a checkpoint of the REFERENCE (TensorFlow) into this build's checkpoint format.

Run it INSIDE the reference's environment (TensorFlow installed), from the root of the reference
checkout -- it uses the reference's own model code and TensorFlow's own checkpoint reader, then reads
every weight BY ATTRIBUTE (no assumption about variable ordering or object-graph key names):

    cd neural-audio-fp            # the reference
    python /path/to/this/repo/tools/convert_tf_checkpoint.py CHECKPOINT_NAME [CHECKPOINT_INDEX] [-c CONFIG] [-o OUT_DIR]

It writes OUT_DIR/CHECKPOINT_NAME/ckpt-<INDEX>.npz (default OUT_DIR = the config's LOG_ROOT_DIR +
'checkpoint/'), which `run.py generate` / `load_checkpoint` of this build read next to `.pt` files.
The arrays are named like `neural_audio_fp_amd.model.fp.nnfp.tensor_names()`:
  front_conv.<blk>.conv2d_1x3|conv2d_3x1.kernel|bias, front_conv.<blk>.BN_1x3|BN_3x1.gamma|beta,
  div_enc.fc1.kernel (Q,S,32) / .bias (Q,32), div_enc.fc2.kernel (Q,32,1) / .bias (Q,1)  [the Q Dense pairs stacked].

NOT exercised in this repository's tests: TensorFlow is absent from the build image.  It touches only public
keras attributes that the reference's model/fp/nnfp.py defines (front_conv, div_enc, conv2d_1x3, conv2d_3x1,
BN_1x3, BN_3x1, split_fc_layers) and `tf.train.Checkpoint(model=...)` exactly as model/generate.py:30-51 does.
"""
import argparse
import glob
import os
import sys

import numpy as np
import yaml


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('checkpoint_name')
    ap.add_argument('checkpoint_index', nargs='?', default=None)
    ap.add_argument('-c', '--config', default='default')
    ap.add_argument('-o', '--out', default=None)
    args = ap.parse_args()
    sys.path.insert(0, os.getcwd())
    import tensorflow as tf
    from model.fp.nnfp import get_fingerprinter
    cfg = yaml.safe_load(open(os.path.join('config', args.config + '.yaml')))
    m_fp = get_fingerprinter(cfg, trainable=False)
    m_fp(tf.zeros((1, 256, 32, 1)))                                   # build the variables
    ck_dir = cfg['DIR']['LOG_ROOT_DIR'] + f'checkpoint/{args.checkpoint_name}/'
    ckpt = tf.train.Checkpoint(model=m_fp)
    if args.checkpoint_index is None:
        path = tf.train.latest_checkpoint(ck_dir)
        if path is None:
            raise FileNotFoundError(f'no checkpoint in {ck_dir}')
    else:
        path = ck_dir + 'ckpt-' + str(args.checkpoint_index)
    ckpt.restore(path).expect_partial()
    index = int(path.split('-')[-1])
    out = {}
    n_moving = 0
    blocks = [l for l in m_fp.front_conv.layers if hasattr(l, 'conv2d_1x3')]
    assert len(blocks) == 8, len(blocks)
    for b, blk in enumerate(blocks):
        for conv, bn in (('conv2d_1x3', 'BN_1x3'), ('conv2d_3x1', 'BN_3x1')):
            c, n = getattr(blk, conv), getattr(blk, bn)
            out[f'front_conv.{b}.{conv}.kernel'] = c.kernel.numpy()
            out[f'front_conv.{b}.{conv}.bias'] = c.bias.numpy()
            out[f'front_conv.{b}.{bn}.gamma'] = n.gamma.numpy()
            out[f'front_conv.{b}.{bn}.beta'] = n.beta.numpy()
            if hasattr(n, 'moving_mean'):              # MODEL.BN = batch normalisation: the non-trainable moving statistics
                n_moving += n.moving_mean.numpy().size + n.moving_variance.numpy().size
                out[f'front_conv.{b}.{bn}.moving_mean'] = n.moving_mean.numpy()
                out[f'front_conv.{b}.{bn}.moving_variance'] = n.moving_variance.numpy()
    w1, b1, w2, b2 = [], [], [], []
    for seq in m_fp.div_enc.split_fc_layers:
        d1, d2 = seq.layers
        w1.append(d1.kernel.numpy()); b1.append(d1.bias.numpy()); w2.append(d2.kernel.numpy()); b2.append(d2.bias.numpy())
    out['div_enc.fc1.kernel'], out['div_enc.fc1.bias'] = np.stack(w1), np.stack(b1)
    out['div_enc.fc2.kernel'], out['div_enc.fc2.bias'] = np.stack(w2), np.stack(b2)
    n_par = sum(v.size for v in out.values()) - n_moving
    assert n_par == sum(int(np.prod(v.shape)) for v in m_fp.trainable_variables), 'a variable was missed'
    dst = (args.out or cfg['DIR']['LOG_ROOT_DIR'] + 'checkpoint/').rstrip('/') + f'/{args.checkpoint_name}/'
    os.makedirs(dst, exist_ok=True)
    np.savez(dst + f'ckpt-{index}.npz', **{k: v.astype(np.float32) for k, v in out.items()})
    print(f'{path} -> {dst}ckpt-{index}.npz  ({len(out)} tensors, {n_par:,} parameters)')


if __name__ == '__main__':
    main()
