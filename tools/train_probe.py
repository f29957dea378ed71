"""Time the pieces of one train step (config 3 of SURVEY.md 8d: BSZ 1280 = 640 anchors + 640 replicas, Adam)
with torch events on the current stream.  usage: python tools/train_probe.py [BSZ] [adam|lamb] [steps]"""
import os
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neural_audio_fp_amd as nafp  # noqa: E402
from neural_audio_fp_amd.model import trainer as T  # noqa: E402
from neural_audio_fp_amd.model.fp.lamb_optimizer import Adam, LAMB  # noqa: E402

bsz = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
which = sys.argv[2] if len(sys.argv) > 2 else 'adam'
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
cfg = yaml.safe_load(open(os.path.join(ROOT, 'config', 'default.yaml')))
cfg['BSZ']['TR_BATCH_SZ'], cfg['BSZ']['TR_N_ANCHOR'] = bsz, bsz // 2
m_pre, m_specaug, m_fp = T.build_fp(cfg)
opt = Adam(1e-4) if which == 'adam' else LAMB(1e-4)
loss_obj = nafp.NTxentLoss(n_org=bsz // 2, n_rep=bsz // 2, tau=0.05)
bucket = T.GradientBucket(m_fp)
X = next(iter(T.synthetic_batches(cfg, 1)(1)))


def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e


names = ['melspec+aug', 'forward_train', 'ntxent', 'backward', 'optimizer', 'set_weights(next fwd)']
tot = [0.0] * len(names)
for it in range(steps + 3):
    e = [ev()]
    x = torch.cat(X, 0)
    feat = m_specaug(m_pre(x)); e.append(ev())
    emb = m_fp.forward_train(feat); e.append(ev())
    n = bsz // 2
    loss, da, db = loss_obj.loss_and_grad(emb[:n], emb[n:]); e.append(ev())
    grads = m_fp.backward(torch.cat([da, db])); e.append(ev())
    opt.apply_gradients(zip(grads, m_fp.trainable_variables), var_lens=m_fp.variable_lengths()); m_fp.mark_dirty(); e.append(ev())
    m_fp._sync(); e.append(ev())
    torch.cuda.synchronize()
    if it >= 3:
        for k in range(len(names)):
            tot[k] += e[k].elapsed_time(e[k + 1])
for k, nme in enumerate(names):
    print(f'{nme:24s} {tot[k] / steps:8.3f} ms')
print(f'sum {sum(tot) / steps:.3f} ms   loss {float(loss):.4f}')
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(steps):
    T.train_step(X, m_pre, m_specaug, m_fp, loss_obj, opt, bucket)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
fl = 3 * 2 * (278888448 + 36864) * bsz
print(f'train_step wall {dt * 1e3:.3f} ms/step  {1 / dt:.2f} steps/s  {fl / dt / 1e12:.1f} TFLOP/s (3x forward flops)')
