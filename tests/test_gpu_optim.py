"""GPU parity: multi-tensor Adam / LAMB kernels (through the C ABI) vs the oracle, including
the per-variable trust ratios of the stacked divide-and-encode tensors."""
import numpy as np
import pytest
import torch

from oracle import optim as o_opt

pytestmark = pytest.mark.gpu

SHAPES = [((3, 1, 128, 128), None), ((128,), None), ((16, 2, 512), None), ((128, 8, 32), 8 * 32), ((128, 32), 32),
          ((128, 32, 1), 32), ((128, 1), 1)] + [((64 + i,), None) for i in range(60)]   # > 48 tensors: two launches


def _run(opt_cls, oracle_step, nafp_mod, steps=4, **kw):
    rng = np.random.default_rng(0)
    ws = [rng.normal(size=s).astype(np.float32) for s, _ in SHAPES]
    var_lens = [int(np.prod(s)) if vl is None else vl for s, vl in SHAPES]
    tw = [torch.from_numpy(w.copy()).cuda() for w in ws]
    ms = [np.zeros_like(w, dtype=np.float64) for w in ws]; vs = [np.zeros_like(w, dtype=np.float64) for w in ws]
    w64 = [w.astype(np.float64) for w in ws]
    sched = nafp_mod.CosineDecay(1e-3, 10, alpha=1e-6)
    opt = opt_cls(learning_rate=sched, **kw)
    for t in range(1, steps + 1):
        gs = [rng.normal(size=s).astype(np.float32) * (10.0 if i % 3 == 0 else 1.0) for i, (s, _) in enumerate(SHAPES)]
        lr = o_opt.cosine_decay(1e-3, t - 1, 10, 1e-6)
        opt.apply_gradients(zip([torch.from_numpy(g).cuda() for g in gs], tw), var_lens=var_lens)
        for i in range(len(ws)):
            vl = var_lens[i]
            if oracle_step is o_opt.lamb_step and vl != w64[i].size:
                wf, gf, mf, vf = (a.reshape(-1, vl) for a in (w64[i], gs[i].astype(np.float64), ms[i], vs[i]))
                outs = [oracle_step(wf[k], gf[k], mf[k], vf[k], lr, t) for k in range(wf.shape[0])]
                w64[i] = np.stack([o[0] for o in outs]).reshape(w64[i].shape)
                ms[i] = np.stack([o[1] for o in outs]).reshape(w64[i].shape)
                vs[i] = np.stack([o[2] for o in outs]).reshape(w64[i].shape)
            else:
                w64[i], ms[i], vs[i] = oracle_step(w64[i], gs[i].astype(np.float64), ms[i], vs[i], lr, t)
    assert opt.iterations == steps
    for i in range(len(ws)):
        # fp32 moments / update vs float64 oracle over 4 steps of size ~1e-3: abs 2e-6
        assert np.abs(tw[i].cpu().numpy() - w64[i]).max() < 2e-6, i


def test_adam_matches_oracle(nafp):
    from neural_audio_fp_amd.model.fp import lamb_optimizer as m
    _run(m.Adam, o_opt.adam_step, m)


def test_lamb_matches_oracle_with_per_variable_trust_ratio(nafp):
    from neural_audio_fp_amd.model.fp import lamb_optimizer as m
    _run(m.LAMB, o_opt.lamb_step, m)


def test_fingerprinter_variable_lengths_count_576(nafp):
    m_fp = nafp.FingerPrinter(seed=1)
    vl = m_fp.variable_lengths()
    n_vars = sum(v.numel() // l for v, l in zip(m_fp.trainable_variables, vl))
    assert n_vars == 576                      # 64 conv/LN variables + 128 x (W1, b1, W2, b2): SURVEY 8 a7
