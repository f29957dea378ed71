"""CPU: optimizer / schedule oracle (lamb_optimizer.py:123-158, trainer.py:119-140) against
independent formulations."""
import numpy as np
import torch

from oracle import optim as o_opt


def test_cosine_decay_endpoints_and_host_entry(nafp):
    lib = nafp._lib.load()
    assert np.isclose(o_opt.cosine_decay(1e-4, 0, 1000), 1e-4)
    assert np.isclose(o_opt.cosine_decay(1e-4, 1000, 1000), 1e-4 * 1e-6)
    assert np.isclose(o_opt.cosine_decay(1e-4, 5000, 1000), 1e-4 * 1e-6)          # clamps after decay_steps
    assert np.isclose(o_opt.cosine_decay(1e-4, 500, 1000), 1e-4 * (0.5 * (1 - 1e-6) + 1e-6))
    for s in (0, 1, 17, 500, 999, 1000, 1234):
        assert np.isclose(lib.nafp_cosine_decay_lr_host(1e-4, s, 1000, 1e-6), o_opt.cosine_decay(1e-4, s, 1000), rtol=1e-6)


def test_cosine_decay_restarts_vs_torch_warm_restarts_and_host_entry(nafp):
    """LR_SCHEDULE 'COS-RESTART' (trainer.py:125-131): the oracle's CosineDecayRestarts against torch's
    CosineAnnealingWarmRestarts (T_0 = first_decay_steps, T_mult = t_mul, eta_min = alpha * lr0) and the library."""
    lib = nafp._lib.load()
    lr0, first, alpha = 1e-4, 50, 2e-6
    w = torch.zeros(1, requires_grad=True)
    opt = torch.optim.SGD([w], lr=lr0)
    sch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=first, T_mult=2, eta_min=alpha * lr0)
    for s in range(0, 400):
        want = opt.param_groups[0]['lr']
        got = o_opt.cosine_decay_restarts(lr0, s, first, 2.0, 1.0, alpha)
        assert np.isclose(got, want, rtol=1e-9, atol=1e-16), (s, got, want)
        assert np.isclose(lib.nafp_cosine_decay_restarts_lr_host(lr0, s, first, 2.0, 1.0, alpha), want, rtol=2e-6)
        opt.step(); sch.step()
    # restarts at 50, 150, 350: back to lr0
    for s in (0, 50, 150, 350):
        assert np.isclose(o_opt.cosine_decay_restarts(lr0, s, first, 2.0, 1.0, alpha), lr0)
    # t_mul = 1: plain periodic; m_mul scales every period
    assert np.isclose(o_opt.cosine_decay_restarts(1.0, 75, 50, 1.0, 0.5, 0.0), 0.5 * 0.5 * (1 + np.cos(np.pi * 0.5)))
    assert np.isclose(lib.nafp_cosine_decay_restarts_lr_host(1.0, 75, 50, 1.0, 0.5, 0.0), 0.25, rtol=1e-6)


def test_adam_matches_torch_with_rescaled_epsilon():
    # torch: w -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps_t); keras puts eps outside the bias
    # correction, i.e. eps_t = eps_keras / sqrt(1-b2^t).  One step at a time with that eps.
    rng = np.random.default_rng(0)
    w = rng.normal(size=50); m = np.zeros(50); v = np.zeros(50)
    tw = torch.tensor(w.copy(), requires_grad=True)
    for t in range(1, 6):
        g = rng.normal(size=50)
        eps_t = 1e-7 / np.sqrt(1 - 0.999 ** t)
        opt = torch.optim.Adam([tw], lr=1e-3, eps=eps_t)
        if t > 1:
            opt.state[tw] = state
            state['step'] = torch.tensor(float(t - 1))
        tw.grad = torch.tensor(g)
        opt.step()
        state = opt.state[tw]
        w, m, v = o_opt.adam_step(w, g, m, v, 1e-3, t)
        assert np.abs(w - tw.detach().numpy()).max() < 1e-12


def test_lamb_trust_ratio_properties():
    rng = np.random.default_rng(1)
    w = rng.normal(size=(8, 32)); g = rng.normal(size=(8, 32))
    w1, m1, v1 = o_opt.lamb_step(w, g, np.zeros_like(w), np.zeros_like(w), lr=1e-3, step=1)
    # first step: m_hat = g, v_hat = g^2 -> update ~ sign(g) + wd*w; step length = lr * ||w||
    assert np.isclose(np.linalg.norm(w1 - w), 1e-3 * np.linalg.norm(w), rtol=1e-6)
    # zero weights -> ratio 1 (tf.where(w_norm > 0, ..., 1.0))
    z1, _, _ = o_opt.lamb_step(np.zeros(16), np.ones(16), np.zeros(16), np.zeros(16), lr=1e-3, step=1)
    assert np.allclose(z1, -1e-3 * (1 / (1 + 1e-6)))
    # scaling the gradient does not change the step (Adam normalisation + trust ratio)
    w2, _, _ = o_opt.lamb_step(w, 10 * g, np.zeros_like(w), np.zeros_like(w), lr=1e-3, step=1)
    assert np.abs(w2 - w1).max() < 1e-5          # up to eps / |g| relative (tiny |g| entries), times lr * ratio


def test_adam_and_cosine_against_torch_implementations():
    """Independent implementations present in the image: torch.optim.Adam (same moment updates; its epsilon
    sits inside the bias-corrected denominator, so one keras step equals one torch step with
    eps_torch = eps_keras / sqrt(1 - b2^t)) and torch's CosineAnnealingLR (eta_min = alpha * lr0)."""
    import torch
    from oracle import optim as O
    rng = np.random.default_rng(0)
    w0, g = rng.normal(size=50), rng.normal(size=50) * 1e-2
    for t in (1, 2):
        m = np.zeros(50) if t == 1 else 0.1 * g_prev
        v = np.zeros(50) if t == 1 else 0.001 * g_prev ** 2
        want, m2, v2 = O.adam_step(w0, g, m, v, 1e-3, t, eps=1e-7)
        p = torch.tensor(w0.copy(), dtype=torch.float64, requires_grad=True)
        opt = torch.optim.Adam([p], lr=1e-3, betas=(0.9, 0.999), eps=1e-7 / np.sqrt(1 - 0.999 ** t))
        opt.state[p] = {'step': torch.tensor(float(t - 1)), 'exp_avg': torch.tensor(m.copy()), 'exp_avg_sq': torch.tensor(v.copy())}
        p.grad = torch.tensor(g.copy())
        opt.step()
        assert np.abs(p.detach().numpy() - want).max() < 1e-15
        assert np.abs(opt.state[p]['exp_avg'].numpy() - m2).max() < 1e-18 and np.abs(opt.state[p]['exp_avg_sq'].numpy() - v2).max() < 1e-18
        g_prev = g
    lr0, S, alpha = 1e-4, 200, 1e-6
    q = torch.nn.Parameter(torch.zeros(1))
    o = torch.optim.SGD([q], lr=lr0)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(o, T_max=S, eta_min=alpha * lr0)
    for s in range(S + 1):
        assert abs(o.param_groups[0]['lr'] - O.cosine_decay(lr0, s, S, alpha)) < 1e-15
        o.step(); sch.step()
    assert abs(O.cosine_decay(lr0, S + 50, S, alpha) - alpha * lr0) < 1e-18      # keras clamps after decay_steps
