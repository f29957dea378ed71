"""Oracle (TEST INFRASTRUCTURE): optimizer steps and LR schedule.  PARITY UNPINNED
(see oracle/__init__.py).

Follows model/fp/lamb_optimizer.py:96-158 (LAMB, a copy of TF-Addons' LAMB) and the
trainer's choices at model/trainer.py:119-140.  Third-party semantics restated:
  * tf.keras.optimizers.Adam (OptimizerV2, non-amsgrad): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; w -= lr_t * m / (sqrt(v) + eps); eps = 1e-7;
    t = iterations + 1.
  * tf.keras.experimental.CosineDecay(lr0, S, alpha): lr0*((1-alpha)*0.5*(1+cos(pi*min(s,S)/S))+alpha).
  * tf.keras.experimental.CosineDecayRestarts: period k has length S*t_mul^k and amplitude m_mul^k.
"""
import numpy as np


def cosine_decay(lr0, step, decay_steps, alpha=1e-6):
    s = min(step, decay_steps) / decay_steps
    return lr0 * ((1 - alpha) * 0.5 * (1 + np.cos(np.pi * s)) + alpha)


def cosine_decay_restarts(lr0, step, first_decay_steps, t_mul=2.0, m_mul=1.0, alpha=0.0):
    """tf.keras.experimental.CosineDecayRestarts.__call__ (SGDR, Loshchilov & Hutter 2017), used by the reference
    for LR_SCHEDULE 'COS-RESTART' (model/trainer.py:125-131: first_decay_steps = int(0.1 * total), alpha 2e-6)."""
    f = step / first_decay_steps
    if t_mul == 1.0:
        i = np.floor(f)
        f = f - i
    else:
        i = np.floor(np.log(1.0 - f * (1.0 - t_mul)) / np.log(t_mul))
        f = (f - (1.0 - t_mul ** i) / (1.0 - t_mul)) / t_mul ** i
    return lr0 * ((1 - alpha) * 0.5 * m_mul ** i * (1 + np.cos(np.pi * f)) + alpha)


def adam_step(w, g, m, v, lr, step, b1=0.9, b2=0.999, eps=1e-7, dtype=np.float64):
    w, g, m, v = (np.asarray(a, dtype=dtype) for a in (w, g, m, v))
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    lr_t = lr * np.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    return w - lr_t * m / (np.sqrt(v) + eps), m, v


def lamb_step(w, g, m, v, lr, step, b1=0.9, b2=0.999, eps=1e-6, wd=1e-6, dtype=np.float64):
    """One keras VARIABLE (lamb_optimizer.py:123-158): every variable gets weight decay and layer
    adaptation (exclude lists are None at trainer.py:136)."""
    w, g, m, v = (np.asarray(a, dtype=dtype) for a in (w, g, m, v))
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    m_hat = m / (1 - b1 ** step)
    v_hat = v / (1 - b2 ** step)
    update = m_hat / (np.sqrt(v_hat) + eps) + wd * w
    w_norm, u_norm = np.linalg.norm(w), np.linalg.norm(update)
    ratio = w_norm / u_norm if (w_norm > 0 and u_norm > 0) else 1.0
    return w - ratio * lr * update, m, v
