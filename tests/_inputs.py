"""Seeded inputs shared by the CPU and GPU tests and by tests/gen_golden.py."""
import numpy as np

from oracle import nnfp as o_nnfp


def audio(B, seed=0, T=8000):
    """Noise + three sinusoids in 300..3900 Hz per segment, float32 (B,1,T)."""
    rng = np.random.default_rng(seed)
    t = np.arange(T) / 8000.0
    x = 0.1 * rng.normal(size=(B, 1, T))
    for b in range(B):
        f = rng.uniform(300, 3900, size=3)
        x[b, 0] += sum(0.2 * np.sin(2 * np.pi * fi * t + rng.uniform(0, 6.28)) for fi in f)
    return x.astype(np.float32)


def weight_list(w):
    """oracle weight dict -> list in library / keras-variable order (include/nafp.h)."""
    arrays = []
    for j in range(16):
        arrays += [w[f'conv{j}.kernel'], w[f'conv{j}.bias'], w[f'ln{j}.gamma'], w[f'ln{j}.beta']]
    arrays += [w['div.w1'], w['div.b1'], w['div.w2'], w['div.b2']]
    if 'bn0.moving_mean' in w:                       # MODEL.BN = 'batch_norm': the non-trainable moving statistics follow
        for j in range(16):
            arrays += [w[f'bn{j}.moving_mean'], w[f'bn{j}.moving_variance']]
    return arrays


def weights(seed=3, randomize_affine=True):
    return o_nnfp.init_weights(seed=seed, randomize_affine=randomize_affine)


def unit_pairs(n, seed, d=128, noise=0.3):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(n, d))
    b = a + noise * rng.normal(size=(n, d))
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    return a.astype(np.float32), b.astype(np.float32)
