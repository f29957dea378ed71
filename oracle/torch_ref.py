"""Oracle (TEST INFRASTRUCTURE): the same graph in torch-CPU float32, used as the
timed CPU baseline (`bench.py` cpu_baseline, kind "port") and as an independent
cross-check of the numpy restatement.  PARITY UNPINNED (see oracle/__init__.py).

This is the "reference CPU generate path" of BASELINE.md section 4: TensorFlow/kapre
cannot run in this image, so the reference's graph (melspectrogram.py:102-112,
nnfp.py:20-231) is restated with torch's CPU kernels (oneDNN convs, pocketfft):
it is a restatement, not TensorFlow.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import melspec as _mel
from . import nnfp as _nnfp


def melspec_layer(x, group_size=None, segment_norm=False):
    """x: (B,1,8000) float32 tensor -> (B,256,32,1).  torch.stft formulation
    (center=False on the explicitly padded signal, periodic Hann)."""
    x = torch.as_tensor(x, dtype=torch.float32)
    B = x.shape[0]
    xp = F.pad(x[:, 0, :], (512, 512))
    win = torch.hann_window(1024, periodic=True, dtype=torch.float32)
    spec = torch.stft(xp, n_fft=1024, hop_length=256, win_length=1024, window=win, center=False,
                      return_complex=True)                      # (B,513,32)
    mag = spec.abs()
    fb = torch.from_numpy(_mel.mel_filterbank())                # (256,513)
    mel = torch.einsum('mf,bft->bmt', fb, mag)                  # (B,256,32)
    y = torch.log(torch.clamp(mel + 0.06, min=1e-10)) / math.log(10)
    if group_size is None:
        group_size = max(B, 1)
    out = torch.empty_like(y)
    for g0 in range(0, B, group_size):
        g = y[g0:g0 + group_size]
        g = torch.clamp(g - g.max(), min=-80.0)
        if segment_norm:
            mn = g.min()
            g = (g - mn / 2) / torch.abs(mn / 2 + 1e-10)
        out[g0:g0 + group_size] = g
    return out[..., None]


class TorchFingerprinter:
    """Weights given in the keras shapes of oracle.nnfp.init_weights."""

    def __init__(self, w, input_shape=(256, 32, 1), dtype=torch.float32, requires_grad=False, norm='layer_norm2d'):
        """`requires_grad=True` (use dtype=torch.float64): every parameter becomes a leaf in the
        KERAS layout (self.params, ordered as include/nafp.h), so autograd yields the reference
        gradients of the train step (trainer.py:43-47) for the backward-kernel parity tests."""
        self.geo = _nnfp.conv_geometry(input_shape)
        self.dtype = dtype
        self.params = []
        self.norm = norm          # MODEL.BN (nnfp.py:63-71): 'layer_norm2d' | 'layer_norm1d' | anything else = BatchNormalization in inference mode
        self.mm, self.mv = [], []

        def leaf(a):
            t = torch.tensor(np.ascontiguousarray(a), dtype=dtype, requires_grad=requires_grad)
            self.params.append(t)
            return t
        self.k, self.b, self.g, self.bt = [], [], [], []
        for j, g in enumerate(self.geo):
            k = leaf(w[f'conv{j}.kernel'])                                         # (kh,kw,Cin,Cout)
            self.k.append(k.permute(3, 2, 0, 1))                                   # (Cout,Cin,kh,kw)
            self.b.append(leaf(w[f'conv{j}.bias']))
            if norm == 'layer_norm2d':
                # LN params (F,T,C) -> (C,F,T) for NCHW
                self.g.append(leaf(w[f'ln{j}.gamma']).permute(2, 0, 1))
                self.bt.append(leaf(w[f'ln{j}.beta']).permute(2, 0, 1))
            else:
                self.g.append(leaf(w[f'ln{j}.gamma']))                             # (C,)
                self.bt.append(leaf(w[f'ln{j}.beta']))
                if norm != 'layer_norm1d':                                         # non-trainable moving statistics: constants
                    self.mm.append(torch.tensor(np.ascontiguousarray(w[f'bn{j}.moving_mean']), dtype=dtype))
                    self.mv.append(torch.tensor(np.ascontiguousarray(w[f'bn{j}.moving_variance']), dtype=dtype))
        self.w1 = leaf(w['div.w1']); self.b1 = leaf(w['div.b1'])
        self.w2 = leaf(w['div.w2']); self.b2 = leaf(w['div.b2'])

    def front_conv(self, feat):
        x = torch.as_tensor(feat, dtype=self.dtype).permute(0, 3, 1, 2)           # NCHW: (B,1,F,T)
        for j, g in enumerate(self.geo):
            pb, pa = g['pad']
            if g['axis'] == 'T':
                x = F.pad(x, (pb, pa, 0, 0))
            else:
                x = F.pad(x, (0, 0, pb, pa))
            x = F.conv2d(x, self.k[j], self.b[j], stride=g['stride'])
            x = F.elu(x)
            if self.norm == 'layer_norm2d':
                x = F.layer_norm(x, x.shape[1:], self.g[j], self.bt[j], eps=_nnfp.LN_EPS)
            elif self.norm == 'layer_norm1d':
                x = F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), self.g[j], self.bt[j], eps=_nnfp.LN_EPS).permute(0, 3, 1, 2)
            else:
                c = (1, -1, 1, 1)
                x = (x - self.mm[j].view(c)) / torch.sqrt(self.mv[j].view(c) + 1e-3) * self.g[j].view(c) + self.bt[j].view(c)
        return x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)                      # flatten as (F,T,C)

    def div_enc(self, x):
        Q, S, H = self.w1.shape
        xs = x.reshape(x.shape[0], Q, S)
        h = F.elu(torch.einsum('bqs,qsh->bqh', xs, self.w1) + self.b1[None])
        return (torch.einsum('bqh,qho->bqo', h, self.w2) + self.b2[None])[..., 0]

    def __call__(self, feat):
        y = self.div_enc(self.front_conv(feat))
        return y * torch.rsqrt(torch.clamp((y * y).sum(1, keepdim=True), min=1e-12))


def ntxent(emb_org, emb_rep, tau=0.05):
    """cross_entropy formulation of NTxent_loss_single_gpu.py:52-82."""
    ha = emb_org if torch.is_tensor(emb_org) else torch.as_tensor(emb_org, dtype=torch.float64)
    hb = emb_rep if torch.is_tensor(emb_rep) else torch.as_tensor(emb_rep, dtype=torch.float64)
    ha, hb = ha.double(), hb.double()
    n = ha.shape[0]
    mask = ~torch.eye(n, dtype=torch.bool)
    aa = (ha @ ha.T / tau)[mask].reshape(n, n - 1)
    bb = (hb @ hb.T / tau)[mask].reshape(n, n - 1)
    ab = ha @ hb.T / tau
    ba = hb @ ha.T / tau
    tgt = torch.arange(n)
    return F.cross_entropy(torch.cat([ab, aa], 1), tgt) + F.cross_entropy(torch.cat([ba, bb], 1), tgt)
