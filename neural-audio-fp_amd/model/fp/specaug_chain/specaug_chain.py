"""Host mirror of the reference's spec-augment chain, backed by libnafp's masking kernel.

Mirrors `SpecAugChainer` / `get_specaug_chain_layer` (model/fp/specaug_chain/specaug_chain.py:
43-192) and the uniform-mask branch of `SpecNCutout.call` (layers/ncutout_tarray.py:131-186,
214-268): chain entries 'cutout' / 'horizontal' / 'vertical', `SPECAUG_PROBS`, `SPECAUG_N_HOLES`,
`SPECAUG_HOLE_FILL` in {'zeros', 'min'}, attribute `.bypass` (trainer.py:26).  One rectangle set is
shared by the batch (uniform_mask=True, the only mode the reference's factory builds); hole
sizes / centres follow the reference's integer-uniform draws; the random stream itself cannot
match TensorFlow's, so parity is by distribution and by injected rectangles.
"""
import ctypes

import numpy as np
import torch

from .... import _lib


def draw_holes(kind, H, W, n_holes, rng, hole_config=(None, None, None, None)):
    """Rectangles (f0, f1, t0, t1), inclusive, of one chain entry (ncutout_tarray.py:131-186 with
    bsz = 1 and hole_act_prob = 1, as the uniform-mask branch calls it at :254-256)."""
    if kind == 'cutout':
        minw, maxw, minh, maxh = hole_config
    elif kind == 'vertical':
        minw, maxw, minh, maxh, n_holes = 5, 16, -1, -1, 1          # specaug_chain.py:124-132
    elif kind == 'horizontal':
        minw, maxw, minh, maxh, n_holes = -1, -1, 5, 20, 1          # specaug_chain.py:134-142
    else:
        raise NotImplementedError(kind)
    full_w = (minw == -1 and maxw == -1)
    full_h = (minh == -1 and maxh == -1)
    lo_w = W // 10 if minw is None else (W if minw == -1 else minw)          # ncutout_tarray.py:222-233
    hi_w = int(W / 2.5) if maxw is None else (W if maxw == -1 else maxw)
    lo_h = H // 10 if minh is None else (H if minh == -1 else minh)
    hi_h = int(H / 2.5) if maxh is None else (H if maxh == -1 else maxh)
    rects = []
    for _ in range(n_holes):
        w = lo_w if lo_w == hi_w else int(rng.integers(lo_w, hi_w))           # tf.random.uniform int32 [lo, hi)
        h = lo_h if lo_h == hi_h else int(rng.integers(lo_h, hi_h))
        xs = W // 2 if full_w else int(rng.integers(0, W - 1))
        ys = H // 2 if full_h else int(rng.integers(0, H - 1))
        t0 = int(np.clip(xs - w // 2, 0, W - 2)); t1 = int(np.clip(xs + w // 2, 1, W - 1))
        f0 = int(np.clip(ys - h // 2, 0, H - 2)); f1 = int(np.clip(ys + h // 2, 1, H - 1))
        rects.append((f0, f1, t0, t1))
    return rects


class SpecAugChainer:
    def __init__(self, chain_config=['cutout'], probs=1.0, uniform_mask=True, n_holes=1, hole_fill='min',
                 hole_config=[None, None, None, None], seed=None, **kwargs):
        if not uniform_mask:
            raise NotImplementedError('uniform_mask=False (never built by get_specaug_chain_layer)')
        if hole_fill not in ('zeros', 'min'):
            raise NotImplementedError(hole_fill)
        for k in chain_config:
            if k not in ('cutout', 'vertical', 'horizontal'):
                raise NotImplementedError(k)
        self.chain_config = list(chain_config)
        self.probs = probs if type(probs) == list else [probs]
        if len(self.probs) < len(self.chain_config):
            self.probs = self.probs * len(self.chain_config)
        self.uniform_mask, self.hole_fill, self.n_holes, self.hole_config = uniform_mask, hole_fill, n_holes, hole_config
        self.bypass = False
        self.trainable = False
        self.rng = np.random.default_rng(seed)
        self._lib = _lib.load()

    def apply_rects(self, x, rects, active=None, fill=0.0):
        """In place: holes of `rects` -> fill for the active samples.  x: (B,F,T,1) CUDA float32.
        `fill`: a number, or a 1-element CUDA tensor (read on the device: no host round trip)."""
        _lib.require_cuda(x, 'x')
        B, F, T = x.shape[0], x.shape[1], x.shape[2]
        arr = (_lib.Rect * len(rects))(*[_lib.Rect(*r) for r in rects])
        with torch.cuda.device(x.device):
            if torch.is_tensor(fill):
                _lib.check(self._lib.nafp_specaug_apply_fill_dev(_lib.ptr(x), B, F, T, arr, len(rects), _lib.ptr(active),
                                                                 _lib.ptr(_lib.require_cuda(fill, 'fill')),
                                                                 _lib.current_stream()), 'specaug_apply_fill_dev')
            else:
                _lib.check(self._lib.nafp_specaug_apply(_lib.ptr(x), B, F, T, arr, len(rects), _lib.ptr(active),
                                                        float(fill), _lib.current_stream()), 'specaug_apply')
        return x

    def mean_dev(self, x):
        """reduce_mean(x) as a 1-element CUDA tensor (the 'min' hole filler, ncutout_tarray.py:203-204)."""
        out = torch.empty((1,), dtype=torch.float32, device=x.device)
        need = int(self._lib.nafp_specaug_mean_workspace_bytes())
        ws = torch.empty((need,), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(self._lib.nafp_specaug_mean(_lib.ptr(x), x.numel(), _lib.ptr(out), _lib.ptr(ws), need,
                                                   _lib.current_stream()), 'specaug_mean')
        return out

    def __call__(self, x, inplace=False):
        """`inplace=True`: mask `x` itself (the caller owns it and nothing else reads it -- `train_step` passes the front end's
        fresh output) instead of a copy.  With zero filling and every stage always active (the shipped configs) the stages'
        rectangles -- drawn in chain order, the same random sequence -- go to the device as ONE launch: with a constant filler the
        result does not depend on the order."""
        if self.bypass:
            return x
        x = x.float().contiguous()
        if not inplace:
            x = x.clone()
        B, H, W = x.shape[0], x.shape[1], x.shape[2]
        stages = [(k, p) for k, p in zip(self.chain_config, self.probs) if p > 0]
        if self.hole_fill == 'zeros' and stages and all(p >= 1.0 for _, p in stages) and len(stages) * self.n_holes <= 8:
            rects = []
            for kind, _ in stages:
                rects += draw_holes(kind, H, W, self.n_holes, self.rng, tuple(self.hole_config))
            self.apply_rects(x, rects, None, 0.0)
            return x
        for kind, prob in zip(self.chain_config, self.probs):
            if not prob > 0:
                continue
            rects = draw_holes(kind, H, W, self.n_holes, self.rng, tuple(self.hole_config))
            active = None
            if prob < 1.0:      # per-sample activation (ncutout_tarray.py:259)
                active = torch.from_numpy((self.rng.random(B) < prob).astype(np.uint8)).to(x.device)
            fill = self.mean_dev(x) if self.hole_fill == 'min' else 0.0       # 'min' fills with reduce_mean: :203-204
            self.apply_rects(x, rects, active, fill)
        return x

    call = __call__


def get_specaug_chain_layer(cfg, trainable=False):
    """specaug_chain.py:173-192."""
    m = SpecAugChainer(chain_config=cfg['SPEC_AUG']['SPECAUG_CHAIN'], probs=cfg['SPEC_AUG']['SPECAUG_PROBS'],
                       n_holes=cfg['SPEC_AUG']['SPECAUG_N_HOLES'], hole_fill=cfg['SPEC_AUG']['SPECAUG_HOLE_FILL'])
    m.trainable = trainable
    return m
