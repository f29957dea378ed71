"""Host mirror of the reference's spec-augment chain, backed by libnafp's masking kernel.

Mirrors `SpecAugChainer` / `get_specaug_chain_layer` (model/fp/specaug_chain/specaug_chain.py:
43-192) and `SpecNCutout.call` (layers/ncutout_tarray.py:131-186, 214-278): chain entries 'cutout' /
'horizontal' / 'vertical', `SPECAUG_PROBS`, `SPECAUG_N_HOLES`, `SPECAUG_HOLE_FILL` in {'zeros', 'min',
'random', [min_mag, max_mag]} (default.yaml:104), attribute `.bypass` (trainer.py:26).  With
uniform_mask=True (what the reference's factory builds) one rectangle set is shared by the batch and
a sample is masked with probability `prob`; with uniform_mask=False every sample draws its own
rectangles and every hole is kept with probability `prob` (:131-186 with bsz = B, :270-276).  The
'random' fillers are a NOISE TENSOR of the input's shape that each stage draws once, when it first
sees an input (keras `build`, :106-115), and re-scales to the value range of every batch (:207-208).
Hole sizes / centres follow the reference's integer-uniform draws; the random stream itself cannot
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


def _hole_ranges(kind, H, W, n_holes, hole_config):
    """(lo_w, hi_w, lo_h, hi_h, full_w, full_h, n_holes) of one chain entry (specaug_chain.py:113-142, ncutout_tarray.py:222-248)."""
    if kind == 'cutout':
        minw, maxw, minh, maxh = hole_config
    elif kind == 'vertical':
        minw, maxw, minh, maxh, n_holes = 5, 16, -1, -1, 1
    elif kind == 'horizontal':
        minw, maxw, minh, maxh, n_holes = -1, -1, 5, 20, 1
    else:
        raise NotImplementedError(kind)
    lo_w = W // 10 if minw is None else (W if minw == -1 else minw)
    hi_w = int(W / 2.5) if maxw is None else (W if maxw == -1 else maxw)
    lo_h = H // 10 if minh is None else (H if minh == -1 else minh)
    hi_h = int(H / 2.5) if maxh is None else (H if maxh == -1 else maxh)
    return lo_w, hi_w, lo_h, hi_h, (minw == -1 and maxw == -1), (minh == -1 and maxh == -1), n_holes


def draw_holes_per_sample(kind, H, W, n_holes, rng, bsz, hole_config=(None, None, None, None)):
    """(bsz, n_holes, 4) int32 rectangles (f0, f1, t0, t1), one set per sample: `generate_mixed_mask` with bsz = B
    (ncutout_tarray.py:131-170, the uniform_mask=False branch at :270-273).  Same ranges as `draw_holes`, drawn as arrays."""
    lo_w, hi_w, lo_h, hi_h, full_w, full_h, n = _hole_ranges(kind, H, W, n_holes, tuple(hole_config))
    shp = (bsz, n)
    w = np.full(shp, lo_w) if lo_w == hi_w else rng.integers(lo_w, hi_w, shp)
    h = np.full(shp, lo_h) if lo_h == hi_h else rng.integers(lo_h, hi_h, shp)
    xs = np.full(shp, W // 2) if full_w else rng.integers(0, W - 1, shp)
    ys = np.full(shp, H // 2) if full_h else rng.integers(0, H - 1, shp)
    out = np.empty((bsz, n, 4), np.int32)
    out[..., 0] = np.clip(ys - h // 2, 0, H - 2); out[..., 1] = np.clip(ys + h // 2, 1, H - 1)
    out[..., 2] = np.clip(xs - w // 2, 0, W - 2); out[..., 3] = np.clip(xs + w // 2, 1, W - 1)
    return out


class SpecAugChainer:
    def __init__(self, chain_config=['cutout'], probs=1.0, uniform_mask=True, n_holes=1, hole_fill='min',
                 hole_config=[None, None, None, None], seed=None, **kwargs):
        # ncutout_tarray.py:88-95: 'min' | 'zeros' | 'random' | [filler_min, filler_max]
        self.filler_range = None
        if isinstance(hole_fill, (list, tuple)) and len(hole_fill) == 2:
            self.filler_range = (float(hole_fill[0]), float(hole_fill[1]))
        elif hole_fill not in ('zeros', 'min', 'random'):
            raise NotImplementedError(hole_fill)
        for k in chain_config:
            if k not in ('cutout', 'vertical', 'horizontal'):
                raise NotImplementedError(k)
        self.chain_config = list(chain_config)
        self.probs = probs if type(probs) == list else [probs]
        if len(self.probs) < len(self.chain_config):
            self.probs = self.probs * len(self.chain_config)
        self.uniform_mask, self.hole_fill, self.n_holes, self.hole_config = bool(uniform_mask), hole_fill, n_holes, hole_config
        self.bypass = False
        self.trainable = False
        self.rng = np.random.default_rng(seed)
        self._seed = seed
        self._lib = _lib.load()
        self._hf = {}            # stage index -> the stage's noise tensor (keras `build`: drawn when the stage first sees an input)

    # ---- the general form: per-sample rectangles and / or a filler tensor ------------------------------------------------------
    def _noise(self, stage, x):
        """The stage's filler tensor `hf` (ncutout_tarray.py:111-115): uniform [0, 1) ('random') or [filler_min, filler_max), of
        the shape of the FIRST input the stage sees, then fixed.  (A later batch of another size would fail to broadcast in the
        reference; here sample b reads row b % B0.)"""
        hf = self._hf.get(stage)
        if hf is None:
            g = torch.Generator(device=x.device)
            g.manual_seed((int(self._seed) if self._seed is not None else int(self.rng.integers(0, 2 ** 31))) + 7919 * (stage + 1))
            hf = torch.rand(tuple(x.shape[:3]), generator=g, device=x.device, dtype=torch.float32)
            if self.filler_range is not None:
                lo, hi = self.filler_range
                hf = hf * (hi - lo) + lo
            self._hf[stage] = hf = hf.contiguous()
        return hf

    def _scale_offset(self, x, stage):
        """(filler tensor or None, 2-element CUDA tensor {scale, offset}): a hole element becomes filler * scale + offset
        (`get_hole_filler`, ncutout_tarray.py:200-211)."""
        if self.filler_range is not None:                                   # 'random_with_range': the noise as drawn
            return self._noise(stage, x), torch.tensor([1.0, 0.0], dtype=torch.float32, device=x.device)
        if self.hole_fill == 'random':                                      # hf * (max - min) + min of THIS batch
            so = torch.empty((2,), dtype=torch.float32, device=x.device)
            need = int(self._lib.nafp_specaug_mean_workspace_bytes())
            ws = torch.empty((need,), dtype=torch.uint8, device=x.device)
            with torch.cuda.device(x.device):
                _lib.check(self._lib.nafp_specaug_range(_lib.ptr(x), x.numel(), _lib.ptr(so), _lib.ptr(ws), need, _lib.current_stream()),
                           'specaug_range')
            return self._noise(stage, x), so
        if self.hole_fill == 'min':                                         # ones * reduce_mean(x)
            return None, torch.cat([self.mean_dev(x), torch.zeros((1,), dtype=torch.float32, device=x.device)])
        return None, torch.zeros((2,), dtype=torch.float32, device=x.device)

    def apply_rects_ex(self, x, rects, active=None, filler=None, scale_offset=None):
        """In place.  rects: int array (n, 4) (one set for the batch) or (B, n, 4) (one per sample), inclusive f0, f1, t0, t1;
        active: None or (B, n) / (B,) flags; filler: None or (B0, F, T) CUDA float32; scale_offset: 2-element CUDA tensor."""
        _lib.require_cuda(x, 'x')
        B, F, T = x.shape[0], x.shape[1], x.shape[2]
        r = np.ascontiguousarray(np.asarray(rects, np.int32))
        n_sets = 1 if r.ndim == 2 else r.shape[0]
        n = r.shape[-2]
        r_dev = torch.from_numpy(r.reshape(-1)).to(x.device)
        a_dev = None
        if active is not None:
            a = np.asarray(active.cpu() if torch.is_tensor(active) else active).astype(np.uint8)
            if a.ndim == 1:
                a = np.repeat(a[:, None], n, axis=1)
            a_dev = torch.from_numpy(np.ascontiguousarray(a)).to(x.device)
        if scale_offset is None:
            scale_offset = torch.zeros((2,), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(self._lib.nafp_specaug_apply_ex(_lib.ptr(x), B, F, T, _lib.ptr(r_dev), n, n_sets, _lib.ptr(a_dev),
                                                       _lib.ptr(filler), 0 if filler is None else filler.shape[0],
                                                       _lib.ptr(_lib.require_cuda(scale_offset, 'scale_offset')), _lib.current_stream()),
                       'specaug_apply_ex')
        return x

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
        general = (not self.uniform_mask) or self.hole_fill == 'random' or self.filler_range is not None
        for stage, (kind, prob) in enumerate(zip(self.chain_config, self.probs)):
            if not prob > 0:
                continue
            if general:
                filler, so = self._scale_offset(x, stage)              # from the stage's INPUT (the chain is sequential)
                if self.uniform_mask:                                  # one set for the batch, a sample is masked with probability prob
                    rects = np.asarray(draw_holes(kind, H, W, self.n_holes, self.rng, tuple(self.hole_config)), np.int32)
                    active = None if prob >= 1.0 else (self.rng.random(B) < prob)
                else:                                                  # a set per sample, a hole is kept with probability prob
                    rects = draw_holes_per_sample(kind, H, W, self.n_holes, self.rng, B, tuple(self.hole_config))
                    active = None if prob >= 1.0 else (self.rng.random(rects.shape[:2]) < prob)
                self.apply_rects_ex(x, rects, active, filler, so)
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
                       n_holes=cfg['SPEC_AUG']['SPECAUG_N_HOLES'], hole_fill=cfg['SPEC_AUG']['SPECAUG_HOLE_FILL'])     # 'min' | 'zeros' | 'random' | [lo, hi]
    m.trainable = trainable
    return m
