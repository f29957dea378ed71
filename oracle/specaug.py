"""Oracle (TEST INFRASTRUCTURE): spec-augment masking.  PARITY UNPINNED (see oracle/__init__.py).

Follows the uniform-mask branch of SpecNCutout.call (model/fp/specaug_chain/layers/
ncutout_tarray.py:252-268) with generate_single_mask (:117-128): holes are INCLUSIVE index
ranges, x_org + x_aug with act_mask per sample, fill value from get_hole_filler (:200-211)."""
import numpy as np


def apply_holes(x, rects, active=None, fill=0.0):
    """x (B,F,T,1); rects [(f0,f1,t0,t1)] inclusive; active (B,) bool or None."""
    x = np.array(x, dtype=np.float64, copy=True)
    B, F, T = x.shape[:3]
    mask = np.zeros((F, T), bool)
    fi, ti = np.arange(F)[:, None], np.arange(T)[None, :]
    for f0, f1, t0, t1 in rects:
        mask |= (f0 <= fi) & (fi <= f1) & (t0 <= ti) & (ti <= t1)
    act = np.ones(B, bool) if active is None else np.asarray(active, bool)
    for b in range(B):
        if act[b]:
            x[b, mask, :] = fill
    return x


def apply_holes_general(x, rects, active=None, filler=None, scale=0.0, offset=0.0):
    """The general form of SpecNCutout.call (ncutout_tarray.py:252-276 with get_hole_filler, :200-211).

    x (B,F,T,1); rects (n,4) -- one set for the batch, the uniform branch -- or (B,n,4) -- `generate_mixed_mask(bsz, ...)`, the
    uniform_mask=False branch; active None, (B,) (the uniform branch's per-sample act_mask, :259) or (B,n) (the per-hole
    `tf.random.uniform([]) < hole_act_prob` of :176); filler None (ones) or (B0,F,T): the hole value is filler*scale + offset,
    i.e. hf*mean ('min'), hf*0 ('zeros'), hf*(max-min)+min ('random'), hf ('random_with_range').  float32 arithmetic like the
    reference's `x*background + holes*filler`."""
    x = np.array(x, dtype=np.float32, copy=True)
    B, F, T = x.shape[:3]
    r = np.asarray(rects)
    fi, ti = np.arange(F)[:, None], np.arange(T)[None, :]
    for b in range(B):
        rs = r if r.ndim == 2 else r[b]
        mask = np.zeros((F, T), bool)
        for k, (f0, f1, t0, t1) in enumerate(rs):
            if active is not None:
                a = np.asarray(active)
                if not (a[b] if a.ndim == 1 else a[b, k]):
                    continue
            mask |= (f0 <= fi) & (fi <= f1) & (t0 <= ti) & (ti <= t1)
        hf = np.ones((F, T), np.float32) if filler is None else np.asarray(filler, np.float32)[b % len(filler)]
        fill = (hf * np.float32(scale) + np.float32(offset)).astype(np.float32)
        x[b, mask, 0] = fill[mask]
    return x
