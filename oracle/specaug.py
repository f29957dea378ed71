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
