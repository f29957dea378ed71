"""Phase timeline of one GEMM conv (diagnostic): every wave of every tile stamps the shader clock at its phase
boundaries (nafp_conv_timeline), this script turns the stamps into
  * mean cycles per phase (prologue / pipeline fill / K-loop / barrier / epilogue / statistics tail) per wave,
  * per CU: the share of the launch during which 0, 1, 2, 3 ... workgroups are inside their K-loop, and the gap between
    one workgroup leaving a CU slot and the next one's first K-step.

    python tools/conv_timeline.py [conv_index=1] [batch=640]
"""
import ctypes
import os
import sys

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neural_audio_fp_amd as nafp  # noqa: E402
import bench  # noqa: E402
from neural_audio_fp_amd import _lib  # noqa: E402

CH = [128, 128, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 1024, 1024, 1024, 1024]


def geometry(j, F=256, T=32):
    """(cin, cout, positions) of conv j of the encoder (nnfp.py:193-197: 1x3 stride (1,2) then 3x1 stride (2,1))."""
    cin = 1
    st_t = [2, 2, 2, 2, 1, 2, 1, 2]          # stride of block i's 1x3 conv along T (model/fp/nnfp.py: strides table)
    for k in range(j + 1):
        if k % 2 == 0:
            T = -(-T // st_t[k // 2])
        else:
            F = (F + 1) // 2
        if k == j:
            return cin, CH[k], F * T
        cin = CH[k]


def main():
    j = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 640
    dgrad = len(sys.argv) > 3 and sys.argv[3] == 'dgrad'          # the transposed conv of layer j in a train step
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'config', 'default.yaml')))
    lib = _lib.load()
    pre = nafp.get_melspec_layer(cfg)
    fp = nafp.FingerPrinter(seed=0)
    x = bench.make_audio(B, 0, torch).cuda()
    for _ in range(3):
        emb = fp(pre(x, group_size=B))
    torch.cuda.synchronize()
    cin, cout, pos = geometry(j)
    cap = 1 << 24
    buf = torch.zeros(cap, dtype=torch.int64, device='cuda')
    rc = lib.nafp_conv_timeline(ctypes.c_void_p(buf.data_ptr()), cap, -cin if dgrad else cin, cout, pos)
    assert rc == 0
    if dgrad:
        feat = pre(x, group_size=B)
        emb = fp.forward_train(feat)
        fp.backward(torch.randn_like(emb))
    else:
        emb = fp(pre(x, group_size=B))
    torch.cuda.synchronize()
    lib.nafp_conv_timeline(None, 0, 0, 0, 0)
    g = (ctypes.c_int * 5)()
    lib.nafp_conv_timeline_grid(g)
    gx, gy, gz, BM, BN = list(g)
    n_wg = gx * gy * gz
    nw = BM // 32
    print(f'conv {j}: cin {cin} cout {cout} positions {pos}  grid {gx} x {gy} x {gz}  tile {BM} x {BN}  {n_wg} workgroups of {nw} waves')
    t = buf[:n_wg * 64].cpu().numpy().reshape(n_wg, 8, 8)[:, :nw, :].astype(np.int64)
    hw = t[..., 0]
    cu = (hw & 0xffffffff) >> 8 & 0xf
    sh = (hw & 0xffffffff) >> 12 & 0x1
    se = (hw & 0xffffffff) >> 13 & 0x7
    simd = (hw & 0xffffffff) >> 4 & 0x3
    xcc = (hw >> 32) & 0xf
    cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu                       # one id per physical CU
    names = ['geometry (entry -> DMA issue)', 'pipeline fill (-> first operands landed)', 'K-loop', 'barrier after the K-loop',
             'epilogue (loads, ELU, stores issued)', 'statistics tail (-> end)']
    d = np.diff(t[..., 1:8], axis=-1)
    tot = (t[..., 7] - t[..., 1])
    print(f'distinct CUs seen: {len(np.unique(cu_key))}; waves per SIMD id: {np.bincount(simd.ravel())}')
    print(f'mean cycles per wave and tile: total {tot.mean():.0f}')
    for k, n in enumerate(names):
        print(f'  {n:45s} {d[..., k].mean():9.0f}  ({100 * d[..., k].mean() / tot.mean():5.1f} %)   p10 {np.percentile(d[..., k], 10):8.0f}  p90 {np.percentile(d[..., k], 90):8.0f}')
    # launch span and MFMA floor
    span = t[..., 7].max() - t[..., 1].min()
    k_steps = 3 * cin // 16
    mfma_per_wave = k_steps * 8 * 2 * (BN // 64)
    print(f'launch span {span} cycles; MFMA floor per wave-tile {mfma_per_wave * 64} cycles (x waves sharing a SIMD)')
    # per CU: workgroups inside the K-loop over time (wave 0 of each workgroup as the workgroup's clock)
    k0, k1, e1, s0 = t[:, 0, 3], t[:, 0, 4], t[:, 0, 7], t[:, 0, 1]
    keys = cu_key[:, 0]
    occ = np.zeros(16)
    gaps = []
    res_all = []
    for c in np.unique(keys):
        m = keys == c
        ev = np.concatenate([np.stack([k0[m], np.ones(m.sum())], 1), np.stack([k1[m], -np.ones(m.sum())], 1)])
        ev = ev[np.argsort(ev[:, 0], kind='stable')]
        lvl = 0
        for i in range(len(ev) - 1):
            lvl += int(ev[i, 1])
            occ[min(lvl, 15)] += ev[i + 1, 0] - ev[i, 0]
        lo, hi = s0[m].min(), e1[m].max()
        occ[0] += (ev[0, 0] - lo) + (hi - ev[-1, 0])
        # slot turnover: for each workgroup end, the next workgroup start on the same CU (greedy matching in time order)
        ends = np.sort(e1[m]); starts = np.sort(s0[m]); kstarts = k0[m][np.argsort(s0[m])]
        idx = np.searchsorted(starts, ends)
        ok = idx < len(starts)
        gaps.append((starts[idx[ok]] - ends[ok]))
        res_all.append(m.sum())
    occ /= occ.sum()
    print('share of CU time with n workgroups inside the K-loop: ' + '  '.join(f'{n}: {100 * v:.1f} %' for n, v in enumerate(occ) if v > 0.0005))
    gaps = np.concatenate(gaps)
    print(f'workgroup end -> next workgroup entry on the same CU: median {np.median(gaps):.0f} cycles, p90 {np.percentile(gaps, 90):.0f}')
    print(f'entry -> first K-step (geometry + fill): mean {(k0 - s0).mean():.0f}; K-loop end -> workgroup end: mean {(e1 - k1).mean():.0f}')
    print(f'workgroups per CU: min {min(res_all)} max {max(res_all)}')
    # per SIMD: share of the SIMD's busy span with n waves inside their K-loop, and the MFMA work per covered cycle
    sk = (cu_key * 4 + simd).ravel()
    w0, w1 = t[..., 3].ravel(), t[..., 4].ravel()
    ws, we = t[..., 1].ravel(), t[..., 7].ravel()
    order = np.argsort(sk, kind='stable')
    sk, w0, w1, ws, we = sk[order], w0[order], w1[order], ws[order], we[order]
    bounds = np.flatnonzero(np.diff(sk)) + 1
    occ = np.zeros(16); span_tot = 0.0; n_wt = 0
    for a, b in zip(np.r_[0, bounds], np.r_[bounds, len(sk)]):
        ev = np.concatenate([np.stack([w0[a:b], np.ones(b - a)], 1), np.stack([w1[a:b], -np.ones(b - a)], 1)])
        ev = ev[np.argsort(ev[:, 0], kind='stable')]
        lvl = 0
        for i in range(len(ev) - 1):
            lvl += int(ev[i, 1])
            occ[min(lvl, 15)] += ev[i + 1, 0] - ev[i, 0]
        lo, hi = ws[a:b].min(), we[a:b].max()
        occ[0] += (ev[0, 0] - lo) + (hi - ev[-1, 0])
        span_tot += hi - lo; n_wt += b - a
    print('per SIMD, share of its span with n waves inside the K-loop: ' + '  '.join(f'{n}: {100 * v / occ.sum():.1f} %' for n, v in enumerate(occ) if v / occ.sum() > 0.0005))
    print(f'MFMA cycles needed / SIMD span: {100 * n_wt * mfma_per_wave * 64 / span_tot:.1f} %;  / span with >= 1 wave in the K-loop: {100 * n_wt * mfma_per_wave * 64 / (occ.sum() - occ[0]):.1f} %')


if __name__ == '__main__':
    main()
